/*
 * isg_sampler.h -- the reference's distribution samplers on top of a Wichmann-Hill cursor, for
 * host and device.  Same draw order, same accept/reject tests and the same arithmetic order as
 * random.c, with libm replaced by the bit-reproducible isg_math.h functions:
 *
 *   isg_rgamma   random.c:233-250 (rgamma, beta = 1) -> rgamma1 :167-193, rexp :121-130, rgamma2 :195-231
 *   isg_rdirich  random.c:264-280
 *   isg_rnormal  random.c:283-307   (Box-Muller, cosine branch, PI = 3.141592654)
 *   isg_rgeom    random.c:311-321
 *   isg_bucket   random.c:403-430   (disc_unif's bucket search on an already-drawn uniform)
 */
#ifndef ISG_SAMPLER_H
#define ISG_SAMPLER_H
#include "isg_math.h"
#include "isg_wh.h"

#define ISG_E 2.71828182   /* random.c:7 */
#define ISG_PI 3.141592654 /* random.c:8 */

ISG_HD double isg_rgamma1_try(isg_cursor *c, double alpha)
{
	double u0 = isg_cur_next(c), u1 = isg_cur_next(c), r, x;
	if (u0 > ISG_E / (alpha + ISG_E)) {
		r = -isg_log((alpha + ISG_E) * (1 - u0) / (alpha * ISG_E));
		if (u1 > isg_pow(r, alpha - 1)) return -1;
		return r;
	}
	x = (alpha + ISG_E) * u0 / ISG_E;
	r = isg_pow(x, 1 / alpha);
	if (u1 > isg_exp(-r)) return -1;
	return r;
}

ISG_HD double isg_rgamma2_try(isg_cursor *c, double alpha)
{
	double u1, u2, c1, c2, c3, c4, c5, w;
	c1 = alpha - 1;
	c2 = (alpha - 1 / (6 * alpha)) / c1;
	c3 = 2 / c1;
	c4 = c3 + 2;
	c5 = 1 / isg_sqrt(alpha);
	do {
		u1 = isg_cur_next(c);
		u2 = isg_cur_next(c);
		if (alpha > 2.5) u1 = u2 + c5 * (1 - 1.86 * u1);
	} while ((u1 >= 1) || (u1 <= 0));
	w = c2 * u2 / u1;
	if ((c3 * u1 + w + 1 / w) > c4) {
		if ((c3 * isg_log(u1) - isg_log(w) + w) >= 1) return -1;
	}
	return c1 * w;
}

ISG_HD double isg_rgamma(isg_cursor *c, double alpha)
{
	double r = 0;
	if (alpha < 1)
		do { r = isg_rgamma1_try(c, alpha); } while (r < 0);
	if (alpha == 1) r = -isg_log(isg_cur_next(c)); /* rexp(1): -(1/1) * log(u) */
	if (alpha > 1)
		do { r = isg_rgamma2_try(c, alpha); } while (r < 0);
	return r;
}

/* out[k] ~ Dirichlet(count[k] + add); count given as doubles (integer valued) */
ISG_HD void isg_rdirich(isg_cursor *c, const double *count, int n, double *out, double add)
{
	double sum = 0;
	int k;
	for (k = 0; k < n; k++) {
		double g = isg_rgamma(c, count[k] + add);
		out[k] = g;
		sum += g;
	}
	for (k = 0; k < n; k++) out[k] /= sum;
}

ISG_HD double isg_rnormal(isg_cursor *c, double mean, double sd)
{
	double u1 = isg_cur_next(c), u2 = isg_cur_next(c);
	double theta = 2 * ISG_PI * u1;
	double r = isg_sqrt(2 * (-isg_log(u2)));
	return mean + sd * (r * isg_cos(theta));
}

/* (int) conversion with x86 cvttsd2si semantics (out of range / NaN -> INT_MIN) */
ISG_HD int isg_to_int(double v)
{
	if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
	return (int)v;
}

ISG_HD int isg_rgeom_u(double u, double p) /* rgeom with its uniform already drawn */
{
	double v = isg_log(u) / isg_log(1 - p);
	return (int)((unsigned)isg_to_int(v) + 1u);
}
ISG_HD int isg_rgeom(isg_cursor *c, double p) { return isg_rgeom_u(isg_cur_next(c), p); }

/*
 * disc_unif's search: cum[0..n-1] are the running (unnormalised) sums; they are divided by the
 * last one exactly as random.c:412-413 does (the last element becomes cum/cum), then the LAST
 * bucket with cum[i-1] < x <= cum[i] wins, bucket 0 if none matches (random.c:417-429).
 */
ISG_HD int isg_bucket(double x, const double *cum, int n)
{
	double tot = cum[n - 1], prev, cur;
	int i, flag = 0;
	prev = cum[0] / tot;
	if (x <= prev && x >= 0.0) return 0;
	for (i = 1; i < n; i++) {
		cur = cum[i] / tot;
		if (x > prev && x <= cur) flag = i;
		prev = cur;
	}
	return flag;
}

/* dt_stat (mcmc.c:1524-1546): 0 = {0}, 1 = (0,1), 2 = {1}, -1 = out of range (reference exits) */
ISG_HD int isg_dt_stat(double num)
{
	const double eps = 0.001;
	if (num <= 0.000 + eps && num >= 0.000 - eps) return 0;
	if (num >= 1.000 - eps && num <= 1.000 + eps) return 2;
	if (num >= 0.0 + eps && num < 1.000 - eps) return 1;
	return -1;
}

/* genofreq (mcmc.c:1683-1703), diploid; canonical arithmetic: f*f for pow(f,2), exact 2^-(g-1) */
ISG_HD double isg_genofreq(int hom, double f0, double f1, int generation)
{
	double result, temp;
	int i;
	if (hom) {
		result = f0 * f0;
		temp = 2 * f0 * (1 - f0);
		for (i = 1; i < generation; i++) {
			temp /= 2;
			result += temp / 2;
		}
		return result;
	}
	return 2 * f0 * f1 * isg_scalbn(1.0, -(generation - 1));
}

#endif
