/*
 * isg_oracle.c -- TEST INFRASTRUCTURE (see isg_oracle.h).  CPU restatement of the reference
 * algorithm with flat arrays and an explicit RNG object.  Every function cites the reference
 * lines it follows; in the reference configuration (libm, sequential sums, replay schedule) the
 * floating-point operations are issued in the reference's order so that results are bit-equal.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "isg_oracle.h"
#include "isg_math.h"

#define E_CONST 2.71828182  /* random.c:7 */
#define PI_CONST 3.141592654 /* random.c:8 */
#define MIN2(X, Y) (((X) > (Y)) ? (Y) : (X)) /* mcmc.h:10 */

typedef struct {
	long s1, s2, s3;   /* random.c:14-16 */
	long o1, o2, o3;   /* chain origin (keyed schedule) */
	uint64_t count;
} orc_rng;

struct orc_chain {
	orc_params p;
	int *allelenum, *geno, *valid;
	int *z, *generation, *state;
	double *freq, *qq, *qqnum, *self_rates, *indvlkh;
	double alpha, totallkh;
	orc_rng rng;
	orc_keyed_layout ky;
	long iter; /* iteration index for the keyed schedule */
	int err;
};

/* ------------------------------------------------------------------ math switches */
static double m_log(const orc_chain *c, double x) { return c->p.math ? isg_log(x) : log(x); }
static double m_exp(const orc_chain *c, double x) { return c->p.math ? isg_exp(x) : exp(x); }
static double m_pow(const orc_chain *c, double x, double y) { return c->p.math ? isg_pow(x, y) : pow(x, y); }
static double m_sqrt(const orc_chain *c, double x) { return c->p.math ? isg_sqrt(x) : sqrt(x); }
static double m_cos(const orc_chain *c, double x) { return c->p.math ? isg_cos(x) : cos(x); }

/* ------------------------------------------------------------------ RNG (random.c:19-63) */
static double rng_next(orc_chain *c)
{
	orc_rng *r = &c->rng;
	double x;
	r->s1 = (171 * r->s1) % 30269;
	r->s2 = (172 * r->s2) % 30307;
	r->s3 = (170 * r->s3) % 30323;
	r->count++;
	x = r->s1 / 30269.0 + r->s2 / 30307.0 + r->s3 / 30323.0;
	if (c->p.math) { /* canonical: fmod(x,1) for 0 <= x < 3 by exact subtraction */
		if (x >= 2.0) x -= 2.0;
		else if (x >= 1.0) x -= 1.0;
		return x;
	}
	return fmod(x, 1.0);
}

static long modpow(long a, uint64_t e, long m)
{
	long r = 1;
	a %= m;
	while (e) {
		if (e & 1) r = (r * a) % m;
		a = (a * a) % m;
		e >>= 1;
	}
	return r;
}

/* keyed schedule: jump to `pos` uniforms after the chain origin */
static void rng_seek(orc_chain *c, uint64_t pos)
{
	orc_rng *r = &c->rng;
	r->s1 = (r->o1 % 30269) * modpow(171, pos, 30269) % 30269;
	r->s2 = (r->o2 % 30307) * modpow(172, pos, 30307) % 30307;
	r->s3 = (r->o3 % 30323) * modpow(170, pos, 30323) % 30323;
	if (pos == 0) { r->s1 = r->o1; r->s2 = r->o2; r->s3 = r->o3; }
}
static int keyed(const orc_chain *c) { return c->p.sched == ORC_SCHED_KEYED; }

void orc_set_seeds(orc_chain *c, long s1, long s2, long s3)
{
	c->rng.s1 = s1; c->rng.s2 = s2; c->rng.s3 = s3;
}
void orc_get_seeds(const orc_chain *c, long *s) { s[0] = c->rng.s1; s[1] = c->rng.s2; s[2] = c->rng.s3; }
uint64_t orc_rng_count(const orc_chain *c) { return c->rng.count; }
double orc_ran1(orc_chain *c) { return rng_next(c); }

/* ------------------------------------------------------------------ samplers (random.c) */
static double rexp1(orc_chain *c) /* random.c:121-130 with lambda = 1 */
{
	double u = rng_next(c);
	return -(1 / 1.0) * m_log(c, u);
}

static double rgamma1(orc_chain *c, double alpha) /* random.c:167-193 */
{
	double u0 = rng_next(c), u1 = rng_next(c), random, x;
	if (u0 > E_CONST / (alpha + E_CONST)) {
		random = -m_log(c, (alpha + E_CONST) * (1 - u0) / (alpha * E_CONST));
		if (u1 > m_pow(c, random, alpha - 1)) return -1;
		return random;
	}
	x = (alpha + E_CONST) * u0 / E_CONST;
	random = m_pow(c, x, 1 / alpha);
	if (u1 > m_exp(c, -random)) return -1;
	return random;
}

static double rgamma2(orc_chain *c, double alpha) /* random.c:195-231 */
{
	double u1, u2, c1, c2, c3, c4, c5, w;
	int done = 1;
	c1 = alpha - 1;
	c2 = (alpha - 1 / (6 * alpha)) / c1;
	c3 = 2 / c1;
	c4 = c3 + 2;
	c5 = 1 / m_sqrt(c, alpha);
	do {
		u1 = rng_next(c);
		u2 = rng_next(c);
		if (alpha > 2.5) u1 = u2 + c5 * (1 - 1.86 * u1);
	} while ((u1 >= 1) || (u1 <= 0));
	w = c2 * u2 / u1;
	if ((c3 * u1 + w + 1 / w) > c4) {
		if ((c3 * m_log(c, u1) - m_log(c, w) + w) >= 1) done = 0;
	}
	if (done == 0) return -1;
	return c1 * w;
}

static double rgamma(orc_chain *c, double alpha) /* random.c:233-250 with beta = 1 */
{
	double random = 0;
	if (alpha < 1)
		do { random = rgamma1(c, alpha) / 1.0; } while (random < 0);
	if (alpha == 1) random = rexp1(c) / 1.0;
	if (alpha > 1)
		do { random = rgamma2(c, alpha) / 1.0; } while (random < 0);
	return random;
}

static void rdirich(orc_chain *c, const double *alpha, int n, double *out, double add) /* random.c:264-280 */
{
	double tmp, sum = 0;
	int k;
	for (k = 0; k < n; k++) {
		tmp = rgamma(c, alpha[k] + add);
		out[k] = tmp;
		sum += tmp;
	}
	for (k = 0; k < n; k++) out[k] /= sum;
}

static double rnormal(orc_chain *c, double mean, double sd) /* random.c:283-307 */
{
	double u1 = rng_next(c), u2 = rng_next(c), theta, r;
	theta = 2 * PI_CONST * u1;
	r = m_sqrt(c, 2 * (-m_log(c, u2)));
	return mean + sd * (r * m_cos(c, theta));
}

static int to_int_x86(double v) /* (int) cast as cvttsd2si does it (out of range -> INT_MIN) */
{
	if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
	return (int)v;
}

static int rgeom(orc_chain *c, double p) /* random.c:311-321 */
{
	double u = rng_next(c);
	double v = m_log(c, u) / m_log(c, 1 - p);
	return (int)((unsigned)to_int_x86(v) + 1u);
}

static int disc_unif(orc_chain *c, double *vec, int length) /* random.c:403-430 */
{
	int i, flag = 0;
	double x = rng_next(c);
	for (i = 0; i < length; i++) vec[i] /= vec[length - 1];
	if (x < 0.00 || x > vec[length - 1]) c->err = 1; /* nrerror("The value x is outside the interval!") */
	if (x <= vec[0] && x >= 0.00) flag = 0;
	else
		for (i = 1; i < length; i++)
			if (x > vec[i - 1] && x <= vec[i]) flag = i;
	return flag;
}

/* ------------------------------------------------------------------ likelihood pieces */
static double genofreq(const orc_chain *c, int a0, int a1, double f0, double f1, int generation) /* mcmc.c:1683-1703 */
{
	double result, temp;
	int i;
	if (a0 == a1) { /* chcksame()==0: homozygote */
		result = c->p.math ? f0 * f0 : pow(f0, (double)c->p.P); /* runtime exponent: no x*x folding */
		temp = 2 * f0 * (1 - f0);
		for (i = 1; i < generation; i++) {
			temp /= 2;
			result += temp / 2;
		}
	} else {
		double h = c->p.math ? isg_scalbn(1.0, -(generation - 1)) : pow(0.5, (double)(generation - 1));
		result = 2 * f0 * f1 * h;
	}
	return result;
}

#define FREQ(c, k, j, a) ((c)->freq[((long)(k) * (c)->p.L + (j)) * (c)->p.Amax + (a)])
#define GENO(c, i, j, k) ((c)->geno[((long)(i) * (c)->p.L + (j)) * (c)->p.P + (k)])
#define ZZ(c, i, j, k) ((c)->z[((long)(i) * (c)->p.L + (j)) * (c)->p.P + (k)])

typedef struct { double s; isg_acc a; int exact; } summer;
static void sum_init(summer *s, int exact) { s->s = 0; s->exact = exact; isg_acc_zero(&s->a); }
static void sum_add(summer *s, double v) { if (s->exact) isg_acc_add(&s->a, v); else s->s += v; }
static double sum_val(const summer *s) { return s->exact ? isg_acc_value(&s->a) : s->s; }
/* per-locus log-probability terms (|v| < 1024): the specialised accumulator of isg_math.h */
typedef struct { double s; isg_acc2 a; int exact; } summer2;
static void sum2_init(summer2 *s, int exact) { s->s = 0; s->exact = exact; isg_acc2_zero(&s->a); }
static void sum2_add(summer2 *s, double v) { if (s->exact) isg_acc2_add(&s->a, v); else s->s += v; }
static double sum2_val(const summer2 *s) { return s->exact ? isg_acc2_value(&s->a) : s->s; }

/* mcmc.c:1726-1773 (diploid); mode 1's log_ld_noselfing_indv (mcmc.c:1869-1890) when gen < 0 */
static double log_ld_indv(const orc_chain *c, int gen, int i)
{
	int j, m, k;
	summer2 t;
	const int K = c->p.K;
	sum2_init(&t, c->p.accum);
	for (j = 0; j < c->p.L; j++) {
		int a0, a1, z0, z1;
		if (!c->valid[(long)i * c->p.L + j]) continue;
		a0 = GENO(c, i, j, 0); a1 = GENO(c, i, j, 1);
		z0 = ZZ(c, i, j, 0); z1 = ZZ(c, i, j, 1);
		if (gen < 0) { /* mode 1 */
			sum2_add(&t, m_log(c, FREQ(c, z0, j, a0)));
			sum2_add(&t, m_log(c, FREQ(c, z1, j, a1)));
			if (a0 != a1) sum2_add(&t, m_log(c, 2));
			continue;
		}
		if (c->p.type_freq == 0) {
			double tmp[2];
			for (k = 0; k < 2; k++) {
				int a = k ? a1 : a0;
				tmp[k] = 0;
				for (m = 0; m < K; m++) tmp[k] += FREQ(c, m, j, a) * c->qq[(long)i * K + m];
			}
			sum2_add(&t, m_log(c, genofreq(c, a0, a1, tmp[0], tmp[1], gen)));
		} else {
			if (z0 == z1) {
				sum2_add(&t, m_log(c, genofreq(c, a0, a1, FREQ(c, z0, j, a0), FREQ(c, z1, j, a1), gen)));
			} else {
				sum2_add(&t, m_log(c, FREQ(c, z0, j, a0)));
				sum2_add(&t, m_log(c, FREQ(c, z1, j, a1)));
				if (a0 != a1) sum2_add(&t, m_log(c, 2));
			}
		}
	}
	return sum2_val(&t);
}

/* genofreq_inbreedcoff (mcmc.c:1705-1723), diploid */
static double genofreq_F(const orc_chain *c, int a0, int a1, double f0, double f1, double inbreed)
{
	if (a0 == a1) return m_pow(c, f0, (double)c->p.P) * (1 - inbreed) + f0 * inbreed;
	return 2 * f0 * f1 * (1 - inbreed);
}

/* the terms log_ld_F_pop (mcmc.c:1776-1809) adds for individual i, handed to `add` one by one in its order */
typedef void (*term_sink)(void *ctx, double v);
static void F_pop_terms(const orc_chain *c, const double *inbreed, int i, term_sink add, void *ctx)
{
	int j;
	for (j = 0; j < c->p.L; j++) {
		int a0, a1, z0, z1;
		if (!c->valid[(long)i * c->p.L + j]) continue;
		a0 = GENO(c, i, j, 0); a1 = GENO(c, i, j, 1);
		z0 = ZZ(c, i, j, 0); z1 = ZZ(c, i, j, 1);
		if (z0 == z1) {
			add(ctx, m_log(c, genofreq_F(c, a0, a1, FREQ(c, z0, j, a0), FREQ(c, z1, j, a1), inbreed[z0])));
		} else {
			add(ctx, m_log(c, FREQ(c, z0, j, a0)));
			add(ctx, m_log(c, FREQ(c, z1, j, a1)));
			if (a0 != a1) add(ctx, m_log(c, 2));
		}
	}
}
/* log_ld_F_indv (mcmc.c:1812-1847): the same terms with the individual's own coefficient */
static void F_indv_terms(const orc_chain *c, double inbreed, int i, term_sink add, void *ctx)
{
	int j;
	for (j = 0; j < c->p.L; j++) {
		int a0, a1, z0, z1;
		if (!c->valid[(long)i * c->p.L + j]) continue;
		a0 = GENO(c, i, j, 0); a1 = GENO(c, i, j, 1);
		z0 = ZZ(c, i, j, 0); z1 = ZZ(c, i, j, 1);
		if (z0 == z1) {
			add(ctx, m_log(c, genofreq_F(c, a0, a1, FREQ(c, z0, j, a0), FREQ(c, z1, j, a1), inbreed)));
		} else {
			add(ctx, m_log(c, FREQ(c, z0, j, a0)));
			add(ctx, m_log(c, FREQ(c, z1, j, a1)));
			if (a0 != a1) add(ctx, m_log(c, 2));
		}
	}
}
static void sink2(void *ctx, double v) { sum2_add((summer2 *)ctx, v); }
static void sink1(void *ctx, double v) { sum_add((summer *)ctx, v); }
static double log_ld_F_pop(const orc_chain *c, const double *inbreed, int i)
{
	summer2 t;
	sum2_init(&t, c->p.accum);
	F_pop_terms(c, inbreed, i, sink2, &t);
	return sum2_val(&t);
}
static double log_ld_F_indv(const orc_chain *c, double inbreed, int i)
{
	summer2 t;
	sum2_init(&t, c->p.accum);
	F_indv_terms(c, inbreed, i, sink2, &t);
	return sum2_val(&t);
}
/* log_ld_F_total (mcmc.c:1850-1868), mode 4.  Reference configuration: the per-individual values summed in order;
 * canonical configuration: ONE exact sum over all terms, rounded once (what a count-based evaluation computes) */
static double log_ld_F_total(const orc_chain *c, const double *inbreed)
{
	summer t;
	int i;
	sum_init(&t, c->p.accum);
	for (i = 0; i < c->p.N; i++) {
		if (c->p.accum) F_pop_terms(c, inbreed, i, sink1, &t);
		else sum_add(&t, log_ld_F_pop(c, inbreed, i));
	}
	return sum_val(&t);
}

static int dt_stat(orc_chain *c, double num) /* mcmc.c:1524-1546 */
{
	double eps = 0.001;
	if (num <= 0.000 + eps && num >= 0.000 - eps) return 0;
	if (num >= 1.000 - eps && num <= 1.000 + eps) return 2;
	if (num >= 0.0 + eps && num < 1.000 - eps) return 1;
	c->err = 2; /* reference: prints and exit(1) */
	return 1;
}

static double proposal(const orc_chain *c, const double *inbreed) /* mcmc.c:1630-1648 */
{
	summer ld;
	int i, j;
	const int K = c->p.K;
	sum_init(&ld, c->p.accum);
	for (i = 0; i < c->p.N; i++) {
		double temp = 0;
		for (j = 0; j < K; j++) temp += c->qq[(long)i * K + j] * inbreed[j];
		sum_add(&ld, m_log(c, m_pow(c, temp, c->generation[i] - 1) * (1 - temp)));
	}
	return sum_val(&ld);
}

static double adpt_indp(orc_chain *c, int *stat_tmp, int stat) /* mcmc.c:1461-1520 */
{
	double tmp = 0, tt;
	if (stat == 0) {
		if (rng_next(c) < 0.50) { tmp = 0.000; *stat_tmp = 0; }
		else { tmp = rng_next(c); *stat_tmp = 1; }
	} else if (stat == 2) {
		if (rng_next(c) < 0.5) { tmp = 1.000; *stat_tmp = 2; }
		else { tmp = rng_next(c); *stat_tmp = 1; }
	} else if (stat == 1) {
		tt = rng_next(c);
		if (tt <= 0.05) { tmp = 0.0000; *stat_tmp = 0; }
		else if (tt >= 0.95) { tmp = 1.000; *stat_tmp = 2; }
		else { tmp = rng_next(c); *stat_tmp = 1; }
	} else {
		c->err = 3;
	}
	return tmp;
}

static double q_trans(int a, int b) /* mcmc.c:1566-1593 */
{
	double temp = 0;
	if (a == 0) { if (b == 0) temp = 0.5; if (b == 1) temp = 0.5; }
	else if (a == 2) { if (b == 2) temp = 0.5; if (b == 1) temp = 0.5; }
	else if (a == 1) { if (b == 0 || b == 2) temp = 0.05; if (b == 1) temp = 0.90; }
	return temp;
}
static double hastings_stat(const int *tmp, const int *prev, int num) /* mcmc.c:1550-1563 */
{
	int i;
	double temp = 1.0;
	for (i = 0; i < num; i++) temp *= q_trans(prev[i], tmp[i]) / q_trans(tmp[i], prev[i]);
	return temp;
}

/* ------------------------------------------------------------------ keyed layout */
static void keyed_layout(orc_chain *c)
{
	orc_keyed_layout *k = &c->ky;
	uint64_t N = c->p.N, L = c->p.L, P = c->p.P, K = c->p.K, A = c->p.Amax;
	k->SP = 16 * A + 16;
	k->SZ = P * L + 16 * K + 16;
	k->ZI0 = 1 + 2 * N;
	k->B0 = k->ZI0 + N * k->SZ;
	k->offS = K * L * k->SP;
	k->offG = k->offS + 4 * K;
	k->offZ = k->offG + 2 * N;
	k->offA = k->offZ + N * k->SZ;
	k->BLK = k->offA + 4 + (c->p.mode == 3 ? 2 * N : 0); /* mode 3: update_S_IND of individual i at offA + 4 + 2 i */
}
void orc_keyed_get_layout(const orc_chain *c, orc_keyed_layout *out) { *out = c->ky; }
static uint64_t iter_base(const orc_chain *c) { return c->ky.B0 + (uint64_t)c->iter * c->ky.BLK; }

/* ------------------------------------------------------------------ sweeps */
void orc_count_alleles(orc_chain *c, int *cnt) /* mcmc.c:810-845 (same counts, direct indexing) */
{
	long i, j, k;
	const orc_params *p = &c->p;
	memset(cnt, 0, sizeof(int) * (size_t)p->K * p->L * p->Amax);
	for (j = 0; j < p->L; j++)
		for (i = 0; i < p->N; i++)
			if (c->valid[i * p->L + j])
				for (k = 0; k < p->P; k++)
					cnt[((long)ZZ(c, i, j, k) * p->L + j) * p->Amax + GENO(c, i, j, k)]++;
}

void orc_update_P(orc_chain *c) /* mcmc.c:799-861 */
{
	const orc_params *p = &c->p;
	int *cnt = malloc(sizeof(int) * (size_t)p->K * p->L * p->Amax);
	double *tmp = malloc(sizeof(double) * p->Amax);
	int i, j, k;
	orc_count_alleles(c, cnt);
	for (i = 0; i < p->K; i++)
		for (j = 0; j < p->L; j++)
			if (c->allelenum[j] > 1) {
				for (k = 0; k < c->allelenum[j]; k++) tmp[k] = (double)cnt[((long)i * p->L + j) * p->Amax + k];
				if (keyed(c)) rng_seek(c, iter_base(c) + ((uint64_t)i * p->L + j) * c->ky.SP);
				rdirich(c, tmp, c->allelenum[j], &FREQ(c, i, j, 0), 1.0);
			}
	free(cnt);
	free(tmp);
}

void orc_update_S_POP(orc_chain *c) /* mcmc.c:913-983 */
{
	const orc_params *p = &c->p;
	const int K = p->K;
	double delta0 = 0.05, mhratio, *tmp = malloc(sizeof(double) * K);
	int i, j, *tem_stat = malloc(sizeof(int) * K);
	if (keyed(c)) rng_seek(c, iter_base(c) + c->ky.offS);
	for (j = 0; j < K; j++) {
		for (i = 0; i < K; i++) {
			tmp[i] = c->self_rates[i];
			if (p->back_refl == 0) tem_stat[i] = c->state[i];
		}
		if (p->back_refl == 1) {
			tmp[j] = rng_next(c) * 2 * delta0 - delta0;
			tmp[j] += c->self_rates[j];
			if (tmp[j] <= 0.000) tmp[j] = 0.000 - tmp[j];
			else if (tmp[j] >= 1.000) tmp[j] = 1.000 - (tmp[j] - 1.000);
		} else {
			tmp[j] = adpt_indp(c, &tem_stat[j], c->state[j]);
		}
		mhratio = m_exp(c, proposal(c, tmp) - proposal(c, c->self_rates));
		if (p->back_refl == 0) mhratio *= hastings_stat(tem_stat, c->state, K);
		if (rng_next(c) < MIN2(1, mhratio)) {
			c->self_rates[j] = tmp[j];
			if (p->back_refl == 0) c->state[j] = tem_stat[j];
		}
	}
	free(tmp);
	free(tem_stat);
}

void orc_update_F_POP(orc_chain *c) /* update_inbreedcoff_POP, mcmc.c:986-1051 (the coefficients live in self_rates[]) */
{
	const orc_params *p = &c->p;
	const int K = p->K;
	double delta0 = 0.05, mhratio, *tmp = malloc(sizeof(double) * K);
	int i, j, *tem_stat = malloc(sizeof(int) * K);
	if (keyed(c)) rng_seek(c, iter_base(c) + c->ky.offS);
	for (j = 0; j < K; j++) {
		for (i = 0; i < K; i++) {
			tmp[i] = c->self_rates[i];
			if (p->back_refl == 0) tem_stat[i] = c->state[i];
		}
		if (p->back_refl == 1) {
			tmp[j] = rng_next(c) * 2 * delta0 - delta0;
			tmp[j] += c->self_rates[j];
			if (tmp[j] <= 0.000) tmp[j] = 0.000 - tmp[j];
			else if (tmp[j] >= 1.000) tmp[j] = 1.000 - (tmp[j] - 1.000);
		} else {
			tmp[j] = adpt_indp(c, &tem_stat[j], c->state[j]);
		}
		mhratio = log_ld_F_total(c, tmp) - log_ld_F_total(c, c->self_rates);
		if (p->back_refl == 0) mhratio *= hastings_stat(tem_stat, c->state, K);
		if (rng_next(c) < m_exp(c, MIN2(1, mhratio))) { /* sic: MIN2(1, .) on the log ratio (mcmc.c:1040) */
			c->self_rates[j] = tmp[j];
			if (p->back_refl == 0) c->state[j] = tem_stat[j];
		}
	}
	free(tmp);
	free(tem_stat);
}

/* dgeom (mcmc.c:1596-1604) */
static double dgeom(const orc_chain *c, double self, int gen) { return m_pow(c, self, (double)(gen - 1)) * (1 - self); }

void orc_update_S_IND(orc_chain *c) /* mcmc.c:864-884 (mode 3, uniform prior) */
{
	const orc_params *p = &c->p;
	double tmp, delta0 = 0.05, mhratio;
	int j;
	for (j = 0; j < p->N; j++) {
		if (keyed(c)) rng_seek(c, iter_base(c) + c->ky.offA + 4 + 2 * (uint64_t)j);
		tmp = rng_next(c) * 2 * delta0 - delta0;
		tmp += c->self_rates[j];
		if (tmp <= 0.0) tmp = 0.0 - tmp;
		if (tmp >= 1.0) tmp = 1.0 - (tmp - 1);
		mhratio = m_exp(c, m_log(c, dgeom(c, tmp, c->generation[j])) - m_log(c, dgeom(c, c->self_rates[j], c->generation[j])));
		c->self_rates[j] = (rng_next(c) < MIN2(1, mhratio)) ? tmp : c->self_rates[j];
	}
}

void orc_update_F_IND(orc_chain *c) /* mcmc.c:888-910 (mode 5, uniform prior; the coefficients live in self_rates[N]) */
{
	const orc_params *p = &c->p;
	double tmp, delta0 = 0.05, mhratio;
	int j;
	for (j = 0; j < p->N; j++) {
		if (keyed(c)) rng_seek(c, iter_base(c) + c->ky.offG + 2 * (uint64_t)j);
		tmp = rng_next(c) * 2 * delta0 - delta0;
		tmp += c->self_rates[j];
		if (tmp <= 0.0) tmp = 0.0 - tmp;
		if (tmp >= 1.0) tmp = 1.0 - (tmp - 1);
		mhratio = m_exp(c, log_ld_F_indv(c, tmp, j) - log_ld_F_indv(c, c->self_rates[j], j));
		c->self_rates[j] = (rng_next(c) < MIN2(1, mhratio)) ? tmp : c->self_rates[j];
	}
}

/* log_ld_indv_K (mcmc.c:1893-1914): individual i entirely in cluster k */
static double log_ld_indv_K(const orc_chain *c, int i, int k)
{
	int j;
	summer2 t;
	sum2_init(&t, c->p.accum);
	for (j = 0; j < c->p.L; j++) {
		int a0, a1;
		if (!c->valid[(long)i * c->p.L + j]) continue;
		a0 = GENO(c, i, j, 0); a1 = GENO(c, i, j, 1);
		sum2_add(&t, m_log(c, FREQ(c, k, j, a0)));
		sum2_add(&t, m_log(c, FREQ(c, k, j, a1)));
		if (a0 != a1) sum2_add(&t, m_log(c, 2));
	}
	return sum2_val(&t);
}

/* update_Z (mcmc.c:1094-1120), mode 0: the whole individual goes to one cluster.  zz[i] is kept in generation[i]
 * (mode 0 has no generations) and mirrored into every allele copy of z so that the allele counts of update_P
 * (mcmc.c:825-829) are the ordinary ones. */
void orc_update_Z(orc_chain *c, int init_flag)
{
	const orc_params *p = &c->p;
	const int K = p->K;
	int i, j, m, zz;
	double *tmp = malloc(sizeof(double) * K), temp = 0;
	for (i = 0; i < p->N; i++) {
		if (keyed(c)) rng_seek(c, init_flag ? 1 + 2 * (uint64_t)i : iter_base(c) + c->ky.offG + 2 * (uint64_t)i);
		for (m = 0; m < K; m++) {
			if (init_flag == 1) tmp[m] = (double)(m + 1) / K;
			else {
				tmp[m] = log_ld_indv_K(c, i, m);
				if (m == 0) temp = tmp[m];
				tmp[m] = m_exp(c, tmp[m] - temp);
				if (m >= 1) tmp[m] += tmp[m - 1];
			}
		}
		zz = disc_unif(c, tmp, K);
		c->generation[i] = zz;
		for (j = 0; j < p->L; j++)
			if (c->valid[(long)i * p->L + j]) { ZZ(c, i, j, 0) = zz; ZZ(c, i, j, 1) = zz; }
	}
	free(tmp);
}

void orc_update_G(orc_chain *c) /* mcmc.c:1053-1091 */
{
	const orc_params *p = &c->p;
	int i, j, stat, gen = 0;
	double selfing, mhratio;
	for (i = 0; i < p->N; i++) {
		selfing = 0;
		if (p->mode == 3) selfing = c->self_rates[i]; /* mcmc.c:1069-1070 */
		else for (j = 0; j < p->K; j++) selfing += c->qq[(long)i * p->K + j] * c->self_rates[j];
		if (keyed(c)) rng_seek(c, iter_base(c) + c->ky.offG + 2 * (uint64_t)i);
		stat = dt_stat(c, selfing);
		if (stat == 1) {
			gen = rgeom(c, 1 - selfing);
			if (gen < 1) gen = 1;
			if (gen > 50) gen = 50;
		} else if (stat == 0) gen = 1;
		else gen = 50;
		mhratio = m_exp(c, log_ld_indv(c, gen, i) - log_ld_indv(c, c->generation[i], i));
		if (rng_next(c) < MIN2(1, mhratio)) c->generation[i] = gen;
	}
}

void orc_update_ZQ(orc_chain *c, int init_flag) /* mcmc.c:1122-1203 */
{
	const orc_params *p = &c->p;
	const int K = p->K;
	int i, j, k, m;
	double *tmp = malloc(sizeof(double) * K);
	for (i = 0; i < p->N; i++) {
		if (keyed(c)) rng_seek(c, init_flag ? c->ky.ZI0 + (uint64_t)i * c->ky.SZ : iter_base(c) + c->ky.offZ + (uint64_t)i * c->ky.SZ);
		for (j = 0; j < p->L; j++) {
			if (!c->valid[(long)i * p->L + j]) continue;
			for (k = 0; k < p->P; k++) {
				for (m = 0; m < K; m++) {
					if (init_flag == 1) tmp[m] = (double)(m + 1) / K;
					else {
						tmp[m] = c->qq[(long)i * K + m] * FREQ(c, m, j, GENO(c, i, j, k));
						if (m >= 1) tmp[m] += tmp[m - 1];
					}
				}
				ZZ(c, i, j, k) = disc_unif(c, tmp, K);
			}
		}
		for (m = 0; m < K; m++) c->qqnum[(long)i * K + m] = 0.0;
		for (j = 0; j < p->L; j++)
			if (c->valid[(long)i * p->L + j])
				for (k = 0; k < p->P; k++) c->qqnum[(long)i * K + ZZ(c, i, j, k)] += 1.0;
		for (k = 0; k < K; k++) tmp[k] = c->qqnum[(long)i * K + k];
		rdirich(c, tmp, K, &c->qq[(long)i * K], c->alpha);
	}
	free(tmp);
}

void orc_update_alpha(orc_chain *c) /* mcmc.c:1244-1263 */
{
	const orc_params *p = &c->p;
	double mhratio = 1.0, ralpha;
	int i, m;
	if (keyed(c)) rng_seek(c, iter_base(c) + c->ky.offA);
	ralpha = rnormal(c, c->alpha, 1.0);
	if (ralpha > 0) {
		for (i = 0; i < p->N; i++)
			for (m = 0; m < p->K; m++) {
				double q = c->qq[(long)i * p->K + m], n = c->qqnum[(long)i * p->K + m];
				mhratio *= m_pow(c, q, ralpha + n) / m_pow(c, q, n + c->alpha);
			}
		c->alpha = (rng_next(c) < MIN2(1, mhratio)) ? ralpha : c->alpha;
	}
}

void orc_cal_lkh(orc_chain *c) /* mcmc.c:1916-1942 */
{
	int i;
	summer t;
	sum_init(&t, c->p.accum);
	for (i = 0; i < c->p.N; i++) {
		c->indvlkh[i] = (c->p.mode == 0) ? log_ld_indv_K(c, i, c->generation[i]) : (c->p.mode == 5) ? log_ld_F_indv(c, c->self_rates[i], i) : (c->p.mode == 4) ? log_ld_F_pop(c, c->self_rates, i)
						 : log_ld_indv(c, (c->p.mode == 2 || c->p.mode == 3) ? c->generation[i] : -1, i);
		sum_add(&t, c->indvlkh[i]);
	}
	c->totallkh = sum_val(&t);
}

void orc_iteration(orc_chain *c) /* mcmc.c:210-215 / 152-155 */
{
	orc_update_P(c);
	if (c->p.mode == 0) { /* mcmc.c:113-115 */
		orc_update_Z(c, 0);
		orc_cal_lkh(c);
		c->iter++;
		return;
	}
	if (c->p.mode == 2) {
		orc_update_S_POP(c);
		orc_update_G(c);
	}
	if (c->p.mode == 4) orc_update_F_POP(c);
	if (c->p.mode == 3) {
		orc_update_S_IND(c);
		orc_update_G(c);
	}
	if (c->p.mode == 5) orc_update_F_IND(c);
	orc_update_ZQ(c, 0);
	orc_update_alpha(c);
	orc_cal_lkh(c);
	c->iter++;
}

/* chain set-up in three stages (0: alpha, 1: generations + selfing rates, 2: ZQ init) so that the
 * dump tool can print state in between; mcmc.c:471-487, 193-206 */
void orc_chain_init_stage(orc_chain *c, const float *initd_row, int stage)
{
	const orc_params *p = &c->p;
	int i;
	if (stage == 0) {
		c->rng.o1 = c->rng.s1; c->rng.o2 = c->rng.s2; c->rng.o3 = c->rng.s3;
		c->iter = 0;
		if (p->mode != 0) c->alpha = rng_next(c) * 10; /* mcmc_POP_no_admixture does not go through initial_chn */
	} else if (stage == 1) {
		if (p->mode == 2) {
			for (i = 0; i < p->N; i++) {
				double pr = rng_next(c); /* argument evaluated before rgeom's own draw */
				c->generation[i] = rgeom(c, pr);
				if (c->generation[i] > 50) c->generation[i] = 50;
			}
			for (i = 0; i < p->K; i++) {
				c->self_rates[i] = initd_row[i];
				if (p->back_refl == 0) c->state[i] = dt_stat(c, c->self_rates[i]);
			}
		}
		if (p->mode == 3) { /* mcmc_INDV_selfing, mcmc.c:324-331; sic: the generations are not clamped to 50 here */
			if (keyed(c)) {
				for (i = 0; i < p->N; i++) {
					rng_seek(c, 1 + 2 * (uint64_t)i);
					c->self_rates[i] = rng_next(c);
					c->generation[i] = rgeom(c, 1 - c->self_rates[i]);
				}
			} else {
				for (i = 0; i < p->N; i++) c->self_rates[i] = rng_next(c);
				for (i = 0; i < p->N; i++) c->generation[i] = rgeom(c, 1 - c->self_rates[i]);
			}
		}
		if (p->mode == 5) /* mcmc_INDV_inbreedcoff, mcmc.c:412-415 */
			for (i = 0; i < p->N; i++) {
				if (keyed(c)) rng_seek(c, 1 + 2 * (uint64_t)i);
				c->self_rates[i] = rng_next(c);
			}
		if (p->mode == 4) /* mcmc_POP_inbreedcoff, mcmc.c:255-259: no generations */
			for (i = 0; i < p->K; i++) {
				c->self_rates[i] = initd_row[i];
				if (p->back_refl == 0) c->state[i] = dt_stat(c, c->self_rates[i]);
			}
	} else if (p->mode == 0) {
		orc_update_Z(c, 1);
	} else {
		orc_update_ZQ(c, 1);
	}
}

void orc_chain_init(orc_chain *c, const float *initd_row)
{
	orc_chain_init_stage(c, initd_row, 0);
	orc_chain_init_stage(c, initd_row, 1);
	orc_chain_init_stage(c, initd_row, 2);
}
void orc_iter_advance(orc_chain *c) { c->iter++; }

/* ------------------------------------------------------------------ object */
orc_chain *orc_create(const orc_params *p, const int *allelenum, const int *geno, const int *miss)
{
	orc_chain *c = calloc(1, sizeof(*c));
	long i, j, nl = (long)p->N * p->L;
	c->p = *p;
	c->allelenum = malloc(sizeof(int) * p->L);
	memcpy(c->allelenum, allelenum, sizeof(int) * p->L);
	c->geno = malloc(sizeof(int) * nl * p->P);
	memcpy(c->geno, geno, sizeof(int) * nl * p->P);
	c->valid = malloc(sizeof(int) * nl);
	for (i = 0; i < p->N; i++)
		for (j = 0; j < p->L; j++) c->valid[i * p->L + j] = (miss[i * p->L + j] != 1 && allelenum[j] > 1);
	c->z = calloc(nl * p->P, sizeof(int));
	c->generation = calloc(p->N, sizeof(int));
	c->state = calloc(p->K, sizeof(int));
	c->freq = calloc((size_t)p->K * p->L * p->Amax, sizeof(double));
	c->qq = calloc((size_t)p->N * p->K, sizeof(double));
	c->qqnum = calloc((size_t)p->N * p->K, sizeof(double));
	c->self_rates = calloc(p->K > p->N ? p->K : p->N, sizeof(double)); /* [K]; mode 3: one per individual */
	c->indvlkh = calloc(p->N, sizeof(double));
	c->rng.s1 = 13; c->rng.s2 = 4; c->rng.s3 = 1972; /* random.c:10-12 */
	keyed_layout(c);
	return c;
}
void orc_destroy(orc_chain *c)
{
	if (!c) return;
	free(c->allelenum); free(c->geno); free(c->valid); free(c->z); free(c->generation); free(c->state);
	free(c->freq); free(c->qq); free(c->qqnum); free(c->self_rates); free(c->indvlkh); free(c);
}
int *orc_z(orc_chain *c) { return c->z; }
double *orc_freq(orc_chain *c) { return c->freq; }
double *orc_qq(orc_chain *c) { return c->qq; }
double *orc_qqnum(orc_chain *c) { return c->qqnum; }
int *orc_generation(orc_chain *c) { return c->generation; }
double *orc_self_rates(orc_chain *c) { return c->self_rates; }
int *orc_state(orc_chain *c) { return c->state; }
double *orc_indvlkh(orc_chain *c) { return c->indvlkh; }
double orc_alpha(const orc_chain *c) { return c->alpha; }
void orc_set_alpha(orc_chain *c, double a) { c->alpha = a; }
double orc_totallkh(const orc_chain *c) { return c->totallkh; }
const int *orc_valid(const orc_chain *c) { return c->valid; }
int orc_error(const orc_chain *c) { return c->err; }

/* ------------------------------------------------------------------ CHAIN accumulation */
static void runmean(double *m, double x, long step) /* the update rule of mcmc.c:1327-1332 etc. */
{
	if (*m != 0) *m = *m * ((step + x / *m) / (1 + step));
	else *m = x / (1 + step);
}

static void store_chn(orc_chain *c, orc_result *r) /* mcmc.c:1320-1456 */
{
	const orc_params *p = &c->p;
	long i, n;
	runmean(&r->totallkh, c->totallkh, r->step);
	runmean(&r->totallkh2, c->totallkh * c->totallkh, r->step);
	for (i = 0; i < p->N; i++) runmean(&r->indvlkh[i], c->indvlkh[i], r->step);
	for (i = 0; i < (long)p->N * p->K; i++) {
		runmean(&r->qq[i], c->qq[i], r->step);
		runmean(&r->qq2[i], c->qq[i] * c->qq[i], r->step);
	}
	if (p->mode == 5)
		for (i = 0; i < p->N; i++) {
			runmean(&r->self_rates[i], c->self_rates[i], r->step);
			runmean(&r->self_rates2[i], c->self_rates[i] * c->self_rates[i], r->step);
		}
	if (p->mode == 3) {
		for (i = 0; i < p->N; i++) {
			runmean(&r->self_rates[i], c->self_rates[i], r->step);
			runmean(&r->self_rates2[i], c->self_rates[i] * c->self_rates[i], r->step);
			runmean(&r->gen[i], c->generation[i], r->step);
			runmean(&r->gen2[i], c->generation[i] * c->generation[i], r->step);
		}
	}
	if (p->mode == 4)
		for (i = 0; i < p->K; i++) {
			runmean(&r->self_rates[i], c->self_rates[i], r->step);
			runmean(&r->self_rates2[i], c->self_rates[i] * c->self_rates[i], r->step);
		}
	if (p->mode == 2) {
		for (i = 0; i < p->K; i++) {
			runmean(&r->self_rates[i], c->self_rates[i], r->step);
			runmean(&r->self_rates2[i], c->self_rates[i] * c->self_rates[i], r->step);
		}
		for (i = 0; i < p->N; i++) {
			runmean(&r->gen[i], c->generation[i], r->step);
			runmean(&r->gen2[i], c->generation[i] * c->generation[i], r->step);
		}
	}
	if (p->print_freq == 1) {
		n = (long)p->K * p->L * p->Amax;
		for (i = 0; i < n; i++) {
			long a = i % p->Amax, j = (i / p->Amax) % p->L;
			if (a >= c->allelenum[j]) continue;
			runmean(&r->freq[i], c->freq[i], r->step);
			runmean(&r->freq2[i], c->freq[i] * c->freq[i], r->step);
		}
	}
	r->step++;
}

static double *ones(long n)
{
	double *v = malloc(sizeof(double) * (n > 0 ? n : 1));
	long i;
	for (i = 0; i < n; i++) v[i] = 1;
	return v;
}

static void allocate_chn(orc_chain *c, orc_result *r) /* mcmc.c:588-642, 644-738 (all running means start at 1) */
{
	const orc_params *p = &c->p;
	r->step = 0;
	r->totallkh = 1;
	r->totallkh2 = 1;
	r->indvlkh = ones(p->N);
	r->qq = ones((long)p->N * p->K);
	r->qq2 = ones((long)p->N * p->K);
	r->self_rates = ones(p->K > p->N ? p->K : p->N); r->self_rates2 = ones(p->K > p->N ? p->K : p->N);
	r->gen = ones(p->N); r->gen2 = ones(p->N);
	r->freq = ones((long)p->K * p->L * p->Amax); r->freq2 = ones((long)p->K * p->L * p->Amax);
}

void orc_store_chn(orc_chain *c, orc_result *r, int allocate)
{
	if (allocate) allocate_chn(c, r); else store_chn(c, r);
}

void orc_result_free(orc_result *r)
{
	free(r->indvlkh); free(r->qq); free(r->qq2); free(r->self_rates); free(r->self_rates2);
	free(r->gen); free(r->gen2); free(r->freq); free(r->freq2);
	memset(r, 0, sizeof(*r));
}

int orc_check_empty_cluster(const orc_chain *c);
static int check_empty_cluster(const orc_chain *c) { return orc_check_empty_cluster(c); }
int orc_check_empty_cluster(const orc_chain *c) /* mcmc.c:1944-1974 */
{
	int k, j;
	for (k = 0; k < c->p.K; k++) {
		double sum = 0;
		for (j = 0; j < c->p.N; j++) sum += c->qq[(long)j * c->p.K + k];
		if (sum < 0.01) return 1;
	}
	return 0;
}

void orc_run_chain(orc_chain *c, const float *initd_row, long update, long burnin, int thinning, int ckrep,
		   double *convg, orc_result *res) /* mcmc.c:182-239 */
{
	long cnt_step = 0, step;
	memset(res, 0, sizeof(*res));
	res->steps = (long)((update - burnin) / thinning);
	orc_chain_init(c, initd_row);
	for (step = 0; step < update; step++) {
		orc_iteration(c);
		if (step == burnin - 1) allocate_chn(c, res);
		if (step >= burnin && (step + 1 - burnin) % thinning == 0) {
			store_chn(c, res);
			if (convg && cnt_step < ckrep) convg[cnt_step] = c->totallkh;
			cnt_step++;
		}
		if (cnt_step == c->p.nstep_check_empty_cluster) {
			if ((res->flag_empty_cluster = check_empty_cluster(c)) == 1) break;
		}
	}
	if (keyed(c)) { /* leave the stream at a position that does not depend on consumption */
		rng_seek(c, c->ky.B0 + (uint64_t)update * c->ky.BLK);
	}
}

double orc_gelman_rubin(const double *vec, int numchains, int totrep) /* check_converg.c:100-153 */
{
	double *psii = malloc(sizeof(double) * numchains), *S = malloc(sizeof(double) * numchains);
	double psi = 0, W = 0, B = 0, V;
	int i, j, rep = totrep / numchains; /* sic: the reference divides ckrep by the chain count */
	for (i = 0; i < numchains; i++) {
		psii[i] = 0;
		for (j = 0; j < rep; j++) psii[i] += vec[i * rep + j];
		psii[i] = psii[i] / rep;
		psi = psi + psii[i];
	}
	psi = psi / numchains;
	for (i = 0; i < numchains; i++) {
		S[i] = 0;
		for (j = 0; j < rep; j++) S[i] += (vec[i * rep + j] - psii[i]) * (vec[i * rep + j] - psii[i]);
		S[i] = S[i] / (rep - 1);
		W += S[i];
	}
	W = W / numchains;
	for (i = 0; i < numchains; i++) B += (psii[i] - psi) * (psii[i] - psi);
	B = (B * rep) / (numchains - 1);
	V = (W * (rep - 1)) / rep + B / rep;
	free(psii);
	free(S);
	return V / W;
}

/* ------------------------------------------------------------------ unit entry points */
double orc_rgamma(orc_chain *c, double a) { return rgamma(c, a); }
int orc_rgeom(orc_chain *c, double p) { return rgeom(c, p); }
double orc_rnormal(orc_chain *c, double mean, double sd) { return rnormal(c, mean, sd); }
void orc_rdirich(orc_chain *c, const double *alpha, int n, double *out, double add) { rdirich(c, alpha, n, out, add); }
int orc_disc_unif(orc_chain *c, double *vec, int n) { return disc_unif(c, vec, n); }
double orc_genofreq(orc_chain *c, int a0, int a1, double f0, double f1, int gen) { return genofreq(c, a0, a1, f0, f1, gen); }

/* ------------------------------------------------------------------ text reader */
int orc_read_text_diploid(const char *path, int *Nout, int *Lout, int **allelenum, int **geno, int **miss)
{
	FILE *f = fopen(path, "r");
	char *line = NULL, **tok = NULL;
	size_t cap = 0;
	long nrows = 0, ntok = 0, L0 = -1, captok = 0, i, j, k;
	int N, L = 0, *an, *g, *ms;
	if (!f) return -1;
	while (getline(&line, &cap, f) > 0) {
		char *s = line;
		long cnt = 0;
		while (*s) {
			while (isspace((unsigned char)*s)) s++;
			if (!*s) break;
			{
				char *b = s;
				while (*s && !isspace((unsigned char)*s)) s++;
				if (ntok == captok) { captok = captok ? captok * 2 : 4096; tok = realloc(tok, sizeof(char *) * captok); }
				tok[ntok] = strndup(b, (size_t)(s - b));
				ntok++; cnt++;
			}
		}
		if (cnt == 0) continue;
		if (L0 < 0) L0 = cnt;
		else if (cnt != L0) { fclose(f); return -2; }
		nrows++;
	}
	fclose(f);
	free(line);
	N = (int)(nrows / 2);
	an = malloc(sizeof(int) * L0);
	g = malloc(sizeof(int) * (size_t)N * L0 * 2);
	/* transform_data: codes in order of first appearance scanning individuals, then copies */
	for (j = 0; j < L0; j++) {
		char **types = malloc(sizeof(char *) * (size_t)N * 2);
		int cnt = 0, m;
		for (i = 0; i < N; i++)
			for (k = 0; k < 2; k++) {
				const char *t = tok[(i * 2 + k) * L0 + j];
				if (strcmp(t, "-9") == 0) continue;
				for (m = 0; m < cnt; m++) if (strcmp(types[m], t) == 0) break;
				if (m == cnt) types[cnt++] = (char *)t;
			}
		if (cnt >= 2) {
			an[L] = cnt;
			for (i = 0; i < N; i++)
				for (k = 0; k < 2; k++) {
					const char *t = tok[(i * 2 + k) * L0 + j];
					int code = -9;
					if (strcmp(t, "-9") != 0)
						for (m = 0; m < cnt; m++) if (strcmp(types[m], t) == 0) code = m;
					g[((long)i * L0 + L) * 2 + k] = code;
				}
			L++;
		}
		free(types);
	}
	/* compact to [N][L][2] and derive missindx (get_missing) */
	*geno = malloc(sizeof(int) * (size_t)N * L * 2);
	ms = malloc(sizeof(int) * (size_t)N * L);
	for (i = 0; i < N; i++)
		for (j = 0; j < L; j++) {
			int a0 = g[((long)i * L0 + j) * 2], a1 = g[((long)i * L0 + j) * 2 + 1];
			(*geno)[((long)i * L + j) * 2] = a0;
			(*geno)[((long)i * L + j) * 2 + 1] = a1;
			ms[i * L + j] = (a0 == -9 || a1 == -9);
		}
	for (i = 0; i < ntok; i++) free(tok[i]);
	free(tok);
	free(g);
	*allelenum = an;
	*miss = ms;
	*Nout = N;
	*Lout = L;
	return 0;
}

/* ------------------------------------------------------------------ isg_math.h test hooks */
double orc_isg_log(double x) { return isg_log(x); }
double orc_isg_exp(double x) { return isg_exp(x); }
double orc_isg_pow(double x, double y) { return isg_pow(x, y); }
double orc_isg_cos(double x) { return isg_cos(x); }
double orc_isg_accsum(const double *v, long n)
{
	isg_acc a;
	long i;
	isg_acc_zero(&a);
	for (i = 0; i < n; i++) isg_acc_add(&a, v[i]);
	return isg_acc_value(&a);
}

/* FNV-64 of int32 / double arrays, as used in the golden files (oracle/dump_fmt.h) */
uint64_t orc_fnv_i32(const int *v, long n)
{
	uint64_t h = 0xcbf29ce484222325ULL;
	long i;
	int b;
	for (i = 0; i < n; i++)
		for (b = 0; b < 4; b++) { h ^= (unsigned char)(((uint32_t)v[i]) >> (8 * b)); h *= 0x100000001b3ULL; }
	return h;
}
uint64_t orc_fnv_f64(const double *v, long n)
{
	uint64_t h = 0xcbf29ce484222325ULL;
	long i;
	int b;
	for (i = 0; i < n; i++) {
		uint64_t u = isg_d2u(v[i]);
		for (b = 0; b < 8; b++) { h ^= (unsigned char)(u >> (8 * b)); h *= 0x100000001b3ULL; }
	}
	return h;
}
