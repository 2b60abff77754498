/*
 * isg_math.h -- canonical, bit-reproducible double-precision math for the InStruct hot path.
 *
 * The reference (mcmc.c / random.c) calls libm's log/exp/pow/sqrt/cos.  glibc's and ROCm-ocml's
 * implementations differ in the last ulp, which would make "CPU vs MI355X" comparisons fuzzy.
 * Every function here is written in plain IEEE-754 double operations (+,-,*,/, sqrt and explicit
 * fma, all correctly rounded on x86-64 and on gfx950 when compiled with -ffp-contract=off), so
 * the SAME bits come out on the host (gcc) and on the device (hipcc).  The CPU oracle's
 * "canonical" mode and the HIP kernels both include this header; the oracle's "reference" mode
 * uses glibc instead and is the one pinned bit-for-bit to the real reference (tests/golden).
 *
 * Algorithms: classic table-free argument reduction + minimax polynomials (the well-known
 * fdlibm-style kernels for log/exp/sin/cos); pow = exp(y*log(x)) with the logarithm carried in
 * double-double.  Accuracy (measured in tests/test_isg_math.py against glibc): log/exp <= 1 ulp,
 * pow <= 2 ulp over the ranges the sampler uses, cos <= 2 ulp on [0, 2*pi].
 *
 * Call sites being replaced (reference file:line):
 *   log  : random.c:128,178,222,294,319  mcmc.c:1645,1747,1758,1763,1766
 *   exp  : random.c:188                  mcmc.c:964,1085
 *   pow  : random.c:179,187              mcmc.c:1258,1645,1690,1700
 *   sqrt : random.c:208,294              cos : random.c:295
 */
#ifndef ISG_MATH_H
#define ISG_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define ISG_HD __host__ __device__ static inline
#else
#define ISG_HD static inline
#endif

ISG_HD uint64_t isg_d2u(double x)
{
	union { double d; uint64_t u; } c;
	c.d = x;
	return c.u;
}
ISG_HD double isg_u2d(uint64_t u)
{
	union { double d; uint64_t u; } c;
	c.u = u;
	return c.d;
}
ISG_HD double isg_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
ISG_HD double isg_sqrt(double x) { return __builtin_sqrt(x); }
ISG_HD double isg_inf(void) { return isg_u2d(0x7ff0000000000000ULL); }
ISG_HD double isg_nan(void) { return isg_u2d(0x7ff8000000000000ULL); }
ISG_HD int isg_isnan(double x) { return (isg_d2u(x) & 0x7fffffffffffffffULL) > 0x7ff0000000000000ULL; }

/* 2^k as a double for -1022 <= k <= 1023 */
ISG_HD double isg_pow2i(int k) { return isg_u2d((uint64_t)(k + 1023) << 52); }

/* x * 2^k with correct handling of over/underflow (k may be far out of range) */
ISG_HD double isg_scalbn(double x, int k)
{
	if (k > 1023) {
		x *= isg_pow2i(1023);
		k -= 1023;
		if (k > 1023) {
			x *= isg_pow2i(1023);
			k -= 1023;
			if (k > 1023) k = 1023;
		}
	} else if (k < -1022) {
		x *= isg_pow2i(-969); /* 2^-1022 * 2^53 */
		k += 969;
		if (k < -1022) {
			x *= isg_pow2i(-969);
			k += 969;
			if (k < -1022) k = -1022;
		}
	}
	return x * isg_pow2i(k);
}

#define ISG_LN2_HI 6.93147180369123816490e-01 /* 0x3fe62e42fee00000: 32 significant bits */
#define ISG_LN2_LO 1.90821492927058770002e-10
#define ISG_INVLN2 1.44269504088896338700e+00
#define ISG_LG1 6.666666666666735130e-01
#define ISG_LG2 3.999999999940941908e-01
#define ISG_LG3 2.857142874366239149e-01
#define ISG_LG4 2.222219843214978396e-01
#define ISG_LG5 1.818357216161805012e-01
#define ISG_LG6 1.531383769920937332e-01
#define ISG_LG7 1.479819860511658591e-01

/* split positive finite x into m in [sqrt(2)/2, sqrt(2)) and k with x = m * 2^k */
ISG_HD double isg_split(double x, int *kout)
{
	uint64_t u = isg_d2u(x);
	int k = 0;
	uint32_t hx;
	if ((u >> 52) == 0) { /* subnormal */
		x *= 18014398509481984.0; /* 2^54 */
		u = isg_d2u(x);
		k = -54;
	}
	hx = (uint32_t)(u >> 32);
	k += (int)(hx >> 20) - 1023;
	hx &= 0x000fffff;
	if (hx >= 0x6a09e) { /* mantissa >= ~sqrt(2): use m/2 */
		u = (u & 0x000fffffffffffffULL) | 0x3fe0000000000000ULL;
		k += 1;
	} else {
		u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
	}
	*kout = k;
	return isg_u2d(u);
}

/* R(z) ~ (log(1+f) - 2s)/s with s = f/(2+f), z = s*s */
ISG_HD double isg_log_poly(double z)
{
	double w = z * z;
	double t1 = w * (ISG_LG2 + w * (ISG_LG4 + w * ISG_LG6));
	double t2 = z * (ISG_LG1 + w * (ISG_LG3 + w * (ISG_LG5 + w * ISG_LG7)));
	return t2 + t1;
}

ISG_HD double isg_log(double x)
{
	uint64_t u = isg_d2u(x);
	int k;
	double m, f, s, z, R, hfsq, dk;
	if ((u << 1) == 0) return -isg_inf();                 /* log(+-0) = -inf */
	if (u >> 63) return isg_nan();                        /* log(<0) */
	if ((u >> 52) == 0x7ff) return x;                     /* inf or nan */
	m = isg_split(x, &k);
	f = m - 1.0;
	dk = (double)k;
	if (f == 0.0) return dk * ISG_LN2_HI + dk * ISG_LN2_LO;
	s = f / (2.0 + f);
	z = s * s;
	R = isg_log_poly(z);
	hfsq = 0.5 * f * f;
	return dk * ISG_LN2_HI - ((hfsq - (s * (hfsq + R) + dk * ISG_LN2_LO)) - f);
}

#define ISG_EXP_P1 1.66666666666666019037e-01
#define ISG_EXP_P2 -2.77777777770155933842e-03
#define ISG_EXP_P3 6.61375632143793436117e-05
#define ISG_EXP_P4 -1.65339022054652515390e-06
#define ISG_EXP_P5 4.13813679705723846039e-08

/* exp(hi_in + lo_in) for |lo_in| << |hi_in|; core shared by isg_exp and isg_pow */
ISG_HD double isg_exp2part(double x, double xlo)
{
	double kd, hi, lo, r, t, c, y;
	int k;
	if (isg_isnan(x)) return x;
	if (x > 7.09782712893383973096e+02) return isg_inf();
	if (x < -7.45133219101941108420e+02) return 0.0;
	kd = __builtin_rint(x * ISG_INVLN2);
	k = (int)kd;
	hi = x - kd * ISG_LN2_HI;
	lo = kd * ISG_LN2_LO - xlo;
	r = hi - lo;
	t = r * r;
	c = r - t * (ISG_EXP_P1 + t * (ISG_EXP_P2 + t * (ISG_EXP_P3 + t * (ISG_EXP_P4 + t * ISG_EXP_P5))));
	y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
	return isg_scalbn(y, k);
}

ISG_HD double isg_exp(double x) { return isg_exp2part(x, 0.0); }

/* pow for the sampler's domain: x >= 0 (x < 0 -> NaN), any finite y */
ISG_HD double isg_pow(double x, double y)
{
	int k;
	double m, f, dh, dl, sh, sl, r, z, R, tail, dk, a, b, bb, lh, ll, ph, pl, t;
	uint64_t ux = isg_d2u(x);
	if (y == 0.0) return 1.0;
	if (isg_isnan(x) || isg_isnan(y)) return isg_nan();
	if (x == 1.0) return 1.0;
	if ((ux << 1) == 0) return (y > 0.0) ? 0.0 : isg_inf();
	if (ux >> 63) return isg_nan();
	if ((ux >> 52) == 0x7ff) return (y > 0.0) ? isg_inf() : 0.0;
	if ((isg_d2u(y) & 0x7fffffffffffffffULL) == 0x7ff0000000000000ULL) {
		int big = x > 1.0;
		return ((y > 0.0) == big) ? isg_inf() : 0.0;
	}
	m = isg_split(x, &k);
	f = m - 1.0;                 /* exact */
	dk = (double)k;
	/* den = m + 1 as double-double (dh + dl) */
	dh = m + 1.0;
	dl = m - (dh - 1.0);
	/* s = f / den as double-double (sh + sl) */
	sh = f / dh;
	r = isg_fma(-sh, dh, f);
	r = r - sh * dl;
	sl = r / dh;
	z = sh * sh;
	R = isg_log_poly(z);
	tail = 2.0 * sl + sh * R + dk * ISG_LN2_LO;
	/* (dk*LN2_HI) + 2*sh : two-sum */
	a = dk * ISG_LN2_HI; /* exact: 32-bit constant times |k| < 2^11 */
	b = 2.0 * sh;
	t = a + b;
	bb = t - a;
	ll = ((a - (t - bb)) + (b - bb)) + tail;
	lh = t + ll;
	ll = ll - (lh - t);
	/* y * (lh + ll) */
	ph = y * lh;
	pl = isg_fma(y, lh, -ph) + y * ll;
	return isg_exp2part(ph, pl);
}

#define ISG_PIO2_1 1.57079632673412561417e+00  /* first 33 bits of pi/2 */
#define ISG_PIO2_1T 6.07710050650619224932e-11 /* pi/2 - PIO2_1 */

/* cos(x) for 0 <= x <= ~7 (Box-Muller angle, random.c:293-295) */
ISG_HD double isg_cos(double x)
{
	double nd = __builtin_rint(x * 6.36619772367581382433e-01); /* 2/pi */
	int n = (int)nd;
	double r = (x - nd * ISG_PIO2_1) - nd * ISG_PIO2_1T;
	double z = r * r, c, s;
	c = 1.0 - (0.5 * z - z * z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 +
		z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 +
		z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))))));
	s = r + r * z * (-1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 +
		z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
		z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
	switch (n & 3) {
	case 0: return c;
	case 1: return -s;
	case 2: return -c;
	default: return s;
	}
}

/*
 * Order-independent accumulation of doubles: 128-bit two's-complement fixed point with 2^-80
 * resolution (terms are truncated toward zero to that grid).  Integer addition is associative,
 * so any reduction tree on the GPU and the sequential loop on the CPU give identical results.
 * Range: |sum| < 2^46.  Non-finite terms are tracked in flags (1: +inf, 2: -inf, 4: nan).
 * Replaces the sequential `temp += log(...)` sums of mcmc.c:1735-1770, 1638-1646, 1940.
 */
typedef struct {
	uint64_t lo;
	uint64_t hi; /* two's complement high word */
	uint32_t flags;
} isg_acc;

ISG_HD void isg_acc_zero(isg_acc *a)
{
	a->lo = 0;
	a->hi = 0;
	a->flags = 0;
}
ISG_HD void isg_acc_add_raw(isg_acc *a, uint64_t lo, uint64_t hi)
{
	uint64_t l = a->lo + lo;
	a->hi = a->hi + hi + (l < lo ? 1u : 0u);
	a->lo = l;
}
ISG_HD void isg_acc_merge(isg_acc *a, const isg_acc *b)
{
	isg_acc_add_raw(a, b->lo, b->hi);
	a->flags |= b->flags;
}
ISG_HD void isg_acc_add(isg_acc *a, double v)
{
	uint64_t u = isg_d2u(v), mant, lo, hi;
	int e = (int)((u >> 52) & 0x7ff), sh;
	if (e == 0x7ff) {
		a->flags |= (u & 0x000fffffffffffffULL) ? 4u : ((u >> 63) ? 2u : 1u);
		return;
	}
	mant = u & 0x000fffffffffffffULL;
	if (e == 0) e = 1; else mant |= 0x0010000000000000ULL;
	/* value = mant * 2^(e-1075); fixed = value * 2^80 = mant * 2^(e-995) */
	sh = e - 995;
	if (sh >= 0) {
		if (sh >= 75) { a->flags |= 4u; return; } /* |v| >= 2^127-ish: outside the supported range */
		if (sh >= 64) { lo = 0; hi = mant << (sh - 64); }
		else if (sh == 0) { lo = mant; hi = 0; }
		else { lo = mant << sh; hi = mant >> (64 - sh); }
	} else {
		sh = -sh;
		lo = (sh >= 64) ? 0 : (mant >> sh);
		hi = 0;
	}
	if (u >> 63) { /* negate */
		lo = ~lo + 1;
		hi = ~hi + (lo == 0 ? 1u : 0u);
	}
	isg_acc_add_raw(a, lo, hi);
}
ISG_HD double isg_acc_value(const isg_acc *a)
{
	uint64_t lo = a->lo, hi = a->hi, top;
	int neg = (int)(hi >> 63), msb, shift;
	double r;
	if (a->flags & 4u) return isg_nan();
	if ((a->flags & 3u) == 3u) return isg_nan();
	if (a->flags & 1u) return isg_inf();
	if (a->flags & 2u) return -isg_inf();
	if (neg) {
		lo = ~lo + 1;
		hi = ~hi + (lo == 0 ? 1u : 0u);
	}
	if (hi == 0 && lo == 0) return 0.0;
	if (hi) msb = 127 - __builtin_clzll(hi); else msb = 63 - __builtin_clzll(lo);
	if (msb <= 52) {
		r = (double)lo; /* exact */
		shift = 0;
	} else {
		int rb, roundbit, sticky;
		shift = msb - 52; /* keep 53 bits; round to nearest even on the discarded ones */
		if (shift >= 64) top = hi >> (shift - 64);
		else top = (hi << (64 - shift)) | (lo >> shift);
		rb = shift - 1;
		if (rb >= 64) {
			roundbit = (int)((hi >> (rb - 64)) & 1);
			sticky = (lo != 0) || (rb > 64 && (hi & ((1ULL << (rb - 64)) - 1)) != 0);
		} else {
			roundbit = (int)((lo >> rb) & 1);
			sticky = rb > 0 && (lo & ((1ULL << rb) - 1)) != 0;
		}
		if (roundbit && (sticky || (top & 1))) top++;
		r = (double)top; /* top <= 2^53: exact */
	}
	r = isg_scalbn(r, shift - 80);
	return neg ? -r : r;
}

/*
 * Order-independent accumulation specialised for terms of magnitude < 1024 (single log-probabilities:
 * the per-locus terms of log_ld_indv, mcmc.c:1735-1770).  A term v is split exactly into
 *     v = h * 2^-20 + m * 2^-51 + r,   h = rint(v * 2^20),  m = rint((v - h * 2^-20) * 2^51),  |r| <= 2^-52
 * and the integers h, m are summed (associative); r is dropped.  A dozen instructions per term instead
 * of the generic isg_acc's variable shifts.  -inf / NaN / out-of-range terms are tracked in flags.
 */
typedef struct {
	int64_t hi, lo;
	uint32_t flags; /* 1: +inf, 2: -inf, 4: nan or |v| >= 1024 */
} isg_acc2;

ISG_HD void isg_acc2_zero(isg_acc2 *a)
{
	a->hi = 0;
	a->lo = 0;
	a->flags = 0;
}
ISG_HD void isg_acc2_add(isg_acc2 *a, double v)
{
	double hs, lo, ms;
	if (!(v > -1024.0 && v < 1024.0)) {
		uint64_t u = isg_d2u(v);
		a->flags |= (u == 0x7ff0000000000000ULL) ? 1u : (u == 0xfff0000000000000ULL) ? 2u : 4u;
		return;
	}
	hs = __builtin_rint(v * 1048576.0);
	lo = isg_fma(hs, -0x1p-20, v); /* exact */
	ms = __builtin_rint(lo * 0x1p51);
	a->hi += (int64_t)(int32_t)hs;
	a->lo += (int64_t)(int32_t)ms;
}
ISG_HD void isg_acc2_merge(isg_acc2 *a, const isg_acc2 *b)
{
	a->hi += b->hi;
	a->lo += b->lo;
	a->flags |= b->flags;
}
ISG_HD double isg_acc2_value(const isg_acc2 *a)
{
	if (a->flags & 4u) return isg_nan();
	if ((a->flags & 3u) == 3u) return isg_nan();
	if (a->flags & 1u) return isg_inf();
	if (a->flags & 2u) return -isg_inf();
	return (double)a->hi * 0x1p-20 + (double)a->lo * 0x1p-51;
}

#endif /* ISG_MATH_H */
