/*
 * isg_wh.h -- Wichmann-Hill generator as a random-access ("counter based") stream.
 *
 * The reference draws every uniform from one process-global stream (random.c:14-47):
 *     s1 = 171*s1 % 30269;  s2 = 172*s2 % 30307;  s3 = 170*s3 % 30323;
 *     u  = fmod(s1/30269.0 + s2/30307.0 + s3/30323.0, 1.0)
 * Each component is a multiplicative LCG modulo a prime, so the state n draws after a known
 * state is s_r * a_r^n mod m_r, and a_r^n only depends on n mod (m_r - 1).  That turns the
 * stream into a function position -> uniform which every GPU lane evaluates independently:
 * one two-level table lookup to land on its first position, then plain stepping.
 *
 * Host and device use the same code (the host tracks the stream position between kernels).
 */
#ifndef ISG_WH_H
#define ISG_WH_H
#include <stdint.h>
#include "isg_math.h"

#define ISG_M1 30269u
#define ISG_M2 30307u
#define ISG_M3 30323u
#define ISG_A1 171u
#define ISG_A2 172u
#define ISG_A3 170u

typedef struct {
	uint32_t s1, s2, s3;
} isg_wh;

/* a^lo (lo < 256) and a^(256*hi) (hi < 119) for the three generators; filled by isg_wh_tables_init */
typedef struct {
	uint16_t lo[3][256];
	uint16_t hi[3][120];
} isg_wh_tables;

static inline void isg_wh_tables_init(isg_wh_tables *t)
{
	static const uint32_t m[3] = {ISG_M1, ISG_M2, ISG_M3}, a[3] = {ISG_A1, ISG_A2, ISG_A3};
	int r, i;
	for (r = 0; r < 3; r++) {
		uint32_t v = 1, b;
		for (i = 0; i < 256; i++) { t->lo[r][i] = (uint16_t)v; v = v * a[r] % m[r]; }
		b = v; /* a^256 */
		v = 1;
		for (i = 0; i < 120; i++) { t->hi[r][i] = (uint16_t)v; v = v * b % m[r]; }
	}
}

/*
 * (a * s) mod m for s < m + 2000 (so a * s < 2^24) without integer division or 32-bit multiplies:
 * 24-bit multiply, quotient estimate through the float reciprocal, two corrections.  Equal to
 * (a * s) % m for every such s (exhaustive check: isg_selftest(), tests/test_wh_stream.py).
 * 24-bit multiplies and float conversions are full-rate VALU operations on CDNA.
 */
ISG_HD uint32_t isg_lcg_fast(uint32_t s, uint32_t a, uint32_t m, float invm)
{
#if defined(__HIP_DEVICE_COMPILE__)
	uint32_t p = __umul24(a, s);
	uint32_t q = (uint32_t)((float)p * invm);
	uint32_t r = p - __umul24(q, m); /* in {-m .. 2m-1} mod 2^32 */
	uint32_t r1 = r + m, r2;
	r = r < r1 ? r : r1;             /* a "negative" r is huge as unsigned: r + m wraps to the small value */
	r2 = r - m;
	return r < r2 ? r : r2;          /* r < m: r - m wraps to a huge value */
#else
	uint32_t p = a * s;
	uint32_t q = (uint32_t)((float)p * invm);
	uint32_t r = p - q * m;
	uint32_t r1 = r + m, r2;
	r = r < r1 ? r : r1;
	r2 = r - m;
	return r < r2 ? r : r2;
#endif
}

ISG_HD void isg_wh_step(isg_wh *s)
{
#if defined(__HIP_DEVICE_COMPILE__)
	s->s1 = isg_lcg_fast(s->s1 & 0xffffu, ISG_A1, ISG_M1, 1.0f / 30269.0f);
	s->s2 = isg_lcg_fast(s->s2 & 0xffffu, ISG_A2, ISG_M2, 1.0f / 30307.0f);
	s->s3 = isg_lcg_fast(s->s3 & 0xffffu, ISG_A3, ISG_M3, 1.0f / 30323.0f);
#else
	s->s1 = (ISG_A1 * s->s1) % ISG_M1;
	s->s2 = (ISG_A2 * s->s2) % ISG_M2;
	s->s3 = (ISG_A3 * s->s3) % ISG_M3;
#endif
}

/* the uniform belonging to the CURRENT state (call after isg_wh_step) */
/* s / m for an integer 0 <= s < m + 70000 and m one of the three moduli: one multiply by the
 * rounded reciprocal plus one fma-based correction step gives the correctly rounded quotient
 * (verified exhaustively for all such s against true division: tests/test_wh_stream.py) */
ISG_HD double isg_wh_div(uint32_t s, double m, double inv)
{
	double x = (double)s, q = x * inv;
	return isg_fma(isg_fma(-q, m, x), inv, q);
}

ISG_HD double isg_wh_value(const isg_wh *s)
{
	double x = isg_wh_div(s->s1, 30269.0, 1.0 / 30269.0) + isg_wh_div(s->s2, 30307.0, 1.0 / 30307.0) +
		   isg_wh_div(s->s3, 30323.0, 1.0 / 30323.0);
	/* fmod(x, 1.0) for 0 <= x < 3: both subtractions are exact */
	if (x >= 2.0) x -= 2.0;
	else if (x >= 1.0) x -= 1.0;
	return x;
}

/* single precision value of the same uniform, |error| < 1e-6: only good enough to pre-filter a
 * decision that is re-taken in double whenever it is close (callers also re-take it near 0 and 1,
 * where the fractional part could wrap differently) */
ISG_HD float isg_wh_value_f32(const isg_wh *s)
{
	float x = (float)s->s1 * (1.0f / 30269.0f) + (float)s->s2 * (1.0f / 30307.0f) + (float)s->s3 * (1.0f / 30323.0f);
	x = (x >= 2.0f) ? x - 2.0f : ((x >= 1.0f) ? x - 1.0f : x);
	return x;
}

ISG_HD double isg_wh_next(isg_wh *s)
{
	isg_wh_step(s);
	return isg_wh_value(s);
}

/* n mod m for n < 2^52 and a small constant m, without a 64-bit integer division (which a GPU lane emulates in a few
 * hundred instructions): quotient estimate in double (off by at most one), exact remainder by fma, one correction
 * either way.  Stream positions stay far below 2^52 (the generator's period is 6.95e12). */
ISG_HD uint32_t isg_mod_u64(uint64_t n, uint32_t m)
{
#if defined(__HIP_DEVICE_COMPILE__)
	const double x = (double)n, md = (double)m;
	const double q = __builtin_floor(x * (1.0 / md));
	double r = isg_fma(-q, md, x);
	r = (r < 0.0) ? r + md : r;
	r = (r >= md) ? r - md : r;
	return (uint32_t)r;
#else
	return (uint32_t)(n % m);
#endif
}

/* state n draws after `s` (n = 0 returns s itself, reduced) */
ISG_HD isg_wh isg_wh_jump(const isg_wh_tables *t, isg_wh s, uint64_t n)
{
	uint32_t e1 = isg_mod_u64(n, ISG_M1 - 1), e2 = isg_mod_u64(n, ISG_M2 - 1), e3 = isg_mod_u64(n, ISG_M3 - 1);
	uint32_t p1 = (uint32_t)t->lo[0][e1 & 255] * t->hi[0][e1 >> 8] % ISG_M1;
	uint32_t p2 = (uint32_t)t->lo[1][e2 & 255] * t->hi[1][e2 >> 8] % ISG_M2;
	uint32_t p3 = (uint32_t)t->lo[2][e3 & 255] * t->hi[2][e3 >> 8] % ISG_M3;
	isg_wh r;
	r.s1 = (s.s1 % ISG_M1) * p1 % ISG_M1;
	r.s2 = (s.s2 % ISG_M2) * p2 % ISG_M2;
	r.s3 = (s.s3 % ISG_M3) * p3 % ISG_M3;
	return r;
}

/* 32-bit variant for offsets inside one phase */
ISG_HD isg_wh isg_wh_jump32(const isg_wh_tables *t, isg_wh s, uint32_t n)
{
	uint32_t e1 = n % (ISG_M1 - 1), e2 = n % (ISG_M2 - 1), e3 = n % (ISG_M3 - 1);
	isg_wh r;
	r.s1 = s.s1 % ISG_M1 * ((uint32_t)t->lo[0][e1 & 255] * t->hi[0][e1 >> 8] % ISG_M1) % ISG_M1;
	r.s2 = s.s2 % ISG_M2 * ((uint32_t)t->lo[1][e2 & 255] * t->hi[1][e2 >> 8] % ISG_M2) % ISG_M2;
	r.s3 = s.s3 % ISG_M3 * ((uint32_t)t->lo[2][e3 & 255] * t->hi[2][e3 >> 8] % ISG_M3) % ISG_M3;
	return r;
}
/* multiplier triple a_r^n mod m_r (apply with isg_wh_mul) */
ISG_HD isg_wh isg_wh_power(const isg_wh_tables *t, uint32_t n)
{
	isg_wh one = {1, 1, 1};
	return isg_wh_jump32(t, one, n);
}
ISG_HD isg_wh isg_wh_mul(isg_wh s, isg_wh p)
{
	isg_wh r;
	r.s1 = s.s1 * p.s1 % ISG_M1;
	r.s2 = s.s2 * p.s2 % ISG_M2;
	r.s3 = s.s3 * p.s3 % ISG_M3;
	return r;
}

/* a stream cursor: number of uniforms drawn through it + either the generator state or, when the
 * uniforms of this stretch of the stream were generated beforehand, a pointer to them */
typedef struct {
	isg_wh s;
	uint32_t used;
	const double *tape;
} isg_cursor;

ISG_HD double isg_cur_next(isg_cursor *c)
{
	if (c->tape) return c->tape[c->used++];
	c->used++;
	return isg_wh_next(&c->s);
}

#endif
