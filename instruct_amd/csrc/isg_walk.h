/*
 * isg_walk.h -- the position recurrence of a run of Dirichlet draws, resolved on the whole chip.
 *
 * rdirich (random.c:264-280) draws its gammas one after the other from the single Wichmann-Hill stream, and every
 * rejected attempt of rgamma1 / rgamma2 (random.c:167-250) costs two more uniforms: where gamma g+1 starts depends on how
 * many attempts gamma g needed.  In the replay schedule update_P (mcmc.c:846-857: K L Dirichlets) and update_ZQ
 * (mcmc.c:1196-1198: one Dirichlet per individual) are such runs.  Written with x = the number of REJECTED attempts so far
 * (a "pair offset": the stream position of a gamma's next attempt is  pos[g] + 2 x,  pos[g] = where it would start had
 * every earlier gamma been accepted at once -- a property of the data) the recurrence is
 *
 *     x_{g+1} = NextSet_g(x_g),   NextSet_g(x) = the smallest x' >= x whose attempt  Acc_g(x')  is accepted,
 *
 * and Acc_g(x') only depends on (shape of g, the two uniforms at pos[g] + 2 x'): a BIT that every lane of the chip can
 * evaluate on its own.  (An attempt whose transformed u1 leaves (0,1), random.c:213-216, draws the next pair inside
 * rgamma2: for the consumption that is a rejection like any other.)
 *
 * The engine:  groups = runs of consecutive gammas (one Dirichlet: the K gammas of an individual, the A_j of a (cluster,
 * locus)); blocks = WK_BG consecutive groups; super-blocks = WK_FAN blocks; segments = as many super-blocks as one
 * workgroup's LDS holds maps for.  x performs a random walk with drift rho per gamma and spread sigma sqrt(gammas): block b
 * is entered somewhere in a WINDOW of offsets known beforehand, and
 *
 *   wk_table    per group: the accept bits of its gammas over the block's window, then for every column the chain of
 *               NextSet steps through the group's gammas -> one byte per (group, entry offset): rejected attempts.
 *   wk_block    per block: the table in LDS, a lane per entry offset walks the block's groups -> map F1 (16-bit deltas)
 *   wk_compose  per super-block: its blocks' maps in LDS, a lane per entry offset -> map F2
 *   wk_top      one workgroup: the segment's F2 maps in LDS, one lane walks the super-blocks from the segment's entry
 *   wk_expand   per super-block: from its entry through its blocks' maps -> every block's entry
 *   wk_final    per block: from its entry through the table -> every group's offset T[g]
 *
 * Every stage is exact; an offset outside its window is reported (WkState.fail) and the caller falls back to the
 * sequential sampler.  update_ZQ's shapes are not known before its Z draws are: there the bits come from INTERVALS of
 * shapes and a byte may be flagged UNCERTAIN (isg_spec_hip.inc); update_P's shapes are the allele counts + 1: exact.
 *
 * The bodies below run on the device (one call per thread, __syncthreads between phases) and, unchanged, under a host
 * emulation that loops over the threads phase by phase (tests/emul): WK_THREADS / WK_SYNC.
 */
#ifndef ISG_WALK_H
#define ISG_WALK_H
#include <stdint.h>
#include <string.h>
#include "isg_math.h"
#include "isg_wh.h"
#include "isg_sampler.h"

#define WK_BG 64        /* groups per block, at most */
#define WK_FAN 16       /* blocks per super-block */
#define WK_MAXSEG 64    /* segments per run */
#define WK_PADC 64      /* table columns evaluated beyond the window: room for a group's own rejections */
#define WK_NGMAX 64     /* gammas per group, at most */
#define WK_OUT16 0xffffu
#define WK_NOENT (-2147483647 - 1)
#define WK_IRR 0xffu    /* table byte: not representable (more than 254 / 126 rejections, or the bits ran out) */
#define WK_UFLAG 0x80u  /* interval mode: the byte's count is a guess (an attempt on its path is uncertain) */
#define WK_MODE_EXACT 0
#define WK_MODE_INTERVAL 1

#if defined(__HIP_DEVICE_COMPILE__)
#define WK_THREADS(t, nt) for (int t = (int)threadIdx.x, once_ = 1; once_; once_ = 0)
#define WK_SYNC() __syncthreads()
#define WK_ATOMIC_OR_LDS(p, v) atomicOr((p), (v))
#define WK_ATOMIC_ADD_LDS(p, v) atomicAdd((p), (v))
#define WK_ATOMIC_OR(p, v) atomicOr((p), (v))
#define WK_ATOMIC_ADD64(p, v) atomicAdd((p), (v))
#define WK_ATOMIC_ADD_F64(p, v) atomicAdd((p), (v))
#define WK_LOG2F(x) __builtin_amdgcn_logf(x)
#define WK_RCPF(x) __builtin_amdgcn_rcpf(x)
/* bit `col` of a row: the 64 lanes of a wave hold 64 consecutive columns starting at a multiple of 64 (loops of stride
 * nthreads over a multiple of 64 columns): one ballot, two stores by lane 0, no atomics */
#define WK_PUT_BIT(row, col, bit) do { const unsigned long long bm_ = __ballot(bit); if ((threadIdx.x & 63u) == 0u) { (row)[(col) >> 5] = (unsigned)bm_; (row)[((col) >> 5) + 1] = (unsigned)(bm_ >> 32); } } while (0)
#else
#include <math.h>
#define WK_THREADS(t, nt) for (int t = 0; t < (nt); t++)
#define WK_SYNC() do { } while (0)
#define WK_ATOMIC_OR_LDS(p, v) (*(p) |= (v))
static inline unsigned wk_host_fetch_add(unsigned *p, unsigned v) { const unsigned o = *p; *p += v; return o; }
#define WK_ATOMIC_ADD_LDS(p, v) wk_host_fetch_add((p), (v))
#define WK_ATOMIC_OR(p, v) (*(p) |= (v))
#define WK_ATOMIC_ADD64(p, v) (*(p) += (v))
#define WK_ATOMIC_ADD_F64(p, v) (*(p) += (v))
#define WK_LOG2F(x) log2f(x)
#define WK_RCPF(x) (1.0f / (x))
#define WK_PUT_BIT(row, col, bit) do { if (bit) (row)[(col) >> 5] |= 1u << ((col) & 31); } while (0)
#endif

typedef struct {
	int g0, ng;               /* first group, groups in this block */
	int wlo;                  /* first table column = pair offset relative to the segment's entry */
	int W;                    /* table columns (a multiple of 64) */
	int ein;                  /* entry offsets wlo .. wlo + ein - 1 are mapped (ein <= W) */
	unsigned foff;            /* its map F1: ein 16-bit deltas at maps + foff */
	unsigned long long toff;  /* its table: ng rows of W bytes at table + toff */
} WkBlock;
typedef struct {
	int b0, nb;               /* first block, blocks */
	unsigned foff;            /* its map F2 (entry window = that of block b0) */
	int pad;
} WkSuper;
typedef struct {
	int g0, g1, b0, nb, s0, ns;
	int bg;                   /* groups per block in this segment */
	int pad;
} WkSeg;
typedef struct {
	unsigned long long xin[WK_MAXSEG + 1]; /* origin of segment s's offsets = where the walk that built its tables entered it (xin[0] given by the caller) */
	long long ent[WK_MAXSEG + 1];          /* where the current walk enters segment s, from that origin (0 unless a later walk keeps the tables) */
	unsigned fail;            /* bit 0: an offset left its window; 1: irregular table byte on the path; 2: uncertain byte in a strict walk; 3: a Dirichlet's check */
	unsigned nfail_block;     /* (diagnostic) first block that failed + 1 */
	double pred;              /* rejected attempts predicted from the shapes (wk_centers, unscaled) */
	unsigned long long namb;  /* attempts decided in double precision (exact mode) */
	/* per walk (a second walk over the same tables zeroes from here on): */
	unsigned long long sum_d, nblk;        /* rejected attempts over all blocks walked, blocks */
	double resid2, ngam;                   /* squared deviations of the blocks' increments from the scaled prediction; gammas */
} WkState;

/* ---------------------------------------------------------------------------------------------------------------- */
/* one attempt                                                                                                        */
/* ---------------------------------------------------------------------------------------------------------------- */
/* rgamma2's constants for a shape (random.c:199-203) in the form the single precision test uses */
typedef struct {
	float c3, c5, c2m1, c2;
	int kind; /* 0: shape 1 (one uniform, always accepted); 1: shape < 1 (rgamma1); 2: 1 < shape <= 2.5; 3: shape > 2.5 */
} WkCoef;
ISG_HD WkCoef wk_coef(double a)
{
	WkCoef k;
	k.kind = (a == 1.0) ? 0 : (a < 1.0) ? 1 : (a > 2.5) ? 3 : 2;
	if (k.kind >= 2) {
		const double c1 = a - 1, c2 = (a - 1 / (6 * a)) / c1;
		k.c3 = (float)(2 / c1);
		k.c5 = (float)(1 / isg_sqrt(a));
		k.c2 = (float)c2;
		k.c2m1 = (float)(c2 - 1.0);
	} else {
		k.c3 = k.c5 = k.c2 = k.c2m1 = 0.f;
	}
	return k;
}
/*
 * The accept / reject decision of one rgamma2 attempt (random.c:205-229) in single precision: returns 1 accepted, 0 not
 * accepted (rejected, or u1 out of (0,1): the next pair is drawn either way).  *amb != 0: the single precision value cannot
 * be trusted (the caller decides in double).  With w = c2 u2 / v, d = w - 1:
 *     (c3 v + w + 1/w) > c4 = c3 + 2     <=>   d^2 > c3 (1 - v) (1 + d)
 *     c3 log v - log w + w >= 1          <=>   d - log(1 + d) >= -c3 log v
 * which avoids the cancellation of w + 1/w - 2 and w - 1 - log w for large shapes (d ~ shape^-1/2).
 */
ISG_HD float wk_h_small(float d) /* d - log(1 + d) for |d| < 1/8: d^2 (1/2 - d/3 + d^2/4 - ...) */
{
	float s = 1.0f / 9.0f;
	s = 1.0f / 8.0f - d * s;
	s = 1.0f / 7.0f - d * s;
	s = 1.0f / 6.0f - d * s;
	s = 1.0f / 5.0f - d * s;
	s = 1.0f / 4.0f - d * s;
	s = 1.0f / 3.0f - d * s;
	s = 1.0f / 2.0f - d * s;
	return d * d * s;
}
ISG_HD int wk_try_f32(const WkCoef k, float u1, float u2, int *amb)
{
	const float t = 1.0f - 1.86f * u1;
	const float v = (k.kind == 3) ? u2 + k.c5 * t : u1;
	const float num = (k.kind == 3) ? k.c2m1 * u2 - k.c5 * t : k.c2 * u2 - u1; /* c2 u2 - v */
	*amb = 0;
	if (!(v > 2e-6f)) { /* (NaN: ambiguous) */
		*amb = !(v < -2e-6f);
		return 0;
	}
	if (!(v < 1.0f - 2e-6f)) {
		*amb = !(v > 1.0f + 2e-6f);
		return 0;
	}
	const float rv = WK_RCPF(v);
	const float d = num * rv;
	const float ad = d < 0 ? -d : d;
	/* absolute uncertainty of d = num / v: the float images of the uniforms and constants and a handful of roundings on the
	 * terms of num (of size c5 and c2 - 1 for large shapes, c2 and 1 otherwise), 2e-7 absolute on v */
	const float ed = (4e-7f * ((k.kind == 3) ? k.c5 + k.c2m1 : k.c2 + 1.0f) + 3e-7f * ad) * rv;
	const float w = 1.0f + d;
	if (!(w > 1e-6f) || !(d < 1e6f)) { *amb = 1; return 0; }
	const float l1 = d * d, r1 = k.c3 * (1.0f - v) * w;
	const float m1 = r1 - l1; /* >= 0: accepted by the first test */
	const float t1 = 2e-5f * (l1 + r1) + (2.0f * ad + ed) * ed;
	if (m1 > t1) return 1;
	const int a1 = !(m1 < -t1); /* the first test is uncertain */
	/* second test: h(d) = d - log(1 + d) against -c3 log v */
	const float h = (ad < 0.125f) ? wk_h_small(d) : d - 0.69314718056f * WK_LOG2F(w);
	const float r2 = -k.c3 * 0.69314718056f * WK_LOG2F(v);
	const float m2 = r2 - h; /* > 0: accepted by the second test */
	const float t2 = 2e-5f * (h + r2) + ad * WK_RCPF(w) * ed + 4e-7f * (ad < 0.125f ? 0.f : ad + 1.0f);
	if (m2 > t2) return 1; /* accepted by the second test whatever the first says */
	if (m2 < -t2) {
		if (a1) *amb = 1; /* rejected by the second test, but the first (accepting) test is uncertain */
		return 0;
	}
	*amb = 1;
	return 0;
}

/*
 * The same decision for an INTERVAL of shapes [lo, hi], both beyond 2.5 (update_ZQ: the shape is a cluster count not yet drawn
 * + alpha).  For large shapes the decision hardly depends on the shape: with c5 = shape^-1/2, D = d / c5, rho = (c2 - 1) / c5,
 * kappa = c3 / c5^2 = 2 shape / (shape - 1) the two tests read
 *     D^2 <= kappa (1 - v) (1 + c5 D)            h(c5 D) / c5^2 < -kappa log v,       D = (rho u2 - t) / v,  v = u2 + c5 t
 * and the shape enters through c5 t in v, c5 D in d and the nearly constant rho, kappa only.  Each quantity carries the range it
 * can move over inside the interval (c5, rho, kappa are monotone in the shape: midpoint and half-range of their end values); a test
 * counts as CERTAIN when its margin clears the ranges of both sides.  Returns the decision at the interval's middle; *unc != 0: it
 * may be different elsewhere in the interval (or single precision cannot tell).
 */
typedef struct {
	float c5, e5, rho, erho, kap, ekap;
	int ok; /* 0: not an interval beyond 2.5: nothing about this gamma is certain */
} WkCoefI;
ISG_HD WkCoefI wk_coef_interval(double lo, double hi)
{
	WkCoefI k;
	k.c5 = k.e5 = k.rho = k.erho = k.kap = k.ekap = 0.f;
	k.ok = (lo > 2.5) && (hi >= lo);
	if (!k.ok) return k;
	const double c5a = 1 / isg_sqrt(lo), c5b = 1 / isg_sqrt(hi);
	const double ra = (1 - 1 / (6 * lo)) / (lo - 1) / c5a, rb = (1 - 1 / (6 * hi)) / (hi - 1) / c5b;
	const double ka = 2 * lo / (lo - 1), kb = 2 * hi / (hi - 1);
	/* midpoints and half-ranges (all three decrease with the shape); a relative 3e-7 on top for their single precision images */
	k.c5 = (float)(0.5 * (c5a + c5b));
	k.e5 = (float)(0.5 * (c5a - c5b) + 3e-7 * c5a);
	k.rho = (float)(0.5 * (ra + rb));
	k.erho = (float)(0.5 * (ra - rb) + 3e-7 * ra);
	k.kap = (float)(0.5 * (ka + kb));
	k.ekap = (float)(0.5 * (ka - kb) + 3e-7 * ka);
	return k;
}
ISG_HD int wk_try_interval_f32(const WkCoefI k, float u1, float u2, int *unc)
{
	const float t = 1.0f - 1.86f * u1, at = t < 0 ? -t : t;
	const float v = u2 + k.c5 * t;
	const float dv = k.e5 * at + 2e-6f; /* (+ the uniforms' single precision images: 1e-6 each) */
	if (!(v > 0.f)) { *unc = !(v + dv < 0.f); return 0; } /* (NaN: uncertain) */
	if (!(v < 1.0f)) { *unc = !(v - dv > 1.0f); return 0; }
	const float vm = v - dv;
	const int u = !(vm > 1e-4f) || !(v + dv < 1.0f); /* somewhere in the interval u1 leaves (0,1) */
	const float rv = WK_RCPF(v);
	const float D = (k.rho * u2 - t) * rv, aD = D < 0 ? -D : D;
	const float d = k.c5 * D, ad = d < 0 ? -d : d;
	const float w = 1.0f + d;
	/* The margins' derivatives with respect to c5 (the one constant that matters: v' = t, D' = -D t / v, d' = D u2 / v), times c5's
	 * half-range, times a factor for the curvature (relative (e5 / c5)^2 terms); rho and kappa move the margins by at most
	 * erho |dm/drho| + ekap |dm/dkap|.  A margin counts when it clears that and the single precision allowance. */
	const float tv = t * rv, dp = D * u2 * rv;
	const float adp = dp < 0 ? -dp : dp;
	const float fac = k.e5 * (1.25f + 4.0f * k.e5 * WK_RCPF(k.c5));
	const float dD_rho = k.erho * u2 * rv; /* range of D through rho */
	if (!(w - fac * adp > 1e-3f) || !(aD < 1e6f) || u) { *unc = 1; return (w > 0.f) && (D * D <= k.kap * (1.0f - v) * w); }
	const float L1 = D * D, R1 = k.kap * (1.0f - v) * w;
	const float m1 = R1 - L1; /* >= 0: accepted by the first test */
	const float m1p = k.kap * ((1.0f - v) * dp - t * w) + 2.0f * L1 * tv;
	const float s1 = 2e-5f * (L1 + R1) + (m1p < 0 ? -m1p : m1p) * fac + (2.0f * aD + k.kap * (1.0f - v) * k.c5) * dD_rho + k.ekap * (1.0f - v) * w + 1e-6f * (aD + 1.0f) * (aD + 1.0f) * rv;
	if (m1 > s1) { *unc = 0; return 1; }
	const int b1 = !(m1 < -s1); /* the first test may pass somewhere in the interval */
	const float lnv = -0.69314718056f * WK_LOG2F(v); /* > 0 */
	const float R2 = k.kap * lnv;
	float H, Hp, dHr;
	if (ad < 0.125f) { /* H = D^2 s(d), s = 1/2 - d/3 + d^2/4 ..., s' = -1/3 + d/2 - 3 d^2/5 + ... */
		const float s = (d == 0.f) ? 0.5f : wk_h_small(d) * WK_RCPF(d * d);
		float sp = 7.0f / 9.0f;
		sp = -6.0f / 8.0f + d * sp;
		sp = 5.0f / 7.0f + d * sp;
		sp = -4.0f / 6.0f + d * sp;
		sp = 3.0f / 5.0f + d * sp;
		sp = -2.0f / 4.0f + d * sp;
		sp = 1.0f / 3.0f + d * sp;
		sp = -sp;
		H = L1 * s;
		Hp = -2.0f * L1 * tv * s + L1 * sp * dp;
		dHr = 2.0f * aD * s * dD_rho;
	} else { /* H = h(d) / c5^2, h' = d / (1 + d) */
		const float hh = d - 0.69314718056f * WK_LOG2F(w);
		const float i5 = WK_RCPF(k.c5 * k.c5);
		H = hh * i5;
		Hp = d * WK_RCPF(w) * dp * i5 - 2.0f * H * WK_RCPF(k.c5);
		dHr = ad * WK_RCPF(w) * k.c5 * dD_rho * i5 + 4e-7f * (ad + 1.0f) * i5;
	}
	const float m2 = R2 - H; /* > 0: accepted by the second test */
	const float m2p = -k.kap * tv - Hp;
	const float s2 = 2e-5f * (H + R2) + (m2p < 0 ? -m2p : m2p) * fac + dHr + k.ekap * lnv + 1e-6f * (aD + 1.0f) * (aD + 1.0f) * rv;
	if (m2 > s2) { *unc = 0; return 1; } /* accepted by the second test whatever the first says */
	if (m2 < -s2) { *unc = b1; return 0; }
	*unc = 1;
	return (m1 >= 0.f) || (m2 > 0.f);
}
/* the same decision exactly as the reference takes it (isg_rgamma2_try without its retry loop), from the pair's position */
ISG_HD int wk_try_exact(const isg_wh_tables *tab, isg_wh base, unsigned long long pos, double a)
{
	isg_wh s = isg_wh_jump(tab, base, pos);
	double u1 = isg_wh_next(&s);
	const double u2 = isg_wh_next(&s);
	const double c1 = a - 1, c2 = (a - 1 / (6 * a)) / c1, c3 = 2 / c1, c4 = c3 + 2, c5 = 1 / isg_sqrt(a);
	if (a > 2.5) u1 = u2 + c5 * (1 - 1.86 * u1);
	if ((u1 >= 1) || (u1 <= 0)) return 0;
	const double w = c2 * u2 / u1;
	if ((c3 * u1 + w + 1 / w) > c4) {
		if ((c3 * isg_log(u1) - isg_log(w) + w) >= 1) return 0;
	}
	return 1;
}

/* smallest set bit at or after X in a row of nwords 32-bit words; -1: none */
ISG_HD int wk_nextset(const unsigned *row, int nwords, int X)
{
	int w = X >> 5;
	if (w >= nwords) return -1;
	const unsigned v = row[w] >> (X & 31);
	if (v) return X + __builtin_ctz(v);
	for (w++; w < nwords; w++)
		if (row[w]) return (w << 5) + __builtin_ctz(row[w]);
	return -1;
}

/* The same through a 64-bit window (the row has a word beyond nwords: an accepted attempt is almost never further than 33 columns away, so
 * the loop above is the rare path), and whether a row of flags has a bit set in [X, result] (urow: null = no such row). */
ISG_HD int wk_nextset_flag(const unsigned *row, const unsigned *urow, int nwords, int X, int *flag)
{
	const int w = X >> 5, sh = X & 31;
	if (w >= nwords) return -1;
	const unsigned long long v = ((((unsigned long long)row[w + 1]) << 32) | (unsigned long long)row[w]) >> sh;
	if (v) {
		const int c = __builtin_ctzll(v);
		if (X + c >= (nwords << 5)) return -1;
		if (urow) {
			const unsigned long long uv = ((((unsigned long long)urow[w + 1]) << 32) | (unsigned long long)urow[w]) >> sh;
			if (uv & ((2ull << c) - 1ull)) *flag = 1;
		}
		return X + c;
	}
	const int X1 = wk_nextset(row, nwords, X);
	if (X1 >= 0 && urow) {
		const int U1 = wk_nextset(urow, nwords, X);
		if (U1 >= 0 && U1 <= X1) *flag = 1;
	}
	return X1;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* wk_table                                                                                                           */
/* ---------------------------------------------------------------------------------------------------------------- */
typedef struct {
	const WkBlock *blk;
	const int *gam0;                 /* [groups + 1] first gamma of a group */
	const unsigned long long *gpos;  /* [gammas] stream position (from the phase's base) of a gamma's first attempt when nothing before it was rejected */
	const int *gcnt;                 /* exact mode: the gamma's count (shape = count + 1.0) */
	const float *alo, *ahi;          /* interval mode: its shape is somewhere in [alo, ahi]; alo <= 2.5: nothing is certain */
	unsigned char *table;
	WkState *st;
	isg_wh base;                     /* the stream at the phase's base */
	const isg_wh_tables *tab;
	const float *tape;               /* the phase's uniforms as floats, stream order from its base (may be null) ... */
	unsigned long long tape_len;     /* ... this many of them: stretches beyond are generated on the spot */
	int seg_b0, seg_g0, bg, seg, mode;
} WkTableArgs;

/* LDS of wk_table: float tape[wk_tape_floats(W)], WkCoefI coef[WK_NGMAX] (exact mode: WkCoef), unsigned amb[1 + WK_AMBCAP], then the bit rows (accept; interval mode: + uncertain) */
#define WK_AMBCAP 255 /* attempts per group whose single precision decision is redone in double after the loop (about one in 1e5) */
ISG_HD size_t wk_tape_floats(int W) { return (((size_t)2 * (W + WK_PADC) + 2 * WK_NGMAX + 8) + 3) & ~(size_t)3; }
ISG_HD size_t wk_table_lds_bytes(int W, int ngmax, int mode)
{
	const size_t rows = (size_t)ngmax * ((W + WK_PADC) / 32 + 1) * (mode == WK_MODE_INTERVAL ? 2 : 1);
	return sizeof(float) * wk_tape_floats(W) + sizeof(WkCoefI) * WK_NGMAX + sizeof(unsigned) * (1 + WK_AMBCAP) + sizeof(unsigned) * rows;
}

/* one workgroup per group: wg = group - seg_g0; nthreads a multiple of 64 */
ISG_HD void wk_table_body(const WkTableArgs A, int wg, int nthreads, unsigned char *lds)
{
	const int g = A.seg_g0 + wg;
	const WkBlock B = A.blk[A.seg_b0 + wg / A.bg];
	const int r = g - B.g0;
	const int gm0 = A.gam0[g], ng = A.gam0[g + 1] - gm0;
	const int W = B.W, Wp = W + WK_PADC, nw = Wp / 32 + 1;
	const unsigned long long p0 = A.gpos[gm0];
	const int span = (int)(A.gpos[gm0 + ng - 1] - p0);
	const int ntape = 2 * Wp + span + 2;
	float *tape = (float *)lds;
	WkCoef *coef = (WkCoef *)(lds + sizeof(float) * wk_tape_floats(W));
	WkCoefI *coefi = (WkCoefI *)coef; /* (interval mode) */
	unsigned *ambl = (unsigned *)(coefi + WK_NGMAX); /* [0]: how many; then (gamma << 20 | column) */
	unsigned *rows = ambl + 1 + WK_AMBCAP; /* accept rows, then (interval mode) uncertain rows */
	const long long x0s = (long long)A.st->xin[A.seg] + (long long)B.wlo;
	if (x0s < 0) { /* a window reaching below offset 0 (a later segment entered almost at once): no table, the walk stops here */
		WK_THREADS(t, nthreads) {
			unsigned char *out = A.table + B.toff + (size_t)r * W;
			for (int col = t; col < W; col += nthreads) out[col] = (unsigned char)WK_IRR;
		}
		return;
	}
	const unsigned long long x0 = (unsigned long long)x0s; /* absolute pair offset of column 0 */
	const unsigned long long s0 = p0 + 2ull * x0; /* stream position of tape[0] */
	const bool interval = (A.mode == WK_MODE_INTERVAL);
	/* phase 0: the stretch of the stream this group can touch, as floats (8 per skip-ahead); the gammas' constants; rows zeroed */
	WK_THREADS(t, nthreads) {
		if (A.tape && s0 + (unsigned long long)ntape <= A.tape_len) { /* (the same for every thread) */
			const float *src = A.tape + s0;
			int i = t;
			for (; i + 3 * nthreads < ntape; i += 4 * nthreads) { /* four loads in flight per thread */
				const float a = src[i], b = src[i + nthreads], c = src[i + 2 * nthreads], d = src[i + 3 * nthreads];
				tape[i] = a; tape[i + nthreads] = b; tape[i + 2 * nthreads] = c; tape[i + 3 * nthreads] = d;
			}
			for (; i < ntape; i += nthreads) tape[i] = src[i];
		} else {
			for (int i0 = t * 16; i0 < ntape; i0 += nthreads * 16) { /* 16 per skip-ahead */
				isg_wh s = isg_wh_jump(A.tab, A.base, s0 + (unsigned long long)i0);
				for (int k = 0; k < 16 && i0 + k < ntape; k++) {
					isg_wh_step(&s);
					/* interval mode: the bits are speculation anyway, the cheaper single precision evaluation (error < 1e-6) will do */
					tape[i0 + k] = interval ? isg_wh_value_f32(&s) : (float)isg_wh_value(&s);
				}
			}
		}
		for (int k = t; k < ng * nw * (interval ? 2 : 1); k += nthreads) rows[k] = 0u;
		if (t == 0) ambl[0] = 0u;
		if (t < ng) {
			if (interval) coefi[t] = wk_coef_interval((double)A.alo[gm0 + t], (double)A.ahi[gm0 + t]);
			else coef[t] = wk_coef((double)A.gcnt[gm0 + t] + 1.0);
		}
	}
	WK_SYNC();
	/* phase 1: the accept bit of every (gamma, column) */
	WK_THREADS(t, nthreads) {
		for (int m = 0; m < ng; m++) {
			const int om = (int)(A.gpos[gm0 + m] - p0);
			WkCoef k0;
			WkCoefI ki;
			if (interval) { ki = coefi[m]; k0 = wk_coef(1.0); }
			else { k0 = coef[m]; ki = wk_coef_interval(0.0, 0.0); }
			for (int col = t; col < Wp; col += nthreads) {
				const int o = om + 2 * col;
				int acc, unc = 0, amb0;
				if (!interval) {
					if (k0.kind == 0) acc = 1; /* rexp: one uniform, no rejection (random.c:243-244); pos[] accounts for the odd step */
					else {
						acc = wk_try_f32(k0, tape[o], tape[o + 1], &amb0);
						if (amb0) { /* noted, decided in double after the loop (nobody waits for a neighbour's logarithms); the list full: here */
							const unsigned slot = WK_ATOMIC_ADD_LDS(&ambl[0], 1u);
							acc = 0;
							if (slot < WK_AMBCAP) ambl[1 + slot] = ((unsigned)m << 20) | (unsigned)col;
							else acc = wk_try_exact(A.tab, A.base, s0 + (unsigned long long)o, (double)A.gcnt[gm0 + m] + 1.0);
						}
					}
				} else if (!ki.ok) { /* small or unknown shapes: nothing about this gamma is certain */
					acc = 1;
					unc = 1;
				} else { /* the decision at the interval's middle, certain if it holds for every shape of the interval */
					acc = wk_try_interval_f32(ki, tape[o], tape[o + 1], &unc);
				}
				WK_PUT_BIT(rows + m * nw, col, acc);
				if (interval) WK_PUT_BIT(rows + (ng + m) * nw, col, unc);
			}
		}
	}
	WK_SYNC();
	if (!interval) { /* phase 1b: the noted attempts exactly */
		WK_THREADS(t, nthreads) {
			const unsigned n = ambl[0] < WK_AMBCAP ? ambl[0] : WK_AMBCAP;
			for (unsigned e = (unsigned)t; e < n; e += (unsigned)nthreads) {
				const int m = (int)(ambl[1 + e] >> 20), col = (int)(ambl[1 + e] & 0xfffffu);
				const int o = (int)(A.gpos[gm0 + m] - p0) + 2 * col;
				if (wk_try_exact(A.tab, A.base, s0 + (unsigned long long)o, (double)A.gcnt[gm0 + m] + 1.0)) WK_ATOMIC_OR_LDS(&rows[m * nw + (col >> 5)], 1u << (col & 31));
			}
			if (t == 0 && ambl[0]) WK_ATOMIC_ADD64(&A.st->namb, (unsigned long long)ambl[0]);
		}
		WK_SYNC();
	}
	/* phase 2: per column the chain through the group's gammas */
	WK_THREADS(t, nthreads) {
		unsigned char *out = A.table + B.toff + (size_t)r * W;
		const int nwv = Wp / 32;
		for (int col = t; col < W; col += nthreads) {
			int X = col, bad = 0, unc = 0;
			for (int m = 0; m < ng; m++) {
				const int X1 = wk_nextset_flag(rows + m * nw, interval ? rows + (ng + m) * nw : (const unsigned *)0, nwv, X, &unc);
				if (X1 < 0) { bad = 1; break; }
				X = X1;
			}
			const int c = X - col;
			unsigned char v;
			if (interval) v = (bad || c > 126) ? (unsigned char)WK_IRR : (unsigned char)(c | (unc ? WK_UFLAG : 0));
			else v = (bad || c > 254) ? (unsigned char)WK_IRR : (unsigned char)c;
			out[col] = v;
		}
	}
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* wk_centers: where the windows sit                                                                                  */
/* ---------------------------------------------------------------------------------------------------------------- */
/* mean number of rejected attempts of rgamma (random.c:233-250) as a function of the shape: 1 / P(accept) - 1, integrated
 * numerically over the unit square of (u1, u2) (3000 x 3000 midpoints); linear in the shape up to 2.5, in its reciprocal beyond */
ISG_HD float wk_expected_rej(float a)
{
	if (!(a > 1.0f)) return (a == 1.0f) ? 0.0f : 0.4f; /* shape 1: none; below 1 (rgamma1): a guess */
	if (a <= 2.5f) {
		if (a <= 1.5f) return 0.28f + (a - 1.0f) * (0.3445f - 0.28f) / 0.5f;
		if (a <= 2.0f) return 0.3445f + (a - 1.5f) * (0.41020f - 0.3445f) / 0.5f;
		if (a <= 2.25f) return 0.41020f + (a - 2.0f) * (0.45448f - 0.41020f) / 0.25f;
		return 0.45448f + (a - 2.25f) * (0.50068f - 0.45448f) / 0.25f;
	}
	const float xs[17] = {1.0f / 2.5f, 1.0f / 2.75f, 1.0f / 3.0f, 1.0f / 3.5f, 1.0f / 4.0f, 1.0f / 5.0f, 1.0f / 6.0f, 1.0f / 8.0f, 1.0f / 10.0f, 1.0f / 14.0f, 1.0f / 20.0f,
			      1.0f / 30.0f, 1.0f / 50.0f, 1.0f / 100.0f, 1.0f / 300.0f, 1.0f / 1000.0f, 0.0f};
	const float ys[17] = {0.76536f, 0.73560f, 0.71170f, 0.67558f, 0.64950f, 0.61426f, 0.59147f, 0.56369f, 0.54735f, 0.52894f, 0.51532f,
			      0.50482f, 0.49649f, 0.49026f, 0.48613f, 0.48468f, 0.48405f};
	const float x = 1.0f / a;
	for (int k = 1; k < 17; k++)
		if (x >= xs[k]) return ys[k] + (x - xs[k]) * (ys[k - 1] - ys[k]) / (xs[k - 1] - xs[k]);
	return ys[16];
}
typedef struct {
	WkBlock *blk;                    /* wlo of the segment's blocks is written here */
	const int *hw;                   /* [blocks] half-width of the entry window (host: k sigma sqrt(gammas before it) + slack) */
	float *bpred;                    /* [blocks] predicted rejections inside the block (unscaled) */
	WkState *st;
	const int *gam0;
	const int *gcnt;
	const float *alo, *ahi;
	WkSeg seg;                       /* wk_centers: this segment (wk_bpred: blocks are numbered through all segments) */
	const WkSeg *segs;               /* wk_centers launched once for all segments: segment = workgroup (null: `seg`) */
	int mode;
	float scale;                     /* measured drift / predicted drift of earlier runs */
} WkCenterArgs;
/* wk_bpred: one workgroup per block of the segment (64 threads: one per group): the rejections the block's gammas are expected to add */
ISG_HD void wk_bpred_body(const WkCenterArgs A, int wg, int nthreads, unsigned char *lds)
{
	float *part = (float *)lds;
	const WkBlock B = A.blk[wg];
	WK_THREADS(t, nthreads) {
		float s = 0.f;
		for (int r = t; r < B.ng; r += nthreads)
			for (int gm = A.gam0[B.g0 + r]; gm < A.gam0[B.g0 + r + 1]; gm++)
				s += (A.mode == WK_MODE_EXACT) ? wk_expected_rej((float)A.gcnt[gm] + 1.0f) : wk_expected_rej(0.5f * (A.alo[gm] + A.ahi[gm]));
		part[t] = s;
	}
	WK_SYNC();
	WK_THREADS(t, nthreads) {
		if (t == 0) {
			float s = 0.f;
			for (int k = 0; k < nthreads; k++) s += part[k]; /* (fixed order: the same sum on every run) */
			A.bpred[wg] = s;
		}
	}
}
/* wk_centers: one workgroup per segment; the windows' first columns from the running sum of the predictions */
ISG_HD void wk_centers_body(WkCenterArgs A, int wg, int nthreads, unsigned char *lds)
{
	float *sum = (float *)lds;
	if (A.segs) A.seg = A.segs[wg];
	WK_THREADS(t, nthreads) {
		for (int b = t; b < A.seg.nb; b += nthreads) sum[b] = A.bpred[A.seg.b0 + b];
	}
	WK_SYNC();
	WK_THREADS(t, nthreads) {
		/* every thread sums the blocks before its own from LDS, always in the same order (a few hundred blocks at most: no scan needed,
		 * and no global access on anybody's chain) */
		for (int b = t; b <= A.seg.nb; b += nthreads) {
			float run = 0.f;
			for (int k = 0; k < b; k++) run += sum[k];
			if (b == A.seg.nb) {
				WK_ATOMIC_ADD_F64(&A.st->pred, (double)run);
			} else {
				const int c = (int)(run * A.scale);
				int lo = c - A.hw[A.seg.b0 + b];
				if (lo < 0 && A.seg.g0 == 0) lo = 0; /* (later segments: a second walk may enter below the origin, hw[first block] > 0) */
				A.blk[A.seg.b0 + b].wlo = lo;
			}
		}
	}
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* the walk                                                                                                           */
/* ---------------------------------------------------------------------------------------------------------------- */
typedef struct {
	const WkBlock *blk;
	const WkSuper *sup;
	const unsigned char *table;
	unsigned short *maps;     /* F1 of every block, F2 of every super-block */
	int *ent_sup;             /* [super-blocks] entry offset (relative to the segment's origin); WK_NOENT: none */
	int *ent_blk;             /* [blocks] */
	unsigned long long *T;    /* [groups + 1] absolute pair offset at which a group starts */
	WkState *st;
	WkSeg seg;
	int segno, mode, strict;  /* strict: an UNCERTAIN byte on a path ends it (interval mode) */
	int keep_origin;          /* the tables exist already (a second walk): the segments' origins stay, the entries move */
	int total_groups;         /* T[total_groups] = the run's final offset */
	const float *bpred;       /* statistics: the blocks' predicted rejections ... */
	const int *gam0;          /* ... and gammas */
	float scale;
} WkWalkArgs;

/* n 16-byte words from global memory into LDS, all threads (both 16-byte aligned) */
ISG_HD void wk_copy16(void *dst, const void *src, int n16, int t, int nthreads)
{
#if defined(__HIP_DEVICE_COMPILE__)
	const uint4 *s4 = (const uint4 *)src;
	uint4 *d4 = (uint4 *)dst;
	int k = t;
	for (; k + 3 * nthreads < n16; k += 4 * nthreads) { /* four loads in flight per thread */
		const uint4 a = s4[k], b = s4[k + nthreads], c = s4[k + 2 * nthreads], d = s4[k + 3 * nthreads];
		d4[k] = a; d4[k + nthreads] = b; d4[k + 2 * nthreads] = c; d4[k + 3 * nthreads] = d;
	}
	for (; k < n16; k += nthreads) d4[k] = s4[k];
#else
	for (int k = t; k < n16; k += nthreads) memcpy((char *)dst + 16 * (size_t)k, (const char *)src + 16 * (size_t)k, 16);
#endif
}
#define WK_EIN8(e) (((e) + 7) & ~7) /* maps are stored and staged in multiples of 8 deltas (16 bytes) */

/* increment of a table byte, or -1 (the path ends here); *why: bit 1 irregular, bit 2 uncertain */
ISG_HD int wk_byte_step(unsigned v, int mode, int strict, unsigned *why)
{
	if (v == WK_IRR) { *why |= 2u; return -1; }
	if (mode == WK_MODE_INTERVAL) {
		if (strict && (v & WK_UFLAG)) { *why |= 4u; return -1; }
		return (int)(v & 0x7fu);
	}
	return (int)v;
}

/* wk_block: workgroup per block; LDS = the block's table (ng * W bytes) */
ISG_HD void wk_block_body(const WkWalkArgs A, int wg, int nthreads, unsigned char *lds)
{
	const WkBlock B = A.blk[A.seg.b0 + wg];
	const int W = B.W;
	WK_THREADS(t, nthreads) {
		wk_copy16(lds, A.table + B.toff, B.ng * W / 16, t, nthreads); /* W is a multiple of 64, toff of 64 */
	}
	WK_SYNC();
	WK_THREADS(t, nthreads) {
		unsigned short *F = A.maps + B.foff;
		for (int e = t; e < B.ein; e += nthreads) {
			int col = e;
			unsigned why = 0;
			for (int r = 0; r < B.ng; r++) {
				if (col >= W) { why |= 1u; break; }
				const int c = wk_byte_step(lds[r * W + col], A.mode, A.strict, &why);
				if (c < 0) break;
				col += c;
			}
			const int d = col - e;
			F[e] = (why || d >= (int)WK_OUT16) ? (unsigned short)WK_OUT16 : (unsigned short)d;
		}
	}
}

/* wk_compose: workgroup per super-block; LDS = the F1 maps of its blocks, back to back */
ISG_HD void wk_compose_body(const WkWalkArgs A, int wg, int nthreads, unsigned char *lds)
{
	const WkSuper S = A.sup[A.seg.s0 + wg];
	unsigned short *L = (unsigned short *)lds;
	WK_THREADS(t, nthreads) {
		int o = 0;
		for (int k = 0; k < S.nb; k++) {
			const WkBlock B = A.blk[S.b0 + k];
			wk_copy16(L + o, A.maps + B.foff, WK_EIN8(B.ein) / 8, t, nthreads);
			o += WK_EIN8(B.ein);
		}
	}
	WK_SYNC();
	WK_THREADS(t, nthreads) {
		const WkBlock B0 = A.blk[S.b0];
		unsigned short *F2 = A.maps + S.foff;
		for (int e = t; e < B0.ein; e += nthreads) {
			int x = B0.wlo + e, o = 0, ok = 1;
			for (int k = 0; k < S.nb; k++) {
				const WkBlock B = A.blk[S.b0 + k];
				const int col = x - B.wlo;
				if (col < 0 || col >= B.ein) { ok = 0; break; }
				const unsigned d = L[o + col];
				if (d == WK_OUT16) { ok = 0; break; }
				x += (int)d;
				o += WK_EIN8(B.ein);
			}
			const int d = x - (B0.wlo + e);
			F2[e] = (!ok || d >= (int)WK_OUT16) ? (unsigned short)WK_OUT16 : (unsigned short)d;
		}
	}
}

/* wk_top: one workgroup; LDS = the F2 maps of the segment's super-blocks; one lane walks them from the segment's entry */
ISG_HD void wk_top_body(const WkWalkArgs A, int nthreads, unsigned char *lds)
{
	unsigned short *L = (unsigned short *)lds;
	WK_THREADS(t, nthreads) {
		int o = 0;
		for (int k = 0; k < A.seg.ns; k++) {
			const WkSuper S = A.sup[A.seg.s0 + k];
			const int ein = A.blk[S.b0].ein;
			wk_copy16(L + o, A.maps + S.foff, WK_EIN8(ein) / 8, t, nthreads);
			o += WK_EIN8(ein);
		}
	}
	WK_SYNC();
	WK_THREADS(t, nthreads) {
		if (t == 0) {
			int x = (int)A.st->ent[A.segno], o = 0, ok = 1;
			if (A.st->fail) ok = 0; /* an earlier segment failed: nothing to continue from */
			for (int k = 0; k < A.seg.ns && ok; k++) {
				const WkSuper S = A.sup[A.seg.s0 + k];
				const WkBlock B0 = A.blk[S.b0];
				const int col = x - B0.wlo;
				A.ent_sup[A.seg.s0 + k] = x;
				if (col < 0 || col >= B0.ein) { ok = 0; break; }
				const unsigned d = L[o + col];
				if (d == WK_OUT16) { ok = 0; break; }
				x += (int)d;
				o += WK_EIN8(B0.ein);
			}
			if (!ok) {
				WK_ATOMIC_OR(&A.st->fail, 1u);
				for (int k = 0; k < A.seg.ns; k++) A.ent_sup[A.seg.s0 + k] = WK_NOENT;
				if (!A.keep_origin) A.st->xin[A.segno + 1] = A.st->xin[A.segno];
			} else if (!A.keep_origin) {
				A.st->xin[A.segno + 1] = A.st->xin[A.segno] + (unsigned long long)x;
				A.st->ent[A.segno + 1] = 0;
			} else {
				A.st->ent[A.segno + 1] = (long long)(A.st->xin[A.segno] + (unsigned long long)x) - (long long)A.st->xin[A.segno + 1];
			}
		}
	}
}

/* wk_expand: workgroup per super-block; from its entry through its blocks' F1 maps (LDS as in wk_compose) */
ISG_HD void wk_expand_body(const WkWalkArgs A, int wg, int nthreads, unsigned char *lds)
{
	const WkSuper S = A.sup[A.seg.s0 + wg];
	unsigned short *L = (unsigned short *)lds;
	const int ent = A.ent_sup[A.seg.s0 + wg];
	WK_THREADS(t, nthreads) {
		int o = 0;
		for (int k = 0; k < S.nb; k++) {
			const WkBlock B = A.blk[S.b0 + k];
			if (ent != WK_NOENT) wk_copy16(L + o, A.maps + B.foff, WK_EIN8(B.ein) / 8, t, nthreads);
			o += WK_EIN8(B.ein);
		}
	}
	WK_SYNC();
	WK_THREADS(t, nthreads) {
		if (t == 0) {
			int x = ent, o = 0, ok = (ent != WK_NOENT);
			for (int k = 0; k < S.nb; k++) {
				const WkBlock B = A.blk[S.b0 + k];
				if (ok) {
					const int col = x - B.wlo;
					A.ent_blk[S.b0 + k] = x;
					if (col < 0 || col >= B.ein || L[o + col] == WK_OUT16) {
						/* this block's own walk (wk_final) names the reason; the blocks behind it have no entry */
						ok = 0;
					} else {
						x += (int)L[o + col];
					}
				} else {
					A.ent_blk[S.b0 + k] = WK_NOENT;
				}
				o += WK_EIN8(B.ein);
			}
		}
	}
}

/* wk_final: workgroup per block; LDS = a strip of the block's table starting at the entry column; one lane walks the groups */
#define WK_STRIP 512
ISG_HD void wk_final_body(const WkWalkArgs A, int wg, int nthreads, unsigned char *lds)
{
	const int b = A.seg.b0 + wg;
	const WkBlock B = A.blk[b];
	const int ent = A.ent_blk[b];
	const int W = B.W;
	const int c0 = (ent == WK_NOENT || ent < B.wlo) ? 0 : ((ent - B.wlo) & ~15);
	int cw = W - c0;
	if (cw > WK_STRIP) cw = WK_STRIP;
	if (cw < 0) cw = 0;
	WK_THREADS(t, nthreads) {
		if (ent != WK_NOENT && ent >= B.wlo && ent - B.wlo < W) {
			const int n16 = cw / 16; /* W, c0 multiples of 16 */
			for (int k = t; k < B.ng * (WK_STRIP / 16); k += nthreads) {
				const int r = k / (WK_STRIP / 16), q = k % (WK_STRIP / 16);
				if (q < n16) wk_copy16(lds + r * WK_STRIP + 16 * q, A.table + B.toff + (size_t)r * W + c0 + 16 * q, 1, 0, 1);
			}
		}
	}
	WK_SYNC();
	WK_THREADS(t, nthreads) {
		if (t == 0 && ent != WK_NOENT) {
			const unsigned long long xin = A.st->xin[A.segno];
			int col = ent - B.wlo;
			unsigned why = 0;
			int r = 0;
			if (col < 0 || col >= B.ein) why = 1u;
			for (; r < B.ng && !why; r++) {
				if (col >= W) { why |= 1u; break; }
				A.T[B.g0 + r] = (unsigned long long)((long long)xin + (long long)(B.wlo + col));
				const int q = col - c0;
				const unsigned v = (q >= 0 && q < cw) ? lds[r * WK_STRIP + q] : A.table[B.toff + (size_t)r * W + col];
				const int c = wk_byte_step(v, A.mode, A.strict, &why);
				if (c < 0) break;
				col += c;
			}
			if (why) {
				WK_ATOMIC_OR(&A.st->fail, why);
				if (!A.st->nfail_block) A.st->nfail_block = (unsigned)b + 1u;
			} else {
				const unsigned long long d = (unsigned long long)((long long)(B.wlo + col) - (long long)ent);
				const double rs = (double)d - (double)(A.bpred[b] * A.scale);
				WK_ATOMIC_ADD64(&A.st->sum_d, d);
				WK_ATOMIC_ADD64(&A.st->nblk, 1ull);
				WK_ATOMIC_ADD_F64(&A.st->resid2, rs * rs);
				WK_ATOMIC_ADD_F64(&A.st->ngam, (double)(A.gam0[B.g0 + B.ng] - A.gam0[B.g0]));
				if (B.g0 + B.ng == A.total_groups) A.T[A.total_groups] = (unsigned long long)((long long)xin + (long long)(B.wlo + col));
			}
		}
	}
}

#endif
