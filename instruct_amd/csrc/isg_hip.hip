/*
 * isg_hip.hip -- MI355X (gfx950) implementation of the InStruct per-iteration MCMC hot path
 * behind the C ABI of include/instruct_hip.h.
 *
 * Device data (all resident in HBM for the lifetime of a context):
 *   geno  uint8 [N][Lp][2]   allele code per copy, 0xFF = locus unused for this individual
 *                            (missing, or allelenum[j] <= 1; mcmc.c:817,1137,1737), Lp = L padded to 8
 *   z     uint8 [N][Lp][2]   cluster of origin per allele copy (UPMCMC.z, mcmc.h:17)
 *   freq  f64   [L][Amax][KP]  allele frequencies, cluster index innermost (one 16B-aligned K-vector
 *                            per (locus, allele): what a Z draw needs in one contiguous read)
 *   cnt   i32   [L][Amax][K] allele counts (seqpop, mcmc.c:807), same locus-major order
 *   qq    f64   [N][K], qqnum i32 [N][K], gen i32 [N], indvlkh f64 [N]
 *
 * Kernels (wave64; no MFMA: there is no dense contraction on this path):
 *   k_count      allele-count histogram: loci tile per workgroup, per-thread private counters in
 *                LDS ([counter][thread] layout: conflict free), coalesced integer atomics flush
 *   k_gprop      update_G proposals: selfing -> dt_stat -> stream position (prefix scan in replay
 *                schedule) -> rgeom proposal + acceptance uniform
 *   k_loglik     log_ld_indv: one workgroup per individual, 8-byte packed loads, per-lane
 *                order-independent fixed-point accumulators, integer tree reduction
 *   k_zq         update_ZQ: Z draws (one uniform per allele copy at its stream position) + per
 *                individual histogram + Dirichlet.  Keyed schedule: one workgroup per individual.
 *                Replay schedule: one persistent workgroup walks the individuals in order, because
 *                individual i+1's first position depends on how many uniforms i's Dirichlet used.
 *   k_pdirich    update_P Dirichlet draws, one lane per (cluster, locus) (keyed schedule)
 *
 * Compile with -ffp-contract=off: isg_math.h relies on plain IEEE operations.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/instruct_hip.h"
#include "isg_math.h"
#include "isg_wh.h"
#include "isg_sampler.h"

#define ISG_KCAP 32
#define ISG_LPT 4 /* loci per lane per pass: 8 bytes of geno + 8 bytes of z */

static thread_local std::string g_err;
static int fail(const std::string &m)
{
	g_err = m;
	return 1;
}
#define HIPCHK(x)                                                                                     \
	do {                                                                                          \
		hipError_t e_ = (x);                                                                  \
		if (e_ != hipSuccess)                                                                 \
			return fail(std::string(#x) + ": " + hipGetErrorString(e_));                  \
	} while (0)

struct DevView {
	int N, L, Lp, K, KP, Amax, mode, type_freq;
	const uint8_t *geno;
	uint8_t *z;
	const int *allelenum;
	const int *nvalid; /* [N] loci used per individual */
	double *freq;
	int *cnt;
	double *qq;
	int *qqnum;
	int *gen;
	int *genprop;
	double *uacc;
	double *indvlkh;
	const isg_wh_tables *tab;
	unsigned *err;
};

struct ProfEntry {
	std::string name;
	double ms;
	long n;
};
struct ProfPending {
	const char *name;
	hipEvent_t e0, e1;
};

struct isg_ctx {
	isg_config cfg;
	DevView d;
	int Amax;
	hipStream_t stream;
	/* host mirrors (reference layouts) */
	std::vector<int> allelenum;
	std::vector<double> freq;   /* [K][L][Amax] */
	std::vector<double> qq;     /* [N][K] */
	std::vector<int> qqnum;     /* [N][K] */
	std::vector<int> gen;       /* [N] */
	std::vector<double> S;      /* [K] */
	std::vector<int> state;     /* [K] */
	std::vector<double> indvlkh;
	std::vector<int> cnt_h;     /* device order [L][Amax][K] */
	std::vector<double> freq_stage; /* device order [L][Amax][KP] */
	double alpha, totallkh;
	bool qq_dirty_host;         /* host qq newer than device */
	/* stream */
	isg_wh rng;                 /* current sequential state (replay) */
	isg_wh origin;              /* chain origin (keyed) */
	long raw_seed[3];
	bool raw_valid;
	uint64_t iter;
	uint64_t ky[9];
	isg_wh_tables tab_h;
	/* device scratch */
	uint64_t *d_pos;
	unsigned *d_err;
	double *d_S;
	/* profiling */
	bool prof;
	std::vector<ProfEntry> prof_entries;
	std::vector<ProfPending> prof_pending;
	std::vector<hipEvent_t> prof_free;
	hipEvent_t prof_cur;
};

/* ------------------------------------------------------------------------------------------ */
/* device helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

/* block-wide exclusive scan of a small per-thread count; returns prefix, *total = block sum */
template <int BLOCK>
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned *sm /* [BLOCK/64 + 1] */, unsigned *total)
{
	const int NW = BLOCK / 64;
	unsigned lane = lane_id(), w = threadIdx.x >> 6, incl = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		unsigned t = __shfl_up(incl, o, 64);
		if (lane >= (unsigned)o) incl += t;
	}
	if (lane == 63) sm[w] = incl;
	__syncthreads();
	unsigned base = 0, tot = 0;
#pragma unroll
	for (int i = 0; i < NW; i++) {
		unsigned s = sm[i];
		if (i < (int)w) base += s;
		tot += s;
	}
	__syncthreads();
	*total = tot;
	return base + incl - v;
}

/* order-independent accumulator: block reduction through 32-bit limbs (no carries while summing) */
struct AccLimbs {
	unsigned long long l0, l1, l2, l3;
	unsigned flags;
};
__device__ __forceinline__ AccLimbs acc_to_limbs(const isg_acc &a)
{
	AccLimbs r;
	r.l0 = a.lo & 0xffffffffULL;
	r.l1 = a.lo >> 32;
	r.l2 = a.hi & 0xffffffffULL;
	r.l3 = a.hi >> 32;
	r.flags = a.flags;
	return r;
}
__device__ __forceinline__ isg_acc limbs_to_acc(const AccLimbs &s)
{
	isg_acc a;
	unsigned long long t0 = s.l0, t1 = s.l1 + (t0 >> 32), t2 = s.l2 + (t1 >> 32), t3 = s.l3 + (t2 >> 32);
	a.lo = (t0 & 0xffffffffULL) | (t1 << 32);
	a.hi = (t2 & 0xffffffffULL) | (t3 << 32);
	a.flags = s.flags;
	return a;
}
template <int BLOCK>
__device__ __forceinline__ isg_acc block_reduce_acc(const isg_acc &a, unsigned long long *sm /* [BLOCK/64][5] */)
{
	const int NW = BLOCK / 64;
	AccLimbs s = acc_to_limbs(a);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		s.l0 += __shfl_down(s.l0, o, 64);
		s.l1 += __shfl_down(s.l1, o, 64);
		s.l2 += __shfl_down(s.l2, o, 64);
		s.l3 += __shfl_down(s.l3, o, 64);
		s.flags |= __shfl_down(s.flags, o, 64);
	}
	unsigned w = threadIdx.x >> 6;
	if (lane_id() == 0) {
		sm[w * 5 + 0] = s.l0;
		sm[w * 5 + 1] = s.l1;
		sm[w * 5 + 2] = s.l2;
		sm[w * 5 + 3] = s.l3;
		sm[w * 5 + 4] = s.flags;
	}
	__syncthreads();
	AccLimbs t = {0, 0, 0, 0, 0};
	for (int i = 0; i < NW; i++) {
		t.l0 += sm[i * 5 + 0];
		t.l1 += sm[i * 5 + 1];
		t.l2 += sm[i * 5 + 2];
		t.l3 += sm[i * 5 + 3];
		t.flags |= (unsigned)sm[i * 5 + 4];
	}
	__syncthreads();
	return limbs_to_acc(t);
}

/* ------------------------------------------------------------------------------------------ */
/* k_count: allele counts                                                                      */
/* ------------------------------------------------------------------------------------------ */
/*
 * grid = (loci tiles, row blocks).  Lane t of a tile owns loci [J0 + LPT*t, +LPT): nobody else
 * in the workgroup touches its counters, so plain LDS read-modify-writes suffice.
 * LDS layout lds[c * BLOCK + t], c = (l * Amax + a) * K + k  -> bank = t mod 32: conflict free.
 */
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_count(DevView d, int rows_per_block)
{
	extern __shared__ unsigned lds_cnt[];
	const int C = ISG_LPT * d.Amax * d.K;
	const int t = threadIdx.x;
	for (int c = 0; c < C; c++) lds_cnt[c * BLOCK + t] = 0;
	const int J0 = blockIdx.x * BLOCK * ISG_LPT;
	const int j0 = J0 + t * ISG_LPT;
	const int r0 = blockIdx.y * rows_per_block;
	int r1 = r0 + rows_per_block;
	if (r1 > d.N) r1 = d.N;
	const size_t rowb = (size_t)d.Lp * 2;
	if (j0 < d.Lp) {
		for (int i = r0; i < r1; i++) {
			const uint2 g = *(const uint2 *)(d.geno + (size_t)i * rowb + (size_t)j0 * 2);
			const uint2 zz = *(const uint2 *)(d.z + (size_t)i * rowb + (size_t)j0 * 2);
			unsigned long long gb = ((unsigned long long)g.y << 32) | g.x, zb = ((unsigned long long)zz.y << 32) | zz.x;
#pragma unroll
			for (int l = 0; l < ISG_LPT; l++) {
				unsigned a0 = (unsigned)(gb >> (16 * l)) & 0xff, a1 = (unsigned)(gb >> (16 * l + 8)) & 0xff;
				unsigned z0 = (unsigned)(zb >> (16 * l)) & 0xff, z1 = (unsigned)(zb >> (16 * l + 8)) & 0xff;
				if (a0 != 0xff) {
					lds_cnt[((l * d.Amax + a0) * d.K + z0) * BLOCK + t] += 1;
					lds_cnt[((l * d.Amax + a1) * d.K + z1) * BLOCK + t] += 1;
				}
			}
		}
	}
	__syncthreads();
	/* flush in global order: element e of the tile's contiguous [locus][a][k] range */
	const int per_locus = d.Amax * d.K;
	int tile_loci = d.L - J0;
	if (tile_loci > BLOCK * ISG_LPT) tile_loci = BLOCK * ISG_LPT;
	if (tile_loci < 0) tile_loci = 0;
	const int E = tile_loci * per_locus;
	for (int e = t; e < E; e += BLOCK) {
		int jl = e / per_locus, rem = e - jl * per_locus; /* rem = a*K + k */
		unsigned v = lds_cnt[((jl % ISG_LPT) * per_locus + rem) * BLOCK + jl / ISG_LPT];
		if (v) atomicAdd(&d.cnt[(size_t)J0 * per_locus + e], (int)v);
	}
}

/* ------------------------------------------------------------------------------------------ */
/* k_gprop: update_G proposals (mcmc.c:1060-1084) and stream positions                          */
/* ------------------------------------------------------------------------------------------ */
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_gprop(DevView d, const double *S, isg_wh base, int keyed, uint64_t *pos_out)
{
	__shared__ unsigned sm[BLOCK / 64 + 1];
	__shared__ double Ssh[ISG_KCAP];
	if (threadIdx.x < (unsigned)d.K) Ssh[threadIdx.x] = S[threadIdx.x];
	__syncthreads();
	unsigned running = 0;
	for (int i0 = 0; i0 < d.N; i0 += BLOCK) {
		int i = i0 + threadIdx.x;
		double selfing = 0;
		int stat = 0;
		if (i < d.N) {
			for (int k = 0; k < d.K; k++) selfing += d.qq[(size_t)i * d.K + k] * Ssh[k];
			stat = isg_dt_stat(selfing);
			if (stat < 0) { atomicOr(d.err, 2u); stat = 1; }
		}
		unsigned flag = (i < d.N && stat == 1) ? 1u : 0u, tot, pre;
		pre = block_excl_scan<BLOCK>(flag, sm, &tot);
		if (i < d.N) {
			uint64_t pos = keyed ? 2ull * (uint64_t)i : (uint64_t)i + running + pre;
			isg_cursor c;
			c.s = isg_wh_jump(d.tab, base, pos);
			c.used = 0;
			int gen;
			if (stat == 1) {
				gen = isg_rgeom(&c, 1 - selfing);
				if (gen < 1) gen = 1;
				if (gen > 50) gen = 50;
			} else if (stat == 0) gen = 1;
			else gen = 50;
			d.genprop[i] = gen;
			d.uacc[i] = isg_cur_next(&c);
		}
		running += tot;
	}
	if (threadIdx.x == 0) *pos_out = (uint64_t)d.N + running;
}

/* ------------------------------------------------------------------------------------------ */
/* k_loglik: log_ld_indv (mcmc.c:1726-1773) for a proposal/current pair or for cal_lkh          */
/* ------------------------------------------------------------------------------------------ */
/* PAIR = true : update_G -- both generations in one pass over the row, MH accept at the end
 * PAIR = false: cal_lkh  -- indvlkh[i] (mode 2: current generation; mode 1: log_ld_noselfing_indv) */
template <int BLOCK, bool PAIR>
__global__ void __launch_bounds__(BLOCK) k_loglik(DevView d)
{
	__shared__ unsigned long long sm[(BLOCK / 64) * 5];
	__shared__ double qsh[ISG_KCAP];
	const int i = blockIdx.x;
	int gp = 0, gc;
	if (PAIR) {
		gp = d.genprop[i];
		gc = d.gen[i];
		if (gp == gc) return; /* identical sums: ratio is exactly 1, accepted, generation unchanged */
	} else {
		gc = (d.mode == 2) ? d.gen[i] : -1;
	}
	if (d.type_freq == 0) {
		if (threadIdx.x < (unsigned)d.K) qsh[threadIdx.x] = d.qq[(size_t)i * d.K + threadIdx.x];
		__syncthreads();
	}
	const double log2c = isg_log(2.0);
	isg_acc accC, accP, accX; /* current-gen terms, proposed-gen terms, generation-independent terms */
	isg_acc_zero(&accC);
	isg_acc_zero(&accP);
	isg_acc_zero(&accX);
	const size_t rowb = (size_t)d.Lp * 2;
	const uint8_t *grow = d.geno + (size_t)i * rowb;
	const uint8_t *zrow = d.z + (size_t)i * rowb;
	for (int j0 = threadIdx.x * ISG_LPT; j0 < d.Lp; j0 += BLOCK * ISG_LPT) {
		const uint2 g = *(const uint2 *)(grow + (size_t)j0 * 2);
		const uint2 zz = *(const uint2 *)(zrow + (size_t)j0 * 2);
		unsigned long long gb = ((unsigned long long)g.y << 32) | g.x, zb = ((unsigned long long)zz.y << 32) | zz.x;
#pragma unroll
		for (int l = 0; l < ISG_LPT; l++) {
			unsigned a0 = (unsigned)(gb >> (16 * l)) & 0xff, a1 = (unsigned)(gb >> (16 * l + 8)) & 0xff;
			unsigned z0 = (unsigned)(zb >> (16 * l)) & 0xff, z1 = (unsigned)(zb >> (16 * l + 8)) & 0xff;
			if (a0 == 0xff) continue;
			const int j = j0 + l;
			const double *F0 = d.freq + ((size_t)j * d.Amax + a0) * d.KP;
			const double *F1 = d.freq + ((size_t)j * d.Amax + a1) * d.KP;
			if (gc < 0) { /* mode 1: mcmc.c:1881-1886 */
				isg_acc_add(&accC, isg_log(F0[z0]));
				isg_acc_add(&accC, isg_log(F1[z1]));
				if (a0 != a1) isg_acc_add(&accC, log2c);
			} else if (d.type_freq == 0) { /* mcmc.c:1739-1748 */
				double t0 = 0, t1 = 0;
				for (int m = 0; m < d.K; m++) t0 += F0[m] * qsh[m];
				for (int m = 0; m < d.K; m++) t1 += F1[m] * qsh[m];
				isg_acc_add(&accC, isg_log(isg_genofreq(a0 == a1, t0, t1, gc)));
				if (PAIR) isg_acc_add(&accP, isg_log(isg_genofreq(a0 == a1, t0, t1, gp)));
			} else if (z0 == z1) { /* mcmc.c:1752-1758 */
				double f0 = F0[z0], f1 = F1[z1];
				isg_acc_add(&accC, isg_log(isg_genofreq(a0 == a1, f0, f1, gc)));
				if (PAIR) isg_acc_add(&accP, isg_log(isg_genofreq(a0 == a1, f0, f1, gp)));
			} else { /* mcmc.c:1760-1767 */
				isg_acc_add(&accX, isg_log(F0[z0]));
				isg_acc_add(&accX, isg_log(F1[z1]));
				if (a0 != a1) isg_acc_add(&accX, log2c);
			}
		}
	}
	isg_acc rc, rp, rx;
	rc = block_reduce_acc<BLOCK>(accC, sm);
	rx = block_reduce_acc<BLOCK>(accX, sm);
	if (PAIR) rp = block_reduce_acc<BLOCK>(accP, sm);
	if (threadIdx.x == 0) {
		isg_acc tc = rc;
		isg_acc_merge(&tc, &rx);
		double lc = isg_acc_value(&tc);
		if (PAIR) {
			isg_acc tp = rp;
			isg_acc_merge(&tp, &rx);
			double lp = isg_acc_value(&tp);
			double mh = isg_exp(lp - lc);
			double thr = (1 > mh) ? mh : 1; /* MIN2(1, mhratio), mcmc.h:10 */
			if (d.uacc[i] < thr) d.gen[i] = gp;
		} else {
			d.indvlkh[i] = lc;
		}
	}
}

/* ------------------------------------------------------------------------------------------ */
/* k_zq: update_ZQ (mcmc.c:1122-1203)                                                          */
/* ------------------------------------------------------------------------------------------ */
template <int BLOCK, int KMAX>
__device__ __forceinline__ uint64_t zq_one(const DevView &d, int i, isg_wh base, uint64_t pos, int init_flag, double alpha,
					   unsigned *sm_scan, int *sm_hist, uint64_t *sm_pos)
{
	const int K = d.K, t = threadIdx.x;
	const int nvalid = d.nvalid[i];
	const bool fast = (nvalid == d.L);
	double q[KMAX];
#pragma unroll
	for (int m = 0; m < KMAX; m++) q[m] = (m < K && !init_flag) ? d.qq[(size_t)i * K + m] : 0.0;
	int cnt[KMAX];
#pragma unroll
	for (int m = 0; m < KMAX; m++) cnt[m] = 0;
	const size_t rowb = (size_t)d.Lp * 2;
	const uint8_t *grow = d.geno + (size_t)i * rowb;
	uint8_t *zrow = d.z + (size_t)i * rowb;
	unsigned running = 0; /* valid loci before this pass */
	for (int jb = 0; jb < d.Lp; jb += BLOCK * ISG_LPT) {
		const int j0 = jb + t * ISG_LPT;
		unsigned long long gb = ~0ull;
		if (j0 < d.Lp) {
			const uint2 g = *(const uint2 *)(grow + (size_t)j0 * 2);
			gb = ((unsigned long long)g.y << 32) | g.x;
		}
		unsigned nv = 0;
#pragma unroll
		for (int l = 0; l < ISG_LPT; l++) nv += (((unsigned)(gb >> (16 * l)) & 0xff) != 0xff) ? 1u : 0u;
		unsigned rank_loci;
		if (fast) {
			rank_loci = (unsigned)j0;
		} else {
			unsigned tot;
			rank_loci = running + block_excl_scan<BLOCK>(nv, sm_scan, &tot);
			running += tot;
		}
		if (j0 < d.Lp) {
			unsigned long long zb = ~0ull;
			if (nv) {
				isg_wh s = isg_wh_jump(d.tab, base, pos + 2ull * rank_loci);
#pragma unroll
				for (int l = 0; l < ISG_LPT; l++) {
					unsigned a0 = (unsigned)(gb >> (16 * l)) & 0xff;
					if (a0 == 0xff) continue;
#pragma unroll
					for (int cp = 0; cp < 2; cp++) {
						unsigned a = (unsigned)(gb >> (16 * l + 8 * cp)) & 0xff;
						double x = isg_wh_next(&s);
						double cum[KMAX];
						if (init_flag) {
#pragma unroll
							for (int m = 0; m < KMAX; m++) cum[m] = (m < K) ? (double)(m + 1) / K : 0.0;
						} else {
							const double *F = d.freq + ((size_t)(j0 + l) * d.Amax + a) * d.KP;
							double run = 0;
#pragma unroll
							for (int m = 0; m < KMAX; m++) {
								if (m < K) {
									double w = q[m] * F[m];
									run = (m == 0) ? w : run + w;
								}
								cum[m] = run;
							}
						}
						/* disc_unif (random.c:403-430) on cum[0..K-1] */
						double tot = cum[0];
#pragma unroll
						for (int m = 1; m < KMAX; m++) if (m == K - 1) tot = cum[m];
						double prev = cum[0] / tot;
						int zsel = 0;
						if (!(x <= prev && x >= 0.0)) {
#pragma unroll
							for (int m = 1; m < KMAX; m++) {
								if (m < K) {
									double cur = cum[m] / tot;
									if (x > prev && x <= cur) zsel = m;
									prev = cur;
								}
							}
						}
#pragma unroll
						for (int m = 0; m < KMAX; m++) cnt[m] += (zsel == m) ? 1 : 0;
						zb = (zb & ~(0xffull << (16 * l + 8 * cp))) | ((unsigned long long)zsel << (16 * l + 8 * cp));
					}
				}
			}
			uint2 zo;
			zo.x = (unsigned)zb;
			zo.y = (unsigned)(zb >> 32);
			*(uint2 *)(zrow + (size_t)j0 * 2) = zo;
		}
	}
	/* qqnum[i][m] (mcmc.c:1176-1194): wave reduce, then LDS */
	__syncthreads();
	if (t < KMAX) sm_hist[t] = 0;
	__syncthreads();
#pragma unroll
	for (int m = 0; m < KMAX; m++) {
		if (m < K) {
			int v = cnt[m];
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
			if (lane_id() == 0 && v) atomicAdd(&sm_hist[m], v);
		}
	}
	__syncthreads();
	if (t == 0) {
		/* rdirich(qqnum[i], K, &qq[i], alpha) (mcmc.c:1196-1198) */
		isg_cursor c;
		c.s = isg_wh_jump(d.tab, base, pos + 2ull * (unsigned)nvalid);
		c.used = 0;
		double g[KMAX], sum = 0;
#pragma unroll
		for (int m = 0; m < KMAX; m++) {
			if (m < K) {
				g[m] = isg_rgamma(&c, (double)sm_hist[m] + alpha);
				sum += g[m];
			}
		}
#pragma unroll
		for (int m = 0; m < KMAX; m++) {
			if (m < K) {
				d.qq[(size_t)i * K + m] = g[m] / sum;
				d.qqnum[(size_t)i * K + m] = sm_hist[m];
			}
		}
		*sm_pos = pos + 2ull * (unsigned)nvalid + c.used;
	}
	__syncthreads();
	return *sm_pos;
}

template <int BLOCK, int KMAX, bool CHAIN>
__global__ void __launch_bounds__(BLOCK) k_zq(DevView d, isg_wh base, uint64_t pos0, uint64_t stride, int init_flag, double alpha,
					      uint64_t *pos_out)
{
	__shared__ unsigned sm_scan[BLOCK / 64 + 1];
	__shared__ int sm_hist[KMAX];
	__shared__ uint64_t sm_pos;
	if (CHAIN) {
		uint64_t pos = pos0;
		for (int i = 0; i < d.N; i++) pos = zq_one<BLOCK, KMAX>(d, i, base, pos, init_flag, alpha, sm_scan, sm_hist, &sm_pos);
		if (threadIdx.x == 0) *pos_out = pos;
	} else {
		int i = blockIdx.x;
		zq_one<BLOCK, KMAX>(d, i, base, pos0 + (uint64_t)i * stride, init_flag, alpha, sm_scan, sm_hist, &sm_pos);
	}
}

/* ------------------------------------------------------------------------------------------ */
/* k_pdirich: update_P Dirichlets (mcmc.c:846-857), keyed schedule: one lane per (cluster, locus) */
/* ------------------------------------------------------------------------------------------ */
__global__ void k_pdirich(DevView d, isg_wh base, uint64_t pos0, uint64_t SP)
{
	const int id = blockIdx.x * blockDim.x + threadIdx.x; /* id = k * L + j (reference order) */
	if (id >= d.K * d.L) return;
	const int k = id / d.L, j = id - k * d.L;
	const int A = d.allelenum[j];
	if (A <= 1) return;
	isg_cursor c;
	c.s = isg_wh_jump(d.tab, base, pos0 + (uint64_t)id * SP);
	c.used = 0;
	double sum = 0;
	for (int a = 0; a < A; a++) {
		double g = isg_rgamma(&c, (double)d.cnt[((size_t)j * d.Amax + a) * d.K + k] + 1.0);
		d.freq[((size_t)j * d.Amax + a) * d.KP + k] = g;
		sum += g;
	}
	for (int a = 0; a < A; a++) d.freq[((size_t)j * d.Amax + a) * d.KP + k] /= sum;
}

/* ------------------------------------------------------------------------------------------ */
/* host side                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* Per-kernel timing with HIP events recorded on the launch stream.  Events come from a pool and are
 * only resolved in prof_collect(), so enabling the profile does not add host/device syncs to the
 * region being timed. */
static void prof_collect(isg_ctx *c)
{
	for (auto &p : c->prof_pending) {
		(void)hipEventSynchronize(p.e1);
		float ms = 0;
		(void)hipEventElapsedTime(&ms, p.e0, p.e1);
		bool found = false;
		for (auto &e : c->prof_entries)
			if (e.name == p.name) {
				e.ms += ms;
				e.n++;
				found = true;
				break;
			}
		if (!found) c->prof_entries.push_back({p.name, (double)ms, 1});
		c->prof_free.push_back(p.e0);
		c->prof_free.push_back(p.e1);
	}
	c->prof_pending.clear();
}
static hipEvent_t prof_event(isg_ctx *c)
{
	if (c->prof_free.empty()) {
		if (c->prof_pending.size() >= 2048) prof_collect(c);
		if (c->prof_free.empty()) {
			hipEvent_t e;
			(void)hipEventCreate(&e);
			return e;
		}
	}
	hipEvent_t e = c->prof_free.back();
	c->prof_free.pop_back();
	return e;
}
static void prof_begin(isg_ctx *c)
{
	if (!c->prof) return;
	c->prof_cur = prof_event(c);
	(void)hipEventRecord(c->prof_cur, c->stream);
}
static void prof_end(isg_ctx *c, const char *name)
{
	if (!c->prof) return;
	hipEvent_t e1 = prof_event(c);
	(void)hipEventRecord(e1, c->stream);
	c->prof_pending.push_back({name, c->prof_cur, e1});
}

static int check_dev_err(isg_ctx *c)
{
	unsigned e = 0;
	HIPCHK(hipMemcpyAsync(&e, c->d_err, sizeof(e), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (e & 2u) return fail("The value of selfing rate or inbreeding coefficient is beyond [0,1]!");
	if (e) return fail("device error flag set");
	return 0;
}

static void keyed_layout(isg_ctx *c)
{
	uint64_t N = c->cfg.N, L = c->cfg.L, P = c->cfg.P, K = c->cfg.K, A = c->Amax;
	uint64_t SP = 16 * A + 16, SZ = P * L + 16 * K + 16, ZI0 = 1 + 2 * N, B0 = ZI0 + N * SZ;
	uint64_t offS = K * L * SP, offG = offS + 4 * K, offZ = offG + 2 * N, offA = offZ + N * SZ, BLK = offA + 4;
	uint64_t v[9] = {SP, SZ, ZI0, B0, offS, offG, offZ, offA, BLK};
	memcpy(c->ky, v, sizeof(v));
}
enum { KY_SP, KY_SZ, KY_ZI0, KY_B0, KY_OFFS, KY_OFFG, KY_OFFZ, KY_OFFA, KY_BLK };
static bool is_keyed(const isg_ctx *c) { return c->cfg.rng_sched == ISG_SCHED_KEYED; }
static uint64_t iter_base(const isg_ctx *c) { return c->ky[KY_B0] + c->iter * c->ky[KY_BLK]; }

/* sequential host draws */
static double host_next(isg_ctx *c)
{
	c->raw_valid = false;
	return isg_wh_next(&c->rng);
}
static void host_seek(isg_ctx *c, uint64_t pos) { c->rng = isg_wh_jump(&c->tab_h, c->origin, pos); c->raw_valid = false; }
static void host_advance(isg_ctx *c, uint64_t n) { c->rng = isg_wh_jump(&c->tab_h, c->rng, n); c->raw_valid = false; }

extern "C" const char *isg_last_error(void) { return g_err.c_str(); }

extern "C" int isg_ctx_create(const isg_config *cfg, const int32_t *allelenum, const int32_t *geno, const int32_t *missindx, isg_ctx **out)
{
	*out = nullptr;
	if (cfg->P != 2) return fail("isg_ctx_create: only diploid data (P = 2) is supported by this build");
	if (cfg->K < 1 || cfg->K > ISG_KCAP) return fail("isg_ctx_create: K must be in 1..32");
	if (cfg->mode != 1 && cfg->mode != 2) return fail("isg_ctx_create: mode must be 1 or 2");
	if (cfg->N < 1 || cfg->L < 1) return fail("isg_ctx_create: empty problem");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("isg_ctx_create: no HIP device available (the MI355X path has no CPU fallback)");
	if (cfg->device < 0 || cfg->device >= ndev) return fail("isg_ctx_create: bad device ordinal");
	HIPCHK(hipSetDevice(cfg->device));
	isg_ctx *c = new isg_ctx();
	c->cfg = *cfg;
	const int N = cfg->N, L = cfg->L, K = cfg->K;
	int Amax = 0;
	for (int j = 0; j < L; j++) Amax = allelenum[j] > Amax ? allelenum[j] : Amax;
	if (Amax > 254) { delete c; return fail("isg_ctx_create: more than 254 alleles at a locus"); }
	if (Amax < 1) Amax = 1;
	c->Amax = Amax;
	const int Lp = (L + 7) & ~7, KP = (K + 1) & ~1;
	c->allelenum.assign(allelenum, allelenum + L);
	/* pack genotypes: one byte per copy, 0xFF = locus unused (mcmc.c:817,1137,1737) */
	std::vector<uint8_t> pk((size_t)N * Lp * 2, 0xff);
	std::vector<int> nvalid(N, 0);
	for (int i = 0; i < N; i++)
		for (int j = 0; j < L; j++) {
			if (missindx[(size_t)i * L + j] == 1 || allelenum[j] <= 1) continue;
			int a0 = geno[((size_t)i * L + j) * 2], a1 = geno[((size_t)i * L + j) * 2 + 1];
			if (a0 < 0 || a1 < 0 || a0 >= allelenum[j] || a1 >= allelenum[j]) { delete c; return fail("isg_ctx_create: allele code out of range at a non-missing locus"); }
			pk[((size_t)i * Lp + j) * 2] = (uint8_t)a0;
			pk[((size_t)i * Lp + j) * 2 + 1] = (uint8_t)a1;
			nvalid[i]++;
		}
	DevView &d = c->d;
	memset(&d, 0, sizeof(d));
	d.N = N; d.L = L; d.Lp = Lp; d.K = K; d.KP = KP; d.Amax = Amax; d.mode = cfg->mode; d.type_freq = cfg->type_freq;
	HIPCHK(hipStreamCreate(&c->stream));
	void *p;
#define DALLOC(field, type, count)                                  \
	HIPCHK(hipMalloc(&p, sizeof(type) * (size_t)(count)));      \
	HIPCHK(hipMemset(p, 0, sizeof(type) * (size_t)(count)));    \
	field = (type *)p;
	uint8_t *dg;
	DALLOC(dg, uint8_t, (size_t)N * Lp * 2);
	d.geno = dg;
	HIPCHK(hipMemcpy(dg, pk.data(), pk.size(), hipMemcpyHostToDevice));
	DALLOC(d.z, uint8_t, (size_t)N * Lp * 2);
	HIPCHK(hipMemset(d.z, 0xff, (size_t)N * Lp * 2));
	int *dan, *dnv;
	DALLOC(dan, int, L);
	HIPCHK(hipMemcpy(dan, allelenum, sizeof(int) * L, hipMemcpyHostToDevice));
	d.allelenum = dan;
	DALLOC(dnv, int, N);
	HIPCHK(hipMemcpy(dnv, nvalid.data(), sizeof(int) * N, hipMemcpyHostToDevice));
	d.nvalid = dnv;
	DALLOC(d.freq, double, (size_t)Lp * Amax * KP);
	DALLOC(d.cnt, int, (size_t)Lp * Amax * K);
	DALLOC(d.qq, double, (size_t)N * K);
	DALLOC(d.qqnum, int, (size_t)N * K);
	DALLOC(d.gen, int, N);
	DALLOC(d.genprop, int, N);
	DALLOC(d.uacc, double, N);
	DALLOC(d.indvlkh, double, N);
	isg_wh_tables_init(&c->tab_h);
	isg_wh_tables *dt;
	DALLOC(dt, isg_wh_tables, 1);
	HIPCHK(hipMemcpy(dt, &c->tab_h, sizeof(isg_wh_tables), hipMemcpyHostToDevice));
	d.tab = dt;
	DALLOC(c->d_pos, uint64_t, 4);
	DALLOC(c->d_err, unsigned, 1);
	DALLOC(c->d_S, double, ISG_KCAP);
	d.err = c->d_err;
#undef DALLOC
	c->freq.assign((size_t)K * L * Amax, 0.0);
	c->freq_stage.assign((size_t)Lp * Amax * KP, 0.0);
	c->qq.assign((size_t)N * K, 0.0);
	c->qqnum.assign((size_t)N * K, 0);
	c->gen.assign(N, 0);
	c->S.assign(K, 0.0);
	c->state.assign(K, 0);
	c->indvlkh.assign(N, 0.0);
	c->cnt_h.assign((size_t)L * Amax * K, 0);
	c->alpha = 0;
	c->totallkh = 0;
	c->iter = 0;
	c->rng.s1 = 13; c->rng.s2 = 4; c->rng.s3 = 1972; /* random.c:10-12 */
	c->origin = c->rng;
	c->raw_seed[0] = 13; c->raw_seed[1] = 4; c->raw_seed[2] = 1972;
	c->raw_valid = true;
	c->prof = false;
	keyed_layout(c);
	*out = c;
	return 0;
}

extern "C" void isg_ctx_destroy(isg_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->cfg.device);
	(void)hipStreamSynchronize(c->stream);
	DevView &d = c->d;
	(void)hipFree((void *)d.geno); (void)hipFree(d.z); (void)hipFree((void *)d.allelenum); (void)hipFree((void *)d.nvalid); (void)hipFree(d.freq); (void)hipFree(d.cnt);
	(void)hipFree(d.qq); (void)hipFree(d.qqnum); (void)hipFree(d.gen); (void)hipFree(d.genprop); (void)hipFree(d.uacc); (void)hipFree(d.indvlkh);
	(void)hipFree((void *)d.tab); (void)hipFree(c->d_pos); (void)hipFree(c->d_err); (void)hipFree(c->d_S);
	prof_collect(c);
	for (auto e : c->prof_free) (void)hipEventDestroy(e);
	(void)hipStreamDestroy(c->stream);
	delete c;
}

extern "C" int isg_set_seeds(isg_ctx *c, long s1, long s2, long s3)
{
	if (s1 < 0 || s2 < 0 || s3 < 0) return fail("isg_set_seeds: negative seeds are not supported");
	c->rng.s1 = (uint32_t)(s1 % ISG_M1); c->rng.s2 = (uint32_t)(s2 % ISG_M2); c->rng.s3 = (uint32_t)(s3 % ISG_M3);
	c->raw_seed[0] = s1; c->raw_seed[1] = s2; c->raw_seed[2] = s3;
	c->raw_valid = true;
	return 0;
}
extern "C" int isg_get_seeds(isg_ctx *c, long s[3])
{
	if (c->raw_valid) { s[0] = c->raw_seed[0]; s[1] = c->raw_seed[1]; s[2] = c->raw_seed[2]; }
	else { s[0] = c->rng.s1; s[1] = c->rng.s2; s[2] = c->rng.s3; }
	return 0;
}
extern "C" double isg_ran1(isg_ctx *c) { return host_next(c); }

extern "C" int isg_keyed_layout(isg_ctx *c, uint64_t out[9])
{
	memcpy(out, c->ky, sizeof(c->ky));
	return 0;
}

/* ---- uploads / downloads ---- */
static int upload_freq(isg_ctx *c)
{
	const int L = c->cfg.L, K = c->cfg.K, A = c->Amax, KP = c->d.KP;
	for (int k = 0; k < K; k++)
		for (int j = 0; j < L; j++)
			for (int a = 0; a < A; a++) c->freq_stage[((size_t)j * A + a) * KP + k] = c->freq[((size_t)k * L + j) * A + a];
	HIPCHK(hipMemcpyAsync(c->d.freq, c->freq_stage.data(), sizeof(double) * (size_t)L * A * KP, hipMemcpyHostToDevice, c->stream));
	return 0;
}
static int download_freq(isg_ctx *c)
{
	const int L = c->cfg.L, K = c->cfg.K, A = c->Amax, KP = c->d.KP;
	HIPCHK(hipMemcpyAsync(c->freq_stage.data(), c->d.freq, sizeof(double) * (size_t)L * A * KP, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (int k = 0; k < K; k++)
		for (int j = 0; j < L; j++)
			for (int a = 0; a < A; a++) c->freq[((size_t)k * L + j) * A + a] = c->freq_stage[((size_t)j * A + a) * KP + k];
	return 0;
}
static int sync_qq_to_host(isg_ctx *c)
{
	HIPCHK(hipMemcpyAsync(c->qq.data(), c->d.qq, sizeof(double) * c->qq.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(c->qqnum.data(), c->d.qqnum, sizeof(int) * c->qqnum.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return 0;
}
static int sync_gen_to_host(isg_ctx *c)
{
	HIPCHK(hipMemcpyAsync(c->gen.data(), c->d.gen, sizeof(int) * c->gen.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return 0;
}

/* ---- counts ---- */
static int launch_count(isg_ctx *c)
{
	DevView &d = c->d;
	const int BLOCK = 256;
	size_t lds = (size_t)ISG_LPT * d.Amax * d.K * BLOCK * sizeof(unsigned);
	if (lds > 160 * 1024) return fail("isg_count_alleles: Amax*K too large for the LDS-resident count tile of this build");
	HIPCHK(hipMemsetAsync(d.cnt, 0, sizeof(int) * (size_t)d.L * d.Amax * d.K, c->stream));
	int tiles = (d.Lp + BLOCK * ISG_LPT - 1) / (BLOCK * ISG_LPT);
	int want_blocks = 1024;
	int rb = (want_blocks + tiles - 1) / tiles;
	if (rb > d.N) rb = d.N;
	if (rb < 1) rb = 1;
	int rows = (d.N + rb - 1) / rb;
	rb = (d.N + rows - 1) / rows;
	static bool attr_set = false;
	if (!attr_set) {
		HIPCHK(hipFuncSetAttribute((const void *)k_count<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		attr_set = true;
	}
	prof_begin(c);
	hipLaunchKernelGGL(k_count<256>, dim3(tiles, rb), dim3(BLOCK), lds, c->stream, d, rows);
	prof_end(c, "k_count");
	HIPCHK(hipGetLastError());
	return 0;
}

extern "C" int isg_count_alleles(isg_ctx *c, int32_t *counts)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (launch_count(c)) return 1;
	const int L = c->cfg.L, K = c->cfg.K, A = c->Amax;
	HIPCHK(hipMemcpyAsync(c->cnt_h.data(), c->d.cnt, sizeof(int) * c->cnt_h.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (int k = 0; k < K; k++)
		for (int j = 0; j < L; j++)
			for (int a = 0; a < A; a++) counts[((size_t)k * L + j) * A + a] = c->cnt_h[((size_t)j * A + a) * K + k];
	return 0;
}

/* ---- update_P ---- */
extern "C" int isg_update_P(isg_ctx *c)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	DevView &d = c->d;
	const int L = c->cfg.L, K = c->cfg.K, A = c->Amax;
	if (launch_count(c)) return 1;
	if (is_keyed(c)) {
		int n = K * L, B = 256;
		prof_begin(c);
		hipLaunchKernelGGL(k_pdirich, dim3((n + B - 1) / B), dim3(B), 0, c->stream, d, c->origin, iter_base(c), c->ky[KY_SP]);
		prof_end(c, "k_pdirich");
		HIPCHK(hipGetLastError());
		return 0;
	}
	/* replay: the K*L Dirichlets consume the stream in (k, j) order with data-dependent length
	 * (random.c:167-250), so they are drawn sequentially on the host from the counts */
	HIPCHK(hipMemcpyAsync(c->cnt_h.data(), d.cnt, sizeof(int) * c->cnt_h.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	isg_cursor cur;
	cur.s = c->rng;
	cur.used = 0;
	std::vector<double> tmp(A);
	for (int k = 0; k < K; k++)
		for (int j = 0; j < L; j++) {
			int Aj = c->allelenum[j];
			if (Aj <= 1) continue;
			for (int a = 0; a < Aj; a++) tmp[a] = (double)c->cnt_h[((size_t)j * A + a) * K + k];
			isg_rdirich(&cur, tmp.data(), Aj, &c->freq[((size_t)k * L + j) * A], 1.0);
		}
	c->rng = cur.s;
	c->raw_valid = false;
	return upload_freq(c);
}

/* ---- update_S_POP (host: O(N K) work, sequential MH over clusters) ---- */
static double proposal(isg_ctx *c, const double *s) /* mcmc.c:1630-1648 */
{
	const int N = c->cfg.N, K = c->cfg.K;
	isg_acc acc;
	isg_acc_zero(&acc);
	for (int i = 0; i < N; i++) {
		double temp = 0;
		for (int j = 0; j < K; j++) temp += c->qq[(size_t)i * K + j] * s[j];
		isg_acc_add(&acc, isg_log(isg_pow(temp, c->gen[i] - 1) * (1 - temp)));
	}
	return isg_acc_value(&acc);
}
static double adpt_indp(isg_ctx *c, int *stat_tmp, int stat) /* mcmc.c:1461-1520 */
{
	double tmp = 0, tt;
	if (stat == 0) {
		if (host_next(c) < 0.50) { tmp = 0.0; *stat_tmp = 0; }
		else { tmp = host_next(c); *stat_tmp = 1; }
	} else if (stat == 2) {
		if (host_next(c) < 0.5) { tmp = 1.0; *stat_tmp = 2; }
		else { tmp = host_next(c); *stat_tmp = 1; }
	} else {
		tt = host_next(c);
		if (tt <= 0.05) { tmp = 0.0; *stat_tmp = 0; }
		else if (tt >= 0.95) { tmp = 1.0; *stat_tmp = 2; }
		else { tmp = host_next(c); *stat_tmp = 1; }
	}
	return tmp;
}
static double q_trans(int a, int b) /* mcmc.c:1566-1593 */
{
	if (a == 0) return (b == 0 || b == 1) ? 0.5 : 0.0;
	if (a == 2) return (b == 2 || b == 1) ? 0.5 : 0.0;
	if (a == 1) return (b == 0 || b == 2) ? 0.05 : (b == 1 ? 0.90 : 0.0);
	return 0.0;
}
extern "C" int isg_update_S_POP(isg_ctx *c)
{
	if (c->cfg.mode != 2) return 0;
	const int K = c->cfg.K;
	if (is_keyed(c)) host_seek(c, iter_base(c) + c->ky[KY_OFFS]);
	std::vector<double> tmp(K);
	std::vector<int> tst(K);
	/* proposal(self_rates) only changes when a move is accepted: carry it instead of recomputing */
	double cur_ld = proposal(c, c->S.data());
	for (int j = 0; j < K; j++) {
		for (int i = 0; i < K; i++) { tmp[i] = c->S[i]; tst[i] = c->state[i]; }
		if (c->cfg.back_refl == 1) {
			tmp[j] = host_next(c) * 2 * 0.05 - 0.05;
			tmp[j] += c->S[j];
			if (tmp[j] <= 0.0) tmp[j] = 0.0 - tmp[j];
			else if (tmp[j] >= 1.0) tmp[j] = 1.0 - (tmp[j] - 1.0);
		} else {
			tmp[j] = adpt_indp(c, &tst[j], c->state[j]);
		}
		double new_ld = proposal(c, tmp.data());
		double mh = isg_exp(new_ld - cur_ld);
		if (c->cfg.back_refl == 0) {
			double h = 1.0;
			for (int i = 0; i < K; i++) h *= q_trans(c->state[i], tst[i]) / q_trans(tst[i], c->state[i]);
			mh *= h;
		}
		double thr = (1 > mh) ? mh : 1;
		if (host_next(c) < thr) {
			c->S[j] = tmp[j];
			if (c->cfg.back_refl == 0) c->state[j] = tst[j];
			cur_ld = new_ld;
		}
	}
	return 0;
}

/* ---- update_G ---- */
extern "C" int isg_update_G(isg_ctx *c)
{
	if (c->cfg.mode != 2) return 0;
	HIPCHK(hipSetDevice(c->cfg.device));
	DevView &d = c->d;
	double *d_S = c->d_S;
	HIPCHK(hipMemcpyAsync(d_S, c->S.data(), sizeof(double) * c->cfg.K, hipMemcpyHostToDevice, c->stream));
	isg_wh base = is_keyed(c) ? isg_wh_jump(&c->tab_h, c->origin, iter_base(c) + c->ky[KY_OFFG]) : c->rng;
	prof_begin(c);
	hipLaunchKernelGGL(k_gprop<1024>, dim3(1), dim3(1024), 0, c->stream, d, (const double *)d_S, base, is_keyed(c) ? 1 : 0, c->d_pos);
	prof_end(c, "k_gprop");
	prof_begin(c);
	hipLaunchKernelGGL((k_loglik<256, true>), dim3(d.N), dim3(256), 0, c->stream, d);
	prof_end(c, "k_loglik_pair");
	HIPCHK(hipGetLastError());
	uint64_t used = 0;
	HIPCHK(hipMemcpyAsync(&used, c->d_pos, sizeof(used), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(c->gen.data(), d.gen, sizeof(int) * c->gen.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (!is_keyed(c)) host_advance(c, used);
	return check_dev_err(c);
}

/* ---- update_ZQ ---- */
template <int KMAX>
static void launch_zq(isg_ctx *c, bool chain, isg_wh base, uint64_t pos0, uint64_t stride, int init_flag)
{
	DevView &d = c->d;
	if (chain)
		hipLaunchKernelGGL((k_zq<1024, KMAX, true>), dim3(1), dim3(1024), 0, c->stream, d, base, pos0, stride, init_flag, c->alpha, c->d_pos);
	else
		hipLaunchKernelGGL((k_zq<256, KMAX, false>), dim3(d.N), dim3(256), 0, c->stream, d, base, pos0, stride, init_flag, c->alpha, c->d_pos);
}
extern "C" int isg_update_ZQ(isg_ctx *c, int init_flag)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	const int K = c->cfg.K;
	bool chain = !is_keyed(c);
	isg_wh base = chain ? c->rng : c->origin;
	uint64_t pos0 = chain ? 0 : (init_flag ? c->ky[KY_ZI0] : iter_base(c) + c->ky[KY_OFFZ]);
	uint64_t stride = c->ky[KY_SZ];
	prof_begin(c);
	if (K <= 4) launch_zq<4>(c, chain, base, pos0, stride, init_flag);
	else if (K <= 8) launch_zq<8>(c, chain, base, pos0, stride, init_flag);
	else if (K <= 16) launch_zq<16>(c, chain, base, pos0, stride, init_flag);
	else launch_zq<32>(c, chain, base, pos0, stride, init_flag);
	prof_end(c, chain ? "k_zq_chain" : "k_zq_keyed");
	HIPCHK(hipGetLastError());
	if (chain) {
		uint64_t used = 0;
		HIPCHK(hipMemcpyAsync(&used, c->d_pos, sizeof(used), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		host_advance(c, used);
	}
	return sync_qq_to_host(c);
}

/* ---- update_alpha (host: ordered product over N*K, mcmc.c:1254-1260) ---- */
extern "C" int isg_update_alpha(isg_ctx *c)
{
	const int N = c->cfg.N, K = c->cfg.K;
	if (is_keyed(c)) host_seek(c, iter_base(c) + c->ky[KY_OFFA]);
	isg_cursor cur;
	cur.s = c->rng;
	cur.used = 0;
	double ralpha = isg_rnormal(&cur, c->alpha, 1.0);
	c->rng = cur.s;
	c->raw_valid = false;
	if (ralpha > 0) {
		double mh = 1.0;
		for (int i = 0; i < N; i++)
			for (int m = 0; m < K; m++) {
				double q = c->qq[(size_t)i * K + m], n = (double)c->qqnum[(size_t)i * K + m];
				mh *= isg_pow(q, ralpha + n) / isg_pow(q, n + c->alpha);
			}
		double thr = (1 > mh) ? mh : 1;
		c->alpha = (host_next(c) < thr) ? ralpha : c->alpha;
	}
	return 0;
}

/* ---- cal_lkh ---- */
extern "C" int isg_cal_lkh(isg_ctx *c)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	DevView &d = c->d;
	prof_begin(c);
	hipLaunchKernelGGL((k_loglik<256, false>), dim3(d.N), dim3(256), 0, c->stream, d);
	prof_end(c, "k_loglik_lkh");
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(c->indvlkh.data(), d.indvlkh, sizeof(double) * c->indvlkh.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	isg_acc acc;
	isg_acc_zero(&acc);
	for (int i = 0; i < c->cfg.N; i++) isg_acc_add(&acc, c->indvlkh[i]);
	c->totallkh = isg_acc_value(&acc);
	return 0;
}

extern "C" int isg_iteration(isg_ctx *c)
{
	if (isg_update_P(c)) return 1;
	if (c->cfg.mode == 2) {
		if (isg_update_S_POP(c)) return 1;
		if (isg_update_G(c)) return 1;
	}
	if (isg_update_ZQ(c, 0)) return 1;
	if (isg_update_alpha(c)) return 1;
	if (isg_cal_lkh(c)) return 1;
	c->iter++;
	return 0;
}
extern "C" int isg_iter_advance(isg_ctx *c) { c->iter++; return 0; } /* sweep-by-sweep drivers (tests) */
extern "C" int isg_run(isg_ctx *c, long n)
{
	for (long i = 0; i < n; i++)
		if (isg_iteration(c)) return 1;
	return 0;
}

/* ---- chain init (mcmc.c:471-487, 193-206) ---- */
extern "C" int isg_chain_init(isg_ctx *c, const float *initd)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	const int N = c->cfg.N, K = c->cfg.K;
	c->origin = c->rng;
	c->iter = 0;
	c->alpha = host_next(c) * 10;
	if (c->cfg.mode == 2) {
		isg_cursor cur;
		cur.s = c->rng;
		cur.used = 0;
		for (int i = 0; i < N; i++) {
			double pr = isg_cur_next(&cur);
			int g = isg_rgeom(&cur, pr);
			if (g > 50) g = 50;
			c->gen[i] = g;
		}
		c->rng = cur.s;
		for (int k = 0; k < K; k++) {
			c->S[k] = (double)initd[k];
			if (c->cfg.back_refl == 0) {
				int st = isg_dt_stat(c->S[k]);
				if (st < 0) return fail("The value of selfing rate or inbreeding coefficient is beyond [0,1]!");
				c->state[k] = st;
			}
		}
		HIPCHK(hipMemcpyAsync(c->d.gen, c->gen.data(), sizeof(int) * N, hipMemcpyHostToDevice, c->stream));
	}
	return isg_update_ZQ(c, 1);
}

/* ---- getters / setters ---- */
extern "C" int isg_get_amax(isg_ctx *c, int32_t *a) { *a = c->Amax; return 0; }
extern "C" int isg_get_z(isg_ctx *c, int32_t *z)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	const int N = c->cfg.N, L = c->cfg.L, Lp = c->d.Lp;
	std::vector<uint8_t> h((size_t)N * Lp * 2);
	HIPCHK(hipMemcpyAsync(h.data(), c->d.z, h.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (int i = 0; i < N; i++)
		for (int j = 0; j < L; j++)
			for (int k = 0; k < 2; k++) {
				uint8_t v = h[((size_t)i * Lp + j) * 2 + k];
				z[((size_t)i * L + j) * 2 + k] = (v == 0xff) ? -1 : (int)v;
			}
	return 0;
}
extern "C" int isg_set_z(isg_ctx *c, const int32_t *z)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	const int N = c->cfg.N, L = c->cfg.L, Lp = c->d.Lp;
	std::vector<uint8_t> h((size_t)N * Lp * 2, 0xff);
	for (int i = 0; i < N; i++)
		for (int j = 0; j < L; j++)
			for (int k = 0; k < 2; k++) {
				int v = z[((size_t)i * L + j) * 2 + k];
				h[((size_t)i * Lp + j) * 2 + k] = (v < 0) ? 0xff : (uint8_t)v;
			}
	HIPCHK(hipMemcpy(c->d.z, h.data(), h.size(), hipMemcpyHostToDevice));
	return 0;
}
extern "C" int isg_get_freq(isg_ctx *c, double *f)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (download_freq(c)) return 1;
	memcpy(f, c->freq.data(), sizeof(double) * c->freq.size());
	return 0;
}
extern "C" int isg_set_freq(isg_ctx *c, const double *f)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	memcpy(c->freq.data(), f, sizeof(double) * c->freq.size());
	if (upload_freq(c)) return 1;
	HIPCHK(hipStreamSynchronize(c->stream));
	return 0;
}
extern "C" int isg_get_qq(isg_ctx *c, double *q) { memcpy(q, c->qq.data(), sizeof(double) * c->qq.size()); return 0; }
extern "C" int isg_set_qq(isg_ctx *c, const double *q)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	memcpy(c->qq.data(), q, sizeof(double) * c->qq.size());
	HIPCHK(hipMemcpy(c->d.qq, q, sizeof(double) * c->qq.size(), hipMemcpyHostToDevice));
	return 0;
}
extern "C" int isg_get_qqnum(isg_ctx *c, double *q)
{
	for (size_t i = 0; i < c->qqnum.size(); i++) q[i] = (double)c->qqnum[i];
	return 0;
}
extern "C" int isg_get_generation(isg_ctx *c, int32_t *g) { memcpy(g, c->gen.data(), sizeof(int) * c->gen.size()); return 0; }
extern "C" int isg_set_generation(isg_ctx *c, const int32_t *g)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	memcpy(c->gen.data(), g, sizeof(int) * c->gen.size());
	HIPCHK(hipMemcpy(c->d.gen, g, sizeof(int) * c->gen.size(), hipMemcpyHostToDevice));
	return 0;
}
extern "C" int isg_get_self_rates(isg_ctx *c, double *s) { memcpy(s, c->S.data(), sizeof(double) * c->S.size()); return 0; }
extern "C" int isg_set_self_rates(isg_ctx *c, const double *s) { memcpy(c->S.data(), s, sizeof(double) * c->S.size()); return 0; }
extern "C" int isg_get_state(isg_ctx *c, int32_t *s) { memcpy(s, c->state.data(), sizeof(int) * c->state.size()); return 0; }
extern "C" int isg_get_indvlkh(isg_ctx *c, double *v) { memcpy(v, c->indvlkh.data(), sizeof(double) * c->indvlkh.size()); return 0; }
extern "C" int isg_get_alpha(isg_ctx *c, double *a) { *a = c->alpha; return 0; }
extern "C" int isg_set_alpha(isg_ctx *c, double a) { c->alpha = a; return 0; }
extern "C" int isg_get_totallkh(isg_ctx *c, double *t) { *t = c->totallkh; return 0; }

/* ---- profiling ---- */
extern "C" int isg_profile_enable(isg_ctx *c, int on) { c->prof = on != 0; return 0; }
extern "C" int isg_profile_count(isg_ctx *c) { prof_collect(c); return (int)c->prof_entries.size(); }
extern "C" int isg_profile_get(isg_ctx *c, int idx, char *name, int cap, double *ms, long *n)
{
	if (idx < 0 || idx >= (int)c->prof_entries.size()) return fail("isg_profile_get: index out of range");
	snprintf(name, cap, "%s", c->prof_entries[idx].name.c_str());
	*ms = c->prof_entries[idx].ms;
	*n = c->prof_entries[idx].n;
	return 0;
}
extern "C" int isg_profile_reset(isg_ctx *c) { prof_collect(c); c->prof_entries.clear(); return 0; }

/* ---- Gelman-Rubin on the gathered log-likelihood samples (check_converg.c:100-153) ---- */
extern "C" double isg_gelman_rubin(const double *vec, int numchains, int totrep)
{
	/* the reference derives the per-chain length as totrep / numchains and indexes with it */
	const int rep = totrep / numchains;
	std::vector<double> psii(numchains), S(numchains);
	double psi = 0, W = 0, B = 0;
	for (int i = 0; i < numchains; i++) {
		psii[i] = 0;
		for (int j = 0; j < rep; j++) psii[i] += vec[i * rep + j];
		psii[i] = psii[i] / rep;
		psi = psi + psii[i];
	}
	psi = psi / numchains;
	for (int i = 0; i < numchains; i++) {
		S[i] = 0;
		for (int j = 0; j < rep; j++) S[i] += (vec[i * rep + j] - psii[i]) * (vec[i * rep + j] - psii[i]);
		S[i] = S[i] / (rep - 1);
		W += S[i];
	}
	W = W / numchains;
	for (int i = 0; i < numchains; i++) B += (psii[i] - psi) * (psii[i] - psi);
	B = (B * rep) / (numchains - 1);
	double V = (W * (rep - 1)) / rep + B / rep;
	return V / W;
}
