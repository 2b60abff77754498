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
#include <array>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include <new>
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
	const unsigned *rankwave; /* [N][nwv]: used loci of individual i before locus 64*w */
	int nwv;
	double *freq;
	float *freqf;        /* single precision copy of freq, [L][Amax][KPF]: pre-filter of the Z draws */
	int KPF;
	/* log-likelihood terms as functions of (locus, alleles, cluster[, generation]) -- rebuilt when freq changes:
	 *   lftab [L][Amax][K]                log freq                          (copies whose partner sits in another cluster)
	 *   lltab [50][L][Amax][Amax][K]      log genofreq(generation, f0, f1)  (both copies in cluster k); null if too large */
	double *lftab, *lltab;
	/* mode 2: the same terms as the two integers isg_acc2 sums, {rint(v 2^20), rint((v - h 2^-20) 2^51)}, locus innermost, ONE allocation addressed
	 * by 32-bit entry numbers: [0] = {0, 0}; 1 + ((g - 1) A A K + (a0 A + a1) K + k) Lp + j = log genofreq(g); lli_F + (a K + k) Lp + j = log freq.
	 * h = INT_MIN marks a term isg_acc2 would flag (infinite, NaN, |v| >= 1024): m holds its flag bits. */
	int2 *lli;
	unsigned lli_F;
	const double *tape;  /* replay schedule: the uniforms of the ZQ phase in stream order */
	unsigned long long tape_len;
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

struct PolyCtx;
struct InbreedCtx;
struct ResolveCtx;
struct PDevCtx;
struct SpecCtx;
/* Host buffers that asynchronous copies read or write start on a page of their own and end on one: the runtime pins pageable memory
 * page by page while a copy is in flight (and hipHostRegister pins whole pages), so two buffers of two chains that share a page -- chains
 * run by threads of one process allocate theirs side by side in the heap -- would pin and release that page under each other. */
template <class T>
struct PageAlloc {
	typedef T value_type;
	PageAlloc() {}
	template <class U> PageAlloc(const PageAlloc<U> &) {}
	T *allocate(size_t n)
	{
		void *p = nullptr;
		const size_t bytes = (n * sizeof(T) + 4095) & ~(size_t)4095;
		if (posix_memalign(&p, 4096, bytes ? bytes : 4096)) throw std::bad_alloc();
		return (T *)p;
	}
	void deallocate(T *p, size_t) { free(p); }
	template <class U> bool operator==(const PageAlloc<U> &) const { return true; }
	template <class U> bool operator!=(const PageAlloc<U> &) const { return false; }
};
template <class T> using hvec = std::vector<T, PageAlloc<T>>;

struct isg_ctx {
	isg_config cfg;
	ResolveCtx *rs = nullptr; /* replay update_ZQ: start positions resolved block-wise (isg_resolve_hip.inc) */
	SpecCtx *zspec = nullptr; /* replay update_ZQ: start positions resolved from intervals of shapes (isg_spec_hip.inc) */
	PDevCtx *pdev = nullptr;  /* replay update_P on the device: the Dirichlets' start positions resolved by the walk engine (isg_walk_hip.inc) */
	PolyCtx *poly = nullptr; /* ploidy 4 state (isg_poly_hip.inc) */
	InbreedCtx *inb = nullptr; /* mode 4 state (isg_modes_hip.inc) */
	DevView d;
	int Amax;
	hipStream_t stream;
	/* host mirrors (reference layouts) */
	std::vector<int> allelenum;
	hvec<double> freq;   /* [K][L][Amax] */
	hvec<double> qq;     /* [N][K] */
	hvec<int> qqnum;     /* [N][K] */
	hvec<int> gen;       /* [N] */
	hvec<double> S;      /* [K] */
	hvec<int> state;     /* [K] */
	hvec<double> indvlkh;
	hvec<int> cnt_h;     /* device order [L][Amax][K] */
	hvec<double> freq_stage; /* device order [L][Amax][KP] */
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
	double *d_Fprop; /* mode 5: proposed coefficients */
	double *d_tape;
	uint64_t tape_cap, nvalid_total;
	void *d_coop;
	int coop; /* 1: several workgroups per individual in the replay-schedule ZQ kernel */
	int spec; /* 1: ... with the next individual's Z drawn ahead for the likely start positions (INSTRUCT_ZQ_SPEC=0 disables) */
	int xcd;  /* 1: try to place the cooperating workgroups on one XCD (INSTRUCT_ZQ_XCD=1 enables) */
	int pipe; /* 1: draw waves + one control wave per workgroup (k_zq_pipe; INSTRUCT_ZQ_PIPE=0 disables) */
	int pipe_xcd; /* 1: its workgroups on one XCD when they fit (INSTRUCT_ZQ_PIPE_XCD=0 disables) */
	unsigned long long *d_pipe = nullptr; /* its granules (one line per publishing wave) */
	unsigned long long *d_spop = nullptr; /* k_spop_tree: limbs of the 2^K exact sums */
	bool counted = false;                 /* in g_live_ctx */
	double host_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* INSTRUCT_HOST_TIMING=1: seconds in the stages of replay update_P's host side */
	long host_n = 0;
	bool host_timing = false;
	std::vector<void *> pinned;           /* host vectors registered with the runtime (pin_host): the per-sweep copies of update_P */
	hipEvent_t ev_tape = nullptr;         /* ... the tape has arrived (update_P_ahead) */
	bool ahead_valid = false;             /* counts and tape of the NEXT update_P were requested at the end of update_alpha ... */
	isg_wh ahead_rng;                     /* ... for this stream position (Z unchanged since: every writer of Z clears the flag) */
	uint64_t ahead_ngamma = 0;
	hipEvent_t ev_cnt = nullptr;          /* replay update_P: the counts have arrived (the tape is still on its way) */
	double *htape = nullptr;              /* replay update_P: the host loop's uniforms (host_tape_begin); pinned: a 2 MB copy per sweep */
	uint64_t htape_cap = 0;
	std::vector<double> pshape;           /* ... the shapes of its gammas in stream order ... */
	std::vector<std::array<double, 5>> pcoef; /* ... and rgamma2's shape-only constants (HostGammaCoef) */
	uint64_t htape_len = 0;
	int host_tape = 1;                    /* INSTRUCT_HOST_TAPE=0: the host loop steps the generator itself */
	int spop_tree = 1;                    /* INSTRUCT_SPOP_TREE=0: the one-workgroup k_spop always */
	size_t pipe_cap = 0;
	double *d_qqsave = nullptr;           /* qq as it was when a cooperative update_ZQ started (it is both input and output) */
	int test_abort = 0;                   /* INSTRUCT_ZQ_TEST_ABORT=n: the n-th cooperative sweep is treated as aborted (tests) */
	long zq_fallbacks = 0;                /* sweeps redone by the single-workgroup kernel */
	int *d_state;
	double *d_ratios, *d_total;
	hvec<double> ratios_h;
	/* which host mirrors are current */
	bool h_qq, h_gen, h_S, h_lkh;
	/* CHAIN running means kept on the device (isg_store_*): qq, qq2 [N][K]; indvlkh, gen, gen2 [N]; freq, freq2 in the
	 * device order of d.freq */
	double *st_qq = nullptr, *st_qq2 = nullptr, *st_lkh = nullptr, *st_gen = nullptr, *st_gen2 = nullptr, *st_freq = nullptr, *st_freq2 = nullptr;
	long st_step = 0;
	bool st_on = false;
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

/* sum of x over the 64 lanes, as a wave-uniform value: rotations inside the rows of 16 lanes (every lane of a row
 * ends up with the row's sum), then the four rows through SGPRs.  All lanes must be active. */
__device__ __forceinline__ unsigned wave_sum_u32(unsigned x)
{
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xf, 0xf, false); /* row_ror:1 */
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xf, 0xf, false); /* row_ror:2 */
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xf, 0xf, false); /* row_ror:4 */
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false); /* row_ror:8 */
	return (unsigned)__builtin_amdgcn_readlane((int)x, 0) + (unsigned)__builtin_amdgcn_readlane((int)x, 16) +
	       (unsigned)__builtin_amdgcn_readlane((int)x, 32) + (unsigned)__builtin_amdgcn_readlane((int)x, 48);
}


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
template <int BLOCK, int LPT>
__global__ void __launch_bounds__(BLOCK) k_count(DevView d, int rows_per_block)
{
	extern __shared__ unsigned lds_cnt[];
	const int C = LPT * d.Amax * d.K;
	const int t = threadIdx.x;
	for (int c = 0; c < C; c++) lds_cnt[c * BLOCK + t] = 0;
	const int J0 = blockIdx.x * BLOCK * LPT;
	const int j0 = J0 + t * LPT;
	const int r0 = blockIdx.y * rows_per_block;
	int r1 = r0 + rows_per_block;
	if (r1 > d.N) r1 = d.N;
	const size_t rowb = (size_t)d.Lp * 2;
	if (j0 < d.Lp) {
		for (int i = r0; i < r1; i++) {
			unsigned long long gb, zb;
			if (LPT == 4) {
				const uint2 g = *(const uint2 *)(d.geno + (size_t)i * rowb + (size_t)j0 * 2);
				const uint2 zz = *(const uint2 *)(d.z + (size_t)i * rowb + (size_t)j0 * 2);
				gb = ((unsigned long long)g.y << 32) | g.x;
				zb = ((unsigned long long)zz.y << 32) | zz.x;
			} else {
				gb = *(const unsigned short *)(d.geno + (size_t)i * rowb + (size_t)j0 * 2);
				zb = *(const unsigned short *)(d.z + (size_t)i * rowb + (size_t)j0 * 2);
			}
#pragma unroll
			for (int l = 0; l < LPT; l++) {
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
	if (tile_loci > BLOCK * LPT) tile_loci = BLOCK * LPT;
	if (tile_loci < 0) tile_loci = 0;
	const int E = tile_loci * per_locus;
	for (int e = t; e < E; e += BLOCK) {
		int jl = e / per_locus, rem = e - jl * per_locus; /* rem = a*K + k */
		unsigned v = lds_cnt[((jl % LPT) * per_locus + rem) * BLOCK + jl / LPT];
		if (v) atomicAdd(&d.cnt[(size_t)J0 * per_locus + e], (int)v);
	}
}

/* fallback for very large Amax*K (tile counters do not fit in LDS): one global atomic per allele copy */
__global__ void k_count_atomic(DevView d)
{
	const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= (size_t)d.N * d.Lp) return;
	const unsigned short g = ((const unsigned short *)d.geno)[id], z = ((const unsigned short *)d.z)[id];
	const unsigned a0 = g & 0xff, a1 = g >> 8;
	if (a0 == 0xff) return;
	const size_t j = id % d.Lp;
	atomicAdd(&d.cnt[(j * d.Amax + a0) * d.K + (z & 0xff)], 1);
	atomicAdd(&d.cnt[(j * d.Amax + a1) * d.K + (z >> 8)], 1);
}

/* ------------------------------------------------------------------------------------------ */
/* k_gprop: update_G proposals (mcmc.c:1060-1084) and stream positions                          */
/* ------------------------------------------------------------------------------------------ */
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_gprop(DevView d, const double *S, isg_wh base, int keyed, uint64_t *pos_out)
{
	__shared__ unsigned sm[BLOCK / 64 + 1];
	__shared__ double Ssh[ISG_KCAP];
	const bool indiv = (d.mode == 3); /* mode 3: S has one selfing rate per individual (mcmc.c:1069-1070) */
	if (!indiv && threadIdx.x < (unsigned)d.K) Ssh[threadIdx.x] = S[threadIdx.x];
	__syncthreads();
	unsigned running = 0;
	for (int i0 = 0; i0 < d.N; i0 += BLOCK) {
		int i = i0 + threadIdx.x;
		double selfing = 0;
		int stat = 0;
		if (i < d.N) {
			if (indiv) selfing = S[i];
			else for (int k = 0; k < d.K; k++) selfing += d.qq[(size_t)i * d.K + k] * Ssh[k];
			stat = isg_dt_stat(selfing);
			if (stat < 0) { /* mcmc.c:1540-1541: the reference prints the value and exits; the host does that after this launch */
				if ((atomicOr(d.err, 2u) & 2u) == 0u) *(double *)(d.err + 2) = selfing; /* the first offender's value */
				stat = 1;
			}
		}
		unsigned flag = (i < d.N && stat == 1) ? 1u : 0u, tot, pre;
		pre = block_excl_scan<BLOCK>(flag, sm, &tot);
		if (i < d.N) {
			uint64_t pos = keyed ? 2ull * (uint64_t)i : (uint64_t)i + running + pre;
			isg_cursor c;
			c.s = isg_wh_jump(d.tab, base, pos);
			c.used = 0;
			c.tape = nullptr;
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
/* block reduction of the specialised accumulator (two 64-bit integer sums + flags) */
template <int BLOCK>
__device__ __forceinline__ isg_acc2 block_reduce_acc2(const isg_acc2 &a, long long *sm /* [BLOCK/64][3] */)
{
	const int NW = BLOCK / 64;
	long long hi = a.hi, lo = a.lo;
	unsigned fl = a.flags;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		hi += __shfl_down(hi, o, 64);
		lo += __shfl_down(lo, o, 64);
		fl |= __shfl_down(fl, o, 64);
	}
	const unsigned w = threadIdx.x >> 6;
	if (lane_id() == 0) {
		sm[w * 3 + 0] = hi;
		sm[w * 3 + 1] = lo;
		sm[w * 3 + 2] = (long long)fl;
	}
	__syncthreads();
	isg_acc2 r;
	r.hi = 0;
	r.lo = 0;
	r.flags = 0;
	for (int i = 0; i < NW; i++) {
		r.hi += sm[i * 3 + 0];
		r.lo += sm[i * 3 + 1];
		r.flags |= (unsigned)sm[i * 3 + 2];
	}
	__syncthreads();
	return r;
}

/* PAIR = true : update_G -- both generations in one pass over the row, MH accept at the end
 * PAIR = false: cal_lkh  -- indvlkh[i] (mode 2: current generation; mode 1: log_ld_noselfing_indv)
 *
 * Every locus contributes either log(genofreq(.., g)) (both copies assigned to the same cluster,
 * mcmc.c:1752-1758, or -y 0, :1739-1748) or log f0 + log f1 (+ log 2) (:1760-1767, mode 1 :1881-1886).
 * To keep the wave convergent each lane takes exactly two logarithms per locus -- of
 * (genofreq(g_cur), genofreq(g_prop)) or of (f0, f1) -- and sorts the results into three
 * order-independent accumulators: current-generation terms, proposed-generation terms, common terms. */
/*
 * The terms log_ld_indv adds per locus (mcmc.c:1735-1770) only depend on (locus, allele pair, cluster pair,
 * generation), and freq changes once per iteration: N x L evaluations of genofreq + log become K L Amax^2 x 50
 * table entries (the same expressions, evaluated once: genofreq's loop over the generations is exactly the
 * recurrence below, mcmc.c:1683-1703) and N x L lookups.  One lane per (locus, a0, a1, cluster).
 */
__global__ void k_lltab(DevView d)
{
	const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t per = (size_t)d.L * d.Amax * d.Amax * d.K;
	if (id >= per) return;
	const int k = (int)(id % d.K);
	const int a1 = (int)((id / d.K) % d.Amax), a0 = (int)((id / d.K / d.Amax) % d.Amax), j = (int)(id / d.K / d.Amax / d.Amax);
	const double f0 = d.freq[((size_t)j * d.Amax + a0) * d.KP + k], f1 = d.freq[((size_t)j * d.Amax + a1) * d.KP + k];
	if (a1 == 0) d.lftab[((size_t)j * d.Amax + a0) * d.K + k] = isg_log(f0);
	if (!d.lltab || d.mode != 2) return;
	if (a0 == a1) { /* isg_genofreq(hom): result after g - 1 rounds of the loop */
		double result = f0 * f0, temp = 2 * f0 * (1 - f0);
		for (int g = 1; g <= 50; g++) {
			d.lltab[(size_t)(g - 1) * per + id] = isg_log(result);
			temp /= 2;
			result += temp / 2;
		}
	} else {
		for (int g = 1; g <= 50; g++) d.lltab[(size_t)(g - 1) * per + id] = isg_log(2 * f0 * f1 * isg_scalbn(1.0, -(g - 1)));
	}
}

/* the terms of k_lltab as isg_acc2's integers, locus innermost (one lane per (allele pair, cluster, locus): consecutive lanes write consecutive entries) */
__device__ __forceinline__ int2 lli_entry(double v)
{
	if (!(v > -1024.0 && v < 1024.0)) {
		const uint64_t u = isg_d2u(v);
		return make_int2((int)0x80000000u, (u == 0x7ff0000000000000ULL) ? 1 : (u == 0xfff0000000000000ULL) ? 2 : 4);
	}
	const double hs = __builtin_rint(v * 1048576.0);
	const double lo = isg_fma(hs, -0x1p-20, v); /* exact */
	return make_int2((int)hs, (int)__builtin_rint(lo * 0x1p51));
}
__global__ void k_lltab_int(DevView d)
{
	const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t AAK = (size_t)d.Amax * d.Amax * d.K;
	if (id == 0) d.lli[0] = make_int2(0, 0);
	if (id >= AAK * d.L) return;
	const int j = (int)(id % d.L), combo = (int)(id / d.L);
	const int k = combo % d.K, a1 = (combo / d.K) % d.Amax, a0 = combo / d.K / d.Amax;
	const double f0 = d.freq[((size_t)j * d.Amax + a0) * d.KP + k], f1 = d.freq[((size_t)j * d.Amax + a1) * d.KP + k];
	if (a1 == 0) d.lli[d.lli_F + ((size_t)a0 * d.K + k) * d.Lp + j] = lli_entry(isg_log(f0));
	int2 *out = d.lli + 1 + (size_t)combo * d.Lp + j;
	const size_t gstride = AAK * d.Lp;
	if (a0 == a1) { /* isg_genofreq(hom): result after g - 1 rounds of the loop */
		double result = f0 * f0, temp = 2 * f0 * (1 - f0);
		for (int g = 1; g <= 50; g++) {
			out[(size_t)(g - 1) * gstride] = lli_entry(isg_log(result));
			temp /= 2;
			result += temp / 2;
		}
	} else {
		for (int g = 1; g <= 50; g++) out[(size_t)(g - 1) * gstride] = lli_entry(isg_log(2 * f0 * f1 * isg_scalbn(1.0, -(g - 1))));
	}
}

/* log_ld_indv (mode 2, -y 1) from the integer tables: straight-line per locus -- two entry numbers (same cluster: the genotype term at the current and,
 * PAIR, the proposed generation; different clusters: the two allele terms; locus unused: entry 0), two 8-byte loads, 64-bit integer adds.  The sums
 * are those k_loglik_tab forms (integer addition: any grouping), so the values are the same bits.
 *   current  = sum A + sum over mixed loci of B + n2 log 2         A = genotype term(gc) | log f0,   B = genotype term(gp) | log f1
 *   proposed = sum B + sum over mixed loci of A + n2 log 2         (PAIR = false: B = 0 | log f1, indvlkh = sum A + sum B + n2 log 2) */
template <int BLOCK, bool PAIR>
__global__ void __launch_bounds__(BLOCK) k_loglik_int(DevView d)
{
	constexpr int NW = BLOCK / 64, NS = PAIR ? 10 : 6;
	__shared__ long long sm[NW * NS];
	const int i = blockIdx.x;
	const int gc = d.gen[i];
	int gp = gc;
	if (PAIR) {
		gp = d.genprop[i];
		if (gp == gc) return;
	}
	const unsigned Lp = (unsigned)d.Lp, A = (unsigned)d.Amax, K = (unsigned)d.K;
	const unsigned gstride = A * A * K * Lp;
	const unsigned baseC = 1u + (unsigned)(gc - 1) * gstride, dCP = (unsigned)(gp - gc) * gstride /* (mod 2^32) */, baseF = d.lli_F;
	const int2 *T = d.lli;
	const size_t rowb = (size_t)d.Lp * 2;
	const uint8_t *grow = d.geno + (size_t)i * rowb;
	const uint8_t *zrow = d.z + (size_t)i * rowb;
	long long sAh = 0, sAm = 0, sBh = 0, sBm = 0, mAh = 0, mAm = 0, mBh = 0, mBm = 0;
	unsigned n2 = 0, flags = 0;
	/* consecutive lanes take consecutive loci (a lane's ISG_LPT loci lie BLOCK apart): a table row is locus innermost, so the lanes that read the
	 * same row read one contiguous stretch of it -- a load instruction touches a few lines per row instead of a line per lane */
	for (unsigned jb = 0; jb < Lp; jb += BLOCK * ISG_LPT) {
		unsigned gw[ISG_LPT], zw[ISG_LPT];
#pragma unroll
		for (int l = 0; l < ISG_LPT; l++) {
			const unsigned jl = jb + (unsigned)l * BLOCK + threadIdx.x;
			gw[l] = (jl < Lp) ? *(const unsigned short *)(grow + (size_t)jl * 2) : 0xffffu;
			zw[l] = (jl < Lp) ? *(const unsigned short *)(zrow + (size_t)jl * 2) : 0u;
		}
#pragma unroll
		for (int l = 0; l < ISG_LPT; l++) {
			const unsigned a0 = gw[l] & 0xff, a1 = gw[l] >> 8;
			const unsigned z0 = zw[l] & 0xff, z1 = zw[l] >> 8;
			const bool valid = (a0 != 0xff), same = (z0 == z1);
			const unsigned j = jb + (unsigned)l * BLOCK + threadIdx.x;
			const unsigned og = baseC + ((a0 * A + a1) * K + z0) * Lp + j;
			const unsigned o0 = baseF + (a0 * K + z0) * Lp + j, o1 = baseF + (a1 * K + z1) * Lp + j;
			unsigned offA = same ? og : o0;
			unsigned offB = PAIR ? (same ? og + dCP : o1) : (same ? 0u : o1);
			offA = valid ? offA : 0u;
			offB = valid ? offB : 0u;
			int2 ea = T[offA], eb = T[offB];
			if (ea.x == (int)0x80000000u || eb.x == (int)0x80000000u) { /* (never with frequencies in (0, 1); the terms isg_acc2 flags) */
				/* bits 0..7: flags of the current generation's total, 8..15: of the proposed one's (each takes the terms listed above) */
				if (ea.x == (int)0x80000000u) { flags |= (unsigned)ea.y | ((PAIR && !same) ? (unsigned)ea.y << 8 : 0u); ea = make_int2(0, 0); }
				if (eb.x == (int)0x80000000u) { flags |= (PAIR ? (unsigned)eb.y << 8 : 0u) | ((!PAIR || !same) ? (unsigned)eb.y : 0u); eb = make_int2(0, 0); }
			}
			sAh += ea.x; sAm += ea.y;
			sBh += eb.x; sBm += eb.y;
			if (PAIR) {
				mAh += same ? 0 : ea.x; mAm += same ? 0 : ea.y;
				mBh += same ? 0 : eb.x; mBm += same ? 0 : eb.y;
			}
			n2 += (valid && !same && a0 != a1) ? 1u : 0u;
		}
	}
	/* block sums */
	long long v[NS];
	v[0] = sAh; v[1] = sAm; v[2] = sBh; v[3] = sBm; v[4] = (long long)n2; v[5] = (long long)flags;
	if (PAIR) { v[6] = mAh; v[7] = mAm; v[8] = mBh; v[9] = mBm; }
#pragma unroll
	for (int q = 0; q < NS; q++) {
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const long long w = __shfl_down(v[q], o, 64);
			v[q] = (q == 5) ? (v[q] | w) : v[q] + w;
		}
		if (lane_id() == 0) sm[(threadIdx.x >> 6) * NS + q] = v[q];
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		long long r[NS];
		for (int q = 0; q < NS; q++) {
			r[q] = 0;
			for (int w = 0; w < NW; w++) r[q] = (q == 5) ? (r[q] | sm[w * NS + q]) : r[q] + sm[w * NS + q];
		}
		const int2 l2 = lli_entry(isg_log(2.0));
		isg_acc2 tc;
		tc.flags = (uint32_t)r[5] & 0xffu;
		if (PAIR) {
			isg_acc2 tp;
			tp.flags = ((uint32_t)r[5] >> 8) & 0xffu;
			tc.hi = r[0] + r[8] + r[4] * l2.x; tc.lo = r[1] + r[9] + r[4] * l2.y;
			tp.hi = r[2] + r[6] + r[4] * l2.x; tp.lo = r[3] + r[7] + r[4] * l2.y;
			const double lc = isg_acc2_value(&tc), lp = isg_acc2_value(&tp);
			const double mh = isg_exp(lp - lc);
			const double thr = (1 > mh) ? mh : 1; /* MIN2(1, mhratio), mcmc.h:10 */
			if (d.uacc[i] < thr) d.gen[i] = gp;
		} else {
			tc.hi = r[0] + r[2] + r[4] * l2.x; tc.lo = r[1] + r[3] + r[4] * l2.y;
			d.indvlkh[i] = isg_acc2_value(&tc);
		}
	}
}

/* log_ld_indv from the tables (same values, same exact accumulation as k_loglik) */
template <int BLOCK, bool PAIR>
__global__ void __launch_bounds__(BLOCK) k_loglik_tab(DevView d)
{
	__shared__ long long sm[(BLOCK / 64) * 3];
	const int i = blockIdx.x;
	int gp = 1, gc;
	if (PAIR) {
		gp = d.genprop[i];
		gc = d.gen[i];
		if (gp == gc) return;
	} else {
		gc = (d.mode == 2) ? d.gen[i] : (d.mode == 4 ? 1 : -1); /* mode 4: one slot holding log genofreq_inbreedcoff */
	}
	const double log2c = isg_log(2.0);
	isg_acc2 accC, accP, accX;
	isg_acc2_zero(&accC);
	isg_acc2_zero(&accP);
	isg_acc2_zero(&accX);
	const size_t rowb = (size_t)d.Lp * 2, per = (size_t)d.L * d.Amax * d.Amax * d.K;
	const uint8_t *grow = d.geno + (size_t)i * rowb;
	const uint8_t *zrow = d.z + (size_t)i * rowb;
	const double *TC = (gc >= 1) ? d.lltab + (size_t)(gc - 1) * per : d.lltab, *TP = d.lltab + (size_t)(gp - 1) * per;
	for (int j0 = threadIdx.x * ISG_LPT; j0 < d.Lp; j0 += BLOCK * ISG_LPT) {
		const uint2 g = *(const uint2 *)(grow + (size_t)j0 * 2);
		const uint2 zz = *(const uint2 *)(zrow + (size_t)j0 * 2);
		unsigned long long gb = ((unsigned long long)g.y << 32) | g.x, zb = ((unsigned long long)zz.y << 32) | zz.x;
#pragma unroll
		for (int l = 0; l < ISG_LPT; l++) {
			unsigned a0 = (unsigned)(gb >> (16 * l)) & 0xff, a1 = (unsigned)(gb >> (16 * l + 8)) & 0xff;
			unsigned z0 = (unsigned)(zb >> (16 * l)) & 0xff, z1 = (unsigned)(zb >> (16 * l + 8)) & 0xff;
			const bool valid = (a0 != 0xff);
			if (!valid) { a0 = a1 = 0; z0 = z1 = 0; }
			const int j = (j0 + l < d.L) ? j0 + l : 0;
			const bool geno_term = (gc >= 0 && z0 == z1);
			if (geno_term) {
				const size_t e = (((size_t)j * d.Amax + a0) * d.Amax + a1) * d.K + z0;
				isg_acc2_add(&accC, valid ? TC[e] : 0.0);
				if (PAIR) isg_acc2_add(&accP, valid ? TP[e] : 0.0);
			} else {
				const double L0 = d.lftab[((size_t)j * d.Amax + a0) * d.K + z0], L1 = d.lftab[((size_t)j * d.Amax + a1) * d.K + z1];
				isg_acc2_add(&accX, valid ? L0 : 0.0);
				isg_acc2_add(&accX, valid ? L1 : 0.0);
				isg_acc2_add(&accX, (valid && a0 != a1) ? log2c : 0.0);
			}
		}
	}
	isg_acc2 rc, rp, rx;
	rc = block_reduce_acc2<BLOCK>(accC, sm);
	rx = block_reduce_acc2<BLOCK>(accX, sm);
	if (PAIR) rp = block_reduce_acc2<BLOCK>(accP, sm);
	if (threadIdx.x == 0) {
		isg_acc2 tc = rc;
		isg_acc2_merge(&tc, &rx);
		double lc = isg_acc2_value(&tc);
		if (PAIR) {
			isg_acc2 tp = rp;
			isg_acc2_merge(&tp, &rx);
			double lp = isg_acc2_value(&tp);
			double mh = isg_exp(lp - lc);
			double thr = (1 > mh) ? mh : 1; /* MIN2(1, mhratio), mcmc.h:10 */
			if (d.uacc[i] < thr) d.gen[i] = gp;
		} else {
			d.indvlkh[i] = lc;
		}
	}
}

template <int BLOCK, bool PAIR>
__global__ void __launch_bounds__(BLOCK) k_loglik(DevView d)
{
	__shared__ long long sm[(BLOCK / 64) * 3];
	__shared__ double qsh[ISG_KCAP];
	const int i = blockIdx.x;
	int gp = 1, gc;
	if (PAIR) {
		gp = d.genprop[i];
		gc = d.gen[i];
		if (gp == gc) return; /* identical sums: ratio is exactly 1, accepted, generation unchanged */
	} else {
		gc = (d.mode == 2 || d.mode == 3) ? d.gen[i] : -1;
	}
	const bool expect = (gc >= 0 && d.type_freq == 0); /* -y 0: expected frequencies under qq */
	if (expect) {
		if (threadIdx.x < (unsigned)d.K) qsh[threadIdx.x] = d.qq[(size_t)i * d.K + threadIdx.x];
		__syncthreads();
	}
	const double log2c = isg_log(2.0);
	isg_acc2 accC, accP, accX; /* current-gen terms, proposed-gen terms, generation-independent terms */
	isg_acc2_zero(&accC);
	isg_acc2_zero(&accP);
	isg_acc2_zero(&accX);
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
			const bool valid = (a0 != 0xff);
			if (!valid) { a0 = a1 = 0; z0 = z1 = 0; }
			const int j = (j0 + l < d.L) ? j0 + l : 0;
			const double *F0 = d.freq + ((size_t)j * d.Amax + a0) * d.KP;
			const double *F1 = d.freq + ((size_t)j * d.Amax + a1) * d.KP;
			double f0, f1;
			if (expect) {
				f0 = 0;
				f1 = 0;
				for (int m = 0; m < d.K; m++) f0 += F0[m] * qsh[m];
				for (int m = 0; m < d.K; m++) f1 += F1[m] * qsh[m];
			} else {
				f0 = F0[z0];
				f1 = F1[z1];
			}
			const bool geno_term = expect || (gc >= 0 && z0 == z1);
			const bool hom = (a0 == a1);
			double v0 = f0, v1 = f1;
			if (gc >= 0) { /* wave-uniform */
				const double gfc = isg_genofreq(hom, f0, f1, gc), gfp = PAIR ? isg_genofreq(hom, f0, f1, gp) : 1.0;
				v0 = geno_term ? gfc : f0;
				v1 = geno_term ? gfp : f1;
			}
			const double L0 = isg_log(v0), L1 = isg_log(v1);
			isg_acc2_add(&accC, (valid && geno_term) ? L0 : 0.0);
			if (PAIR) isg_acc2_add(&accP, (valid && geno_term) ? L1 : 0.0);
			isg_acc2_add(&accX, (valid && !geno_term) ? L0 : 0.0);
			isg_acc2_add(&accX, (valid && !geno_term) ? L1 : 0.0);
			isg_acc2_add(&accX, (valid && !geno_term && !hom) ? log2c : 0.0);
		}
	}
	isg_acc2 rc, rp, rx;
	rc = block_reduce_acc2<BLOCK>(accC, sm);
	rx = block_reduce_acc2<BLOCK>(accX, sm);
	if (PAIR) rp = block_reduce_acc2<BLOCK>(accP, sm);
	if (threadIdx.x == 0) {
		isg_acc2 tc = rc;
		isg_acc2_merge(&tc, &rx);
		double lc = isg_acc2_value(&tc);
		if (PAIR) {
			isg_acc2 tp = rp;
			isg_acc2_merge(&tp, &rx);
			double lp = isg_acc2_value(&tp);
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
/*
 * disc_unif (random.c:403-430) without the K divisions.  The reference divides the running sums by
 * the last one and returns the bucket i with v[i-1] < x <= v[i]; as the v[i] are non-decreasing that
 * bucket is  #{ m <= K-2 : fl(cum[m]/tot) < x }.  The sign of d = fma(x, tot, -cum[m]) decides the
 * comparison whenever |d| exceeds a 2^-48 relative guard band; inside the band (probability ~1e-14
 * per draw) the exact division is evaluated, so the result is always the reference's.
 */
template <int KMAX>
__device__ __forceinline__ int bucket_fast(double x, const double (&cum)[KMAX], double tot, int K)
{
	const double thr = (x * tot) * 0x1p-48;
	bool amb = !(tot > 1e-280 && tot < 1e280);
	int z = 0;
#pragma unroll
	for (int m = 0; m < KMAX - 1; m++) {
		if (m < K - 1) {
			double dd = isg_fma(x, tot, -cum[m]);
			z += (dd > thr) ? 1 : 0;
			amb |= !(dd > thr || dd < -thr);
		}
	}
	if (amb) {
		z = 0;
#pragma unroll
		for (int m = 0; m < KMAX - 1; m++)
			if (m < K - 1) z += (cum[m] / tot < x) ? 1 : 0;
	}
	return z;
}

/* branch-free core of bucket_fast: bucket from the guard-banded comparisons, *amb set when any
 * comparison fell inside the band (the caller then re-evaluates that draw with exact divisions) */
template <int KMAX>
__device__ __forceinline__ int bucket_core(double x, const double (&cum)[KMAX], double tot, int K, bool *amb)
{
	const double p = x * tot, thr = p * 0x1p-48;
	bool a = !(tot > 1e-280 && tot < 1e280);
	int z = 0;
#pragma unroll
	for (int m = 0; m < KMAX - 1; m++) {
		if (m < K - 1) {
			const double dd = isg_fma(x, tot, -cum[m]);
			z += (dd > thr) ? 1 : 0;
			a |= !(dd > thr || dd < -thr);
		}
	}
	*amb = a;
	return z;
}
template <int KMAX>
__device__ __forceinline__ int bucket_exact(double x, const double (&cum)[KMAX], double tot, int K)
{
	int z = 0;
#pragma unroll
	for (int m = 0; m < KMAX - 1; m++)
		if (m < K - 1) z += (cum[m] / tot < x) ? 1 : 0;
	return z;
}

/* running sums cum[m] = sum_{m' <= m} qq[m'] * freq[m'][j][a] in the reference's order (mcmc.c:1146-1147) */
template <int KMAX>
__device__ __forceinline__ double weights(const double *__restrict__ F, const double (&q)[KMAX], double (&cum)[KMAX], int K)
{
	double run = 0;
#pragma unroll
	for (int m = 0; m < KMAX; m += 2) {
		if (m < K) {
			const double2 f = *(const double2 *)(F + m); /* rows are KP = even(K) doubles, 16 B aligned */
			double w = q[m] * f.x;
			run = (m == 0) ? w : run + w;
			cum[m] = run;
			if (m + 1 < K) {
				run = run + q[m + 1] * f.y;
				cum[m + 1] = run;
			}
		}
	}
	return run;
}

#ifdef ISG_STAMPS
__device__ unsigned long long g_stamps[4096 * 8];
#define STAMP(i, k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && (i) < 4096) g_stamps[(i) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
__device__ unsigned long long g_stamps2[2048 * 8]; /* sub-stages of the resolver's walk (STAMPW) */
extern "C" int isg_diag_stamps2(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps2), sizeof(unsigned long long) * 2048 * 8); }
extern "C" int isg_diag_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 4096 * 8); }
#else
#define STAMP(i, k) do { } while (0)
#endif

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
	return __hiloint2double(hi, lo);
}

/* workgroup barrier for hand-offs that go through LDS only: waits for this wave's LDS traffic, not for
 * its outstanding global stores (Z bytes, qq rows) -- nobody in the workgroup reads those back */
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* same running sums from frequency rows already in registers */
template <int KMAX>
__device__ __forceinline__ double weights_reg(const double (&F)[KMAX], const double (&q)[KMAX], double (&cum)[KMAX], int K)
{
	double run = 0;
#pragma unroll
	for (int m = 0; m < KMAX; m++) {
		if (m < K) {
			double w = q[m] * F[m];
			run = (m == 0) ? w : run + w;
		}
		cum[m] = run;
	}
	return run;
}

struct ZqShared {
	isg_wh_tables tab;          /* skip-ahead tables (2.2 KB) */
	unsigned scan[17];
	int hist[2][ISG_KCAP];      /* bucket counts, double buffered by individual parity (chain kernel) */
	double at_val[1024];        /* Dirichlet attempt table: value (< 0: rejected) ... */
	unsigned char at_used[1024];/* ... and uniforms consumed, per (gamma m, even start offset) */
	double gval[ISG_KCAP];      /* accepted gamma values, stream order */
	unsigned long long used_total;
	unsigned amask[2][ISG_KCAP]; /* cooperative kernel: per gamma, bit o = the attempt at even offset o is accepted ... */
	unsigned rmask[2][ISG_KCAP]; /* ... / did NOT consume exactly two uniforms (double buffered by individual parity) */
	int ghist[2][ISG_KCAP];      /* cooperative kernel: the other workgroups' bucket counts */
	int hist3[3][ISG_KCAP], ghist3[3][ISG_KCAP]; /* k_zq_spec: the same, triple buffered (its Dirichlet has no barrier) */
};

/*
 * rdirich(qqnum[i], K, &qq[i], alpha) (mcmc.c:1196-1198, random.c:264-280).  The K gammas consume the
 * stream one after the other and every rejected attempt restarts at the position the previous one
 * stopped at (random.c:233-250); an attempt's outcome depends only on (shape, start position).  So
 * all (gamma m, even offset) attempts are evaluated concurrently into a table and one lane then walks
 * the table in stream order -- same values, same consumption as the sequential loop.  Shapes equal
 * to 1 (odd consumption, random.c:243-244) or walks that leave the table fall back to the plain loop.
 * Returns (for every thread) the number of uniforms consumed.
 */
template <int BLOCK, int KMAX, bool WRITE = true>
__device__ __forceinline__ unsigned dirichlet_block(const DevView &d, ZqShared &sh, int i, isg_wh dstart, double alpha, int par)
{
	const double *dtape = nullptr; /* (the cooperative kernels read their uniforms from the tape: dirichlet_coop) */
	const unsigned long long dstart_off = 0;
	const int K = d.K, t = threadIdx.x;
	int *hist = sh.hist[par];
	int noff = BLOCK / K;
	if (noff > 32) noff = 32;
	if (t < K * noff) {
		const int m = t / noff, o = t - m * noff;
		const double a = (double)hist[m] + alpha;
		isg_cursor c;
		c.used = 0;
		c.tape = dtape ? dtape + 2 * o : nullptr; /* uniforms of this stretch already on the tape */
		if (!dtape) c.s = isg_wh_jump(&sh.tab, dstart, dstart_off + 2ull * (unsigned)o);
		double r = -1;
		if (a < 1) r = isg_rgamma1_try(&c, a);
		else if (a > 1) r = isg_rgamma2_try(&c, a);
		else c.used = 255; /* shape 1: exponential draw, odd consumption -> sequential fallback */
		sh.at_val[t] = r;
		sh.at_used[t] = (unsigned char)(c.used > 255 ? 255 : c.used);
	}
	lds_barrier();
	STAMP(i, 4);
	if (t < KMAX) sh.hist[par ^ 1][t] = 0; /* the other buffer: last read two barriers ago */
	if (t < 64) {
		/* wave 0 walks the table in stream order.  With at most 256 entries every lane first takes its
		 * (up to 4) entries into registers; a step of the walk is then a pair of v_readlane (the walk state
		 * is wave-uniform) instead of a dependent LDS round trip. */
		const int nent = K * noff;
		const bool small = (nent <= 256);
		double v4[4];
		unsigned u4[4];
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int e = t + 64 * r;
			v4[r] = (small && e < nent) ? sh.at_val[e] : -1.0;
			u4[r] = (small && e < nent) ? (unsigned)sh.at_used[e] : 255u;
		}
		unsigned off = 0;
		int m = 0;
		bool ok = true;
		while (m < K) {
			const unsigned o = off >> 1;
			if (o >= (unsigned)noff) { ok = false; break; }
			const int e = m * noff + (int)o;
			double v;
			unsigned u;
			if (small) {
				const int r = __builtin_amdgcn_readfirstlane(e >> 6), ln = __builtin_amdgcn_readfirstlane(e & 63);
				const double vr = (r == 0) ? v4[0] : (r == 1) ? v4[1] : (r == 2) ? v4[2] : v4[3];
				const unsigned ur = (r == 0) ? u4[0] : (r == 1) ? u4[1] : (r == 2) ? u4[2] : u4[3];
				v = readlane_f64(vr, ln);
				u = (unsigned)__builtin_amdgcn_readlane((int)ur, ln);
			} else {
				v = sh.at_val[e];
				u = sh.at_used[e];
			}
			if (u == 255) { ok = false; break; }
			off += u;
			if (!(v < 0)) {
				if (t == 0) sh.gval[m] = v;
				m++;
			}
		}
		if (t == 0) {
			if (!ok) { /* continue sequentially from (gamma m, offset off) */
				isg_cursor c;
				c.s = isg_wh_jump(&sh.tab, dstart, dstart_off + off);
				c.used = 0;
				c.tape = nullptr;
				for (int mm = m; mm < K; mm++) sh.gval[mm] = isg_rgamma(&c, (double)hist[mm] + alpha);
				off += c.used;
			}
			sh.used_total = off;
		}
	}
	lds_barrier();
	STAMP(i, 6);
	const unsigned used = (unsigned)sh.used_total;
	/* the next individual only needs `used`; qq[i] = g / sum is finished off the critical path
	 * by lanes of the last wave: hist is double buffered and gval is only rewritten three barriers later */
	if (WRITE && t >= BLOCK - 64 && t - (BLOCK - 64) < K) {
		const int mm = t - (BLOCK - 64);
		double sum = 0;
		for (int k2 = 0; k2 < K; k2++) sum += sh.gval[k2];
		d.qq[(size_t)i * K + mm] = sh.gval[mm] / sum;
		d.qqnum[(size_t)i * K + mm] = hist[mm];
	}
	return used;
}

/* per-individual inputs that do not depend on the stream position: fetched one individual ahead */
template <int KMAX>
struct ZqPrefetch {
	unsigned long long gb; /* this lane's first 8 genotype bytes */
	int nvalid;
	double q[KMAX];
};
template <int KMAX>
__device__ __forceinline__ void zq_prefetch(const DevView &d, int i, int init_flag, ZqPrefetch<KMAX> &pf)
{
	const int t = threadIdx.x;
	pf.gb = ~0ull;
	if (t * ISG_LPT < d.Lp) {
		const uint2 g = *(const uint2 *)(d.geno + (size_t)i * d.Lp * 2 + (size_t)t * ISG_LPT * 2);
		pf.gb = ((unsigned long long)g.y << 32) | g.x;
	}
	pf.nvalid = d.nvalid[i];
#pragma unroll
	for (int m = 0; m < KMAX; m++) pf.q[m] = (m < d.K && !init_flag) ? d.qq[(size_t)i * d.K + m] : 0.0;
}

/*
 * Single precision pre-filter of one Z draw.  The bucket is the number of m <= K-2 with
 * fl(cum[m]/tot) < x (see bucket_fast).  Evaluated in float the sign of x*tot - cum[m] is right
 * whenever it clears a guard band of 6e-6*tot (the float rounding of qq, freq, the K-term sums and of
 * the float-evaluated uniform stays below 2e-6*tot); otherwise, or when x is within 4e-6 of 0 or 1,
 * `amb` is raised and the caller redoes THIS draw in double (bucket_fast), so the returned Z is always
 * the reference's.  About 5 draws in 100000 are redone.
 */
/* bucket_f32 split in two: the running sums of a copy (position independent) and the comparison of one uniform */
template <int KMAX>
__device__ __forceinline__ float prefix_f32(const float (&F)[KMAX], const float (&qf)[KMAX], int K, float (&cum)[KMAX])
{
	float run = 0.f;
#pragma unroll
	for (int m = 0; m < KMAX; m++) {
		if (m < K) run = (m == 0) ? qf[m] * F[m] : run + qf[m] * F[m];
		cum[m] = run;
	}
	return run;
}
/* same decision as bucket_f32: the guard band test |dd| <= marg for any m is  !(min |dd| > marg) */
template <int KMAX>
__device__ __forceinline__ int bucket_from_cum(float xf, const float (&cum)[KMAX], float run, int K, bool *amb)
{
	const float p = xf * run, marg = 6e-6f * run;
	float mn = 3.0e38f;
	int z = 0;
#pragma unroll
	for (int m = 0; m < KMAX - 1; m++) {
		if (m < K - 1) {
			const float dd = p - cum[m];
			z += (dd > 0.f) ? 1 : 0;
			mn = __builtin_fminf(mn, __builtin_fabsf(dd));
		}
	}
	*amb = !(mn > marg) || !(run > 1e-30f && run < 1e30f) || !(xf > 4e-6f && xf < 1.0f - 4e-6f);
	return z;
}

template <int KMAX>
__device__ __forceinline__ int bucket_f32(float xf, const float (&F)[KMAX], const float (&qf)[KMAX], int K, bool *amb)
{
	float cum[KMAX];
	const float run = prefix_f32<KMAX>(F, qf, K, cum);
	if (KMAX > 8) return bucket_from_cum<KMAX>(xf, cum, run, K, amb);
	/* few clusters: the per-threshold band tests schedule better than the running minimum (measured: 0.50 vs
	 * 0.58 ms for the keyed kernel at K = 5; the other way round at K = 10) */
	const float p = xf * run, marg = 6e-6f * run;
	bool a = !(run > 1e-30f && run < 1e30f) || !(xf > 4e-6f && xf < 1.0f - 4e-6f);
	int z = 0;
#pragma unroll
	for (int m = 0; m < KMAX - 1; m++) {
		if (m < K - 1) {
			const float dd = p - cum[m];
			z += (dd > 0.f) ? 1 : 0;
			a |= !(dd > marg || dd < -marg);
		}
	}
	*amb = a;
	return z;
}

/* Z draws of one individual; `cur` = stream state at its first position, `off` = that position counted
 * from the start of the phase (index into the uniform tape).  Returns uniforms consumed. */
/* WRITE = false: only the consumption is wanted (Z, qq, qqnum stay as they are) */
template <int BLOCK, int KMAX, bool CHAIN, bool WRITE = true>
__device__ __forceinline__ unsigned zq_one(const DevView &d, ZqShared &sh, int i, isg_wh cur, unsigned long long off, int init_flag,
					   double alpha, isg_wh mult0, isg_wh mult_pass, ZqPrefetch<KMAX> &pf)
{
	constexpr bool HOIST = (KMAX <= 8); /* float frequency rows of a pass in registers before they are used */
	const int K = d.K, t = threadIdx.x;
	STAMP(i, 0);
	const int nvalid = pf.nvalid;
	const bool fast = (nvalid == d.L);
	/* the tape covers this individual's Z draws? (it always does unless the Dirichlets consumed far
	 * more than budgeted; then the uniforms are generated on the fly) */
	const bool taped = CHAIN && d.tape != nullptr && off + 2ull * (unsigned)nvalid <= d.tape_len;
	double q[KMAX];
	float qf[KMAX];
#pragma unroll
	for (int m = 0; m < KMAX; m++) {
		q[m] = pf.q[m];
		qf[m] = (float)pf.q[m];
	}
	unsigned long long gb = pf.gb;
	if (CHAIN && i + 1 < d.N) zq_prefetch<KMAX>(d, i + 1, init_flag, pf); /* lands while this individual is processed */
	double icum[KMAX];
	if (init_flag) {
#pragma unroll
		for (int m = 0; m < KMAX; m++) icum[m] = (m < K) ? (double)(m + 1) / K : 0.0; /* mcmc.c:1144 */
	}
	/* per-wave bucket counts: ballots + scalar popcounts, no cross-lane data movement */
	int wcnt[KMAX];
#pragma unroll
	for (int m = 0; m < KMAX; m++) wcnt[m] = 0;
	unsigned long long pc0 = 0, pc1 = 0;
	unsigned npass = 0;
	const int par = CHAIN ? (i & 1) : 0;
	const size_t rowb = (size_t)d.Lp * 2;
	const uint8_t *grow = d.geno + (size_t)i * rowb;
	uint8_t *zrow = d.z + (size_t)i * rowb;
	unsigned running = 0;
	isg_wh mult = mult0;
	for (int jb = 0; jb < d.Lp; jb += BLOCK * ISG_LPT) {
		const int j0 = jb + t * ISG_LPT;
		unsigned long long gnext = ~0ull;
		if (j0 + BLOCK * ISG_LPT < d.Lp) {
			const uint2 g = *(const uint2 *)(grow + (size_t)(j0 + BLOCK * ISG_LPT) * 2);
			gnext = ((unsigned long long)g.y << 32) | g.x;
		}
		/* phase 1: single precision frequency rows of the (up to) 8 allele copies */
		float R0[HOIST ? ISG_LPT : 1][KMAX], R1[HOIST ? ISG_LPT : 1][KMAX];
		if (HOIST && !init_flag) {
#pragma unroll
			for (int l = 0; l < ISG_LPT; l++) {
				unsigned a0 = (unsigned)(gb >> (16 * l)) & 0xff, a1 = (unsigned)(gb >> (16 * l + 8)) & 0xff;
				const int j = (j0 + l < d.L) ? j0 + l : 0;
				if (a0 == 0xff) { a0 = 0; a1 = 0; }
				const float *F0 = d.freqf + ((size_t)j * d.Amax + a0) * d.KPF, *F1 = d.freqf + ((size_t)j * d.Amax + a1) * d.KPF;
#pragma unroll
				for (int m = 0; m < KMAX; m += 4) {
					if (m < K) {
						const float4 f0 = *(const float4 *)(F0 + m), f1 = *(const float4 *)(F1 + m);
						R0[l][m] = f0.x; R1[l][m] = f1.x;
						if (m + 1 < KMAX) { R0[l][m + 1] = f0.y; R1[l][m + 1] = f1.y; }
						if (m + 2 < KMAX) { R0[l][m + 2] = f0.z; R1[l][m + 2] = f1.z; }
						if (m + 3 < KMAX) { R0[l][m + 3] = f0.w; R1[l][m + 3] = f1.w; }
					}
				}
			}
		}
		/* phase 2: the lane's uniforms (one per allele copy of its used loci, in stream order) */
		unsigned nv = 0;
#pragma unroll
		for (int l = 0; l < ISG_LPT; l++) nv += (((unsigned)(gb >> (16 * l)) & 0xff) != 0xff) ? 1u : 0u;
		unsigned rank; /* used loci of this individual before the lane's first locus */
		if (fast) {
			rank = (unsigned)j0;
		} else {
			unsigned tot;
			rank = running + block_excl_scan<BLOCK>(nv, sh.scan, &tot);
			running += tot;
		}
		/* `x` holds the exact uniforms when they come from the tape; when they are generated here only
		 * the packed generator states are kept: the float pre-filter needs a float value, the exact
		 * double is formed just for the few draws that are redone */
		double x[2 * ISG_LPT];
		float xf[2 * ISG_LPT];
		unsigned long long st[2 * ISG_LPT];
		if (taped) {
			const double *tp = d.tape + off + 2ull * rank;
			unsigned k = 0;
#pragma unroll
			for (int l = 0; l < ISG_LPT; l++) {
				const bool valid = (((unsigned)(gb >> (16 * l)) & 0xff) != 0xff);
				x[2 * l] = valid ? tp[k] : 0.0;
				x[2 * l + 1] = valid ? tp[k + 1] : 0.0;
				xf[2 * l] = (float)x[2 * l];
				xf[2 * l + 1] = (float)x[2 * l + 1];
				k += valid ? 2u : 0u;
			}
		} else {
			isg_wh s;
			if (fast) {
				s = isg_wh_mul(cur, mult);
				mult = isg_wh_mul(mult, mult_pass);
			} else {
				s = isg_wh_jump32(&sh.tab, cur, 2u * rank);
			}
#pragma unroll
			for (int l = 0; l < ISG_LPT; l++) {
				const bool valid = (((unsigned)(gb >> (16 * l)) & 0xff) != 0xff);
				isg_wh s2 = s;
#pragma unroll
				for (int cp = 0; cp < 2; cp++) {
					isg_wh_step(&s2);
					st[2 * l + cp] = (unsigned long long)s2.s1 | ((unsigned long long)s2.s2 << 16) | ((unsigned long long)s2.s3 << 32);
					xf[2 * l + cp] = isg_wh_value_f32(&s2);
				}
				s.s1 = valid ? s2.s1 : s.s1; /* unused loci consume nothing (mcmc.c:1137) */
				s.s2 = valid ? s2.s2 : s.s2;
				s.s3 = valid ? s2.s3 : s.s3;
			}
		}
		auto exact_x = [&](int c8) -> double {
			if (taped) return x[c8];
			isg_wh e;
			e.s1 = (uint32_t)(st[c8] & 0xffff);
			e.s2 = (uint32_t)((st[c8] >> 16) & 0xffff);
			e.s3 = (uint32_t)((st[c8] >> 32) & 0xffff);
			return isg_wh_value(&e);
		};
		/* phase 3: buckets.  Straight-line float pre-filter over the 8 copies; flagged draws are redone
		 * in double afterwards (rare), so no data-dependent branch sits between the dependency chains. */
		int zz[2 * ISG_LPT];
		unsigned redo = 0;
		if (init_flag) {
#pragma unroll
			for (int c8 = 0; c8 < 2 * ISG_LPT; c8++) zz[c8] = bucket_fast<KMAX>(exact_x(c8), icum, 1.0, K); /* vec[K-1] = K/K */
		} else if (HOIST) {
#pragma unroll
			for (int l = 0; l < ISG_LPT; l++) {
				bool amb;
				zz[2 * l] = bucket_f32<KMAX>(xf[2 * l], R0[l], qf, K, &amb);
				redo |= amb ? (1u << (2 * l)) : 0u;
				zz[2 * l + 1] = bucket_f32<KMAX>(xf[2 * l + 1], R1[l], qf, K, &amb);
				redo |= amb ? (2u << (2 * l)) : 0u;
			}
		} else {
			redo = 0xffu;
		}
		if (redo) {
#pragma unroll
			for (int l = 0; l < ISG_LPT; l++) {
				const unsigned a0 = (unsigned)(gb >> (16 * l)) & 0xff, a1 = (unsigned)(gb >> (16 * l + 8)) & 0xff;
				if (a0 == 0xff) continue;
				double cum[KMAX], tot;
				if (redo & (1u << (2 * l))) {
					tot = weights<KMAX>(d.freq + ((size_t)(j0 + l) * d.Amax + a0) * d.KP, q, cum, K);
					zz[2 * l] = bucket_fast<KMAX>(exact_x(2 * l), cum, tot, K);
				}
				if (redo & (2u << (2 * l))) {
					tot = weights<KMAX>(d.freq + ((size_t)(j0 + l) * d.Amax + a1) * d.KP, q, cum, K);
					zz[2 * l + 1] = bucket_fast<KMAX>(exact_x(2 * l + 1), cum, tot, K);
				}
			}
		}
		unsigned long long zb = ~0ull;
#pragma unroll
		for (int l = 0; l < ISG_LPT; l++) {
			const bool valid = (((unsigned)(gb >> (16 * l)) & 0xff) != 0xff);
			const int z0 = valid ? zz[2 * l] : 0xff, z1 = valid ? zz[2 * l + 1] : 0xff; /* 0xff: locus unused */
			if (KMAX <= 8) { /* per-lane counters, 16 bits per cluster (a ballot + scalar popcount per cluster and copy stalls the wave ~80 cycles each) */
				const unsigned long long i0 = valid ? (1ull << (16 * (z0 & 3))) : 0ull, i1 = valid ? (1ull << (16 * (z1 & 3))) : 0ull;
				pc0 += ((z0 & 4) ? 0ull : i0) + ((z1 & 4) ? 0ull : i1);
				if (KMAX > 4) pc1 += ((z0 & 4) ? i0 : 0ull) + ((z1 & 4) ? i1 : 0ull);
			} else {
#pragma unroll
				for (int m = 0; m < KMAX; m++)
					if (m < K) wcnt[m] += __popcll(__ballot(z0 == m)) + __popcll(__ballot(z1 == m));
			}
			zb = (zb & ~(0xffffull << (16 * l))) | ((unsigned long long)(z0 | (z1 << 8)) << (16 * l));
		}
		if (KMAX <= 8 && (++npass & 1023) == 0) { /* 8 copies per pass: the 16-bit fields are emptied long before they can wrap */
#pragma unroll
			for (int m = 0; m < KMAX; m++)
				if (m < K) wcnt[m] += (int)wave_sum_u32((unsigned)(((m < 4 ? pc0 : pc1) >> (16 * (m & 3))) & 0xffffu));
			pc0 = pc1 = 0;
		}
		if (WRITE && j0 < d.Lp) {
			uint2 zo;
			zo.x = (unsigned)zb;
			zo.y = (unsigned)(zb >> 32);
			*(uint2 *)(zrow + (size_t)j0 * 2) = zo;
		}
		gb = gnext;
	}
	STAMP(i, 1);
	if (KMAX <= 8) {
#pragma unroll
		for (int m = 0; m < KMAX; m++)
			if (m < K) wcnt[m] += (int)wave_sum_u32((unsigned)(((m < 4 ? pc0 : pc1) >> (16 * (m & 3))) & 0xffffu));
	}
	/* qqnum[i][m] (mcmc.c:1176-1194): one LDS add per wave and cluster into the (pre-zeroed) buffer */
	STAMP(i, 2);
	if (lane_id() == 0) {
#pragma unroll
		for (int m = 0; m < KMAX; m++)
			if (m < K && wcnt[m]) atomicAdd(&sh.hist[par][m], wcnt[m]);
	}
	lds_barrier();
	STAMP(i, 3);
	const isg_wh dstart = isg_wh_jump32(&sh.tab, cur, 2u * (unsigned)nvalid);
	return 2u * (unsigned)nvalid + dirichlet_block<BLOCK, KMAX, WRITE>(d, sh, i, dstart, alpha, par);
}

template <int BLOCK, int KMAX, bool CHAIN>
__global__ void __launch_bounds__(BLOCK) k_zq(DevView d, isg_wh base, uint64_t pos0, uint64_t stride, int init_flag, double alpha,
					      uint64_t *pos_out)
{
	__shared__ ZqShared sh;
	const int t = threadIdx.x;
	{ /* skip-ahead tables -> LDS */
		const uint16_t *src = (const uint16_t *)d.tab;
		uint16_t *dst = (uint16_t *)&sh.tab;
		for (int k = t; k < (int)(sizeof(isg_wh_tables) / 2); k += BLOCK) dst[k] = src[k];
		if (t < 2 * ISG_KCAP) (&sh.hist[0][0])[t] = 0;
	}
	__syncthreads();
	/* multipliers for lane t: its first copy sits 2*4*t uniforms into a pass, a pass spans 8*BLOCK */
	const isg_wh mult0 = isg_wh_power(&sh.tab, 8u * (unsigned)t);
	const isg_wh mult_pass = isg_wh_power(&sh.tab, 8u * (unsigned)BLOCK);
	ZqPrefetch<KMAX> pf;
	if (CHAIN) {
		isg_wh cur = isg_wh_jump(&sh.tab, base, pos0);
		uint64_t total = 0;
		zq_prefetch<KMAX>(d, 0, init_flag, pf);
		for (int i = 0; i < d.N; i++) {
			unsigned used = zq_one<BLOCK, KMAX, true>(d, sh, i, cur, total, init_flag, alpha, mult0, mult_pass, pf);
			cur = isg_wh_jump32(&sh.tab, cur, used);
			total += used;
		}
		if (t == 0) *pos_out = total;
	} else {
		const int i = blockIdx.x;
		isg_wh cur = isg_wh_jump(&sh.tab, base, pos0 + (uint64_t)i * stride);
		zq_prefetch<KMAX>(d, i, init_flag, pf);
		zq_one<BLOCK, KMAX, false>(d, sh, i, cur, 0, init_flag, alpha, mult0, mult_pass, pf);
	}
}

/* ------------------------------------------------------------------------------------------ */
/* k_zq_coop: update_ZQ, replay schedule, several workgroups per individual                      */
/* ------------------------------------------------------------------------------------------ */
/*
 * In the replay schedule individual i+1 starts where individual i's Dirichlet stopped, so the
 * individuals are processed in order; what CAN be spread out is one individual's loci.  G workgroups
 * (one locus per lane) all work on the same individual: each draws the Z of its loci from the uniform
 * tape at the individual's start offset, counts its buckets and publishes the counts; every workgroup
 * collects all of them and draws the Dirichlet itself (same inputs, same consumption), so each knows where
 * the next individual starts without being told: one exchange per individual.
 *
 * Hand-offs are single naturally aligned 8-byte words that carry their own tag (16 bits derived from
 * the individual's index), written with one agent-scope store and polled with agent-scope loads
 * (MI355X_MICROARCH.md, "data-tagged granules"): no separate flag, no fence ordering to get wrong.
 *   gran[slot][g][w] = counts 3w..3w+2 (16 bits each) | tag(16)  workgroup (or wave) g -> everybody
 * Nobody runs more than one individual ahead of anybody else (the next offset needs everybody's
 * counts), so a ring of 4 slots suffices.  Every spin is bounded and watches a common abort word.
 */
#define ISG_COOP_RING 4
#define ISG_COOP_GMAX 128
#define ISG_COOP_WMAX 11
struct CoopBuf {
	unsigned long long gran[ISG_COOP_RING][ISG_COOP_GMAX * ISG_COOP_WMAX];
	unsigned abort_flag;
	unsigned overflow_flag;
	unsigned long long xcc[ISG_COOP_GMAX]; /* which XCD each workgroup runs on (tagged), for the same-XCD fast path */
};
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v)
{
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* store that stays in the XCD's L2 (no write-through to the memory side): visible to agent-scope loads of
 * workgroups on the SAME XCD only -- used after the workgroups have established that they share one */
__device__ __forceinline__ void st_xcd(unsigned long long *p, unsigned long long v)
{
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned xcc_id()
{
	unsigned x;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
	return x & 0xfu;
}
/*
 * Do all G workgroups of this launch run on one XCD?  (Blocks are dealt round-robin over the 8 XCDs, so the
 * launcher starts 8 G blocks of which every 8th works; this only makes it likely -- the answer comes from the
 * hardware register.)  One tagged agent-scope word per workgroup, everybody reads all of them.
 */
__device__ __forceinline__ bool coop_same_xcd(CoopBuf *cb, int g, int G, unsigned *lds_flag);
/*
 * isg_rgamma2_try (random.c:195-231) for the cooperative kernels' critical path.  The two logarithms only feed the
 * accept / reject decision  c3 log(u1) - log(w) + w >= 1 ; evaluated with single precision logarithms the sign of
 * (that - 1) is certain outside a band of 2e-6 (1 + |terms|) -- v_log_f32 is good to 1 ulp, the float image of the
 * argument to 6e-8 -- and inside the band (or for non-finite intermediates) the double precision expression
 * decides.  Same return value, same consumption as isg_rgamma2_try in every case.
 */
__device__ __forceinline__ double rgamma1_try_pre(double u0, double u1, double alpha) /* isg_rgamma1_try, uniforms drawn */
{
	double r, x;
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
/* pu0, pu1: the first two uniforms, already drawn from *c */
__device__ __forceinline__ double rgamma2_try_dev(isg_cursor *c, double pu0, double pu1, double alpha)
{
	double u1, u2, c1, c2, c3, c4, c5, w;
	c1 = alpha - 1;
	c2 = (alpha - 1 / (6 * alpha)) / c1;
	c3 = 2 / c1;
	c4 = c3 + 2;
	c5 = 1 / isg_sqrt(alpha);
	u1 = pu0;
	u2 = pu1;
	if (alpha > 2.5) u1 = u2 + c5 * (1 - 1.86 * u1);
	while ((u1 >= 1) || (u1 <= 0)) {
		u1 = isg_cur_next(c);
		u2 = isg_cur_next(c);
		if (alpha > 2.5) u1 = u2 + c5 * (1 - 1.86 * u1);
	}
	w = c2 * u2 / u1;
	if ((c3 * u1 + w + 1 / w) > c4) {
		const float l1 = __builtin_amdgcn_logf((float)u1) * 0.693147180559945f, lw = __builtin_amdgcn_logf((float)w) * 0.693147180559945f;
		const double al1 = __builtin_fabs((double)l1), alw = __builtin_fabs((double)lw);
		const double dlt = (c3 * (double)l1 - (double)lw + w) - 1;
		const double tol = 2e-6 * (__builtin_fabs(c3) * (1.0 + al1) + 1.0 + alw) + 1e-12 * __builtin_fabs(w);
		bool rej;
		if (dlt > tol) rej = true;
		else if (dlt < -tol) rej = false;
		else rej = (c3 * isg_log(u1) - isg_log(w) + w) >= 1;
		if (rej) return -1;
	}
	return c1 * w;
}

/* the calling lane polls granule *p until it carries `tag`; every spin is bounded and watches the common abort word */
__device__ __forceinline__ unsigned long long coop_poll(const unsigned long long *p, unsigned tag, CoopBuf *cb)
{
	unsigned long long v = 0;
	for (unsigned spin = 0;; spin++) {
		v = ld_agent(p);
		if ((unsigned)(v >> 48) == tag) break;
		if ((spin & 1023u) == 1023u) {
			if (__hip_atomic_load(&cb->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
			if (spin > (1u << 24)) {
				__hip_atomic_store(&cb->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				break;
			}
		}
		__builtin_amdgcn_s_sleep(1);
	}
	return v;
}

__device__ __forceinline__ bool coop_same_xcd(CoopBuf *cb, int g, int G, unsigned *lds_flag)
{
	const int t = threadIdx.x;
	if (t == 0) {
		*lds_flag = 1u;
		st_agent(&cb->xcc[g], (1ull << 48) | xcc_id());
	}
	__syncthreads();
	if (t < G) {
		const unsigned long long v = coop_poll(&cb->xcc[t], 1u, cb);
		if ((unsigned)(v >> 48) != 1u || (unsigned)(v & 0xfu) != xcc_id()) atomicAnd(lds_flag, 0u);
	}
	__syncthreads();
	return *lds_flag != 0u;
}

/*
 * The Dirichlet of the cooperative kernels.  Every workgroup runs it on the same counts and gets the same
 * consumption, so nobody has to be told where the next individual starts: ONE exchange (the counts) per
 * individual.  Attempts are evaluated as in dirichlet_block; the walk over the attempt table is scalar: per gamma
 * a 32-bit mask of accepted attempts and one of attempts that did not consume exactly two uniforms (a retry inside
 * the attempt, random.c:213-216), so a step is shift / find-first-set / add on SGPRs.  Shape 1 (odd consumption)
 * or a walk that leaves the table continue with the plain sequential loop from that point.
 * Called by all threads after the counts are complete in sh.hist[par]; returns the uniforms consumed.
 * `writer`: this wave stores qq[i] / qqnum[i].
 */
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int BLOCK, int KMAX, class Hook = NoHook>
__device__ __forceinline__ unsigned dirichlet_coop(const DevView &d, ZqShared &sh, int i, const isg_wh &cur, unsigned long long dstart_off,
						  double alpha, int par, const double *dtape, bool writer, Hook hook = Hook())
{
	const int K = d.K, t = threadIdx.x, lane = (int)lane_id();
	const int *hist = sh.hist[par], *ghist = sh.ghist[par]; /* own + collected counts */
	int noff = BLOCK / K;
	if (noff > 32) noff = 32;
	if (t < K * noff) {
		const int m = t / noff, o = t - m * noff;
		const double a = (double)(hist[m] + ghist[m]) + alpha;
		isg_cursor c;
		c.used = 0;
		c.tape = dtape ? dtape + 2 * o : nullptr;
		if (!dtape) c.s = isg_wh_jump(&sh.tab, cur, dstart_off + 2ull * (unsigned)o);
		/* the attempt's first two uniforms, then whatever the caller wants in flight behind them (loads return in
		 * issue order: nothing issued from here on delays these two) */
		const double pu0 = isg_cur_next(&c), pu1 = isg_cur_next(&c);
		hook();
		double r = -1;
		if (a < 1) r = rgamma1_try_pre(pu0, pu1, a);
		else if (a > 1) r = rgamma2_try_dev(&c, pu0, pu1, a);
		else c.used = 255;
		sh.at_val[t] = r;
		sh.at_used[t] = (unsigned char)(c.used > 255 ? 255 : c.used);
		if (!(r < 0)) atomicOr(&sh.amask[par][m], 1u << o);
		if (c.used != 2) atomicOr(&sh.rmask[par][m], 1u << o);
	}
	else hook();
	if (t < KMAX) { /* the other parity's buffers: last read before the previous two barriers */
		sh.hist[par ^ 1][t] = 0;
		sh.ghist[par ^ 1][t] = 0;
		sh.amask[par ^ 1][t] = 0;
		sh.rmask[par ^ 1][t] = 0;
	}
	lds_barrier();
	STAMP(i, 4);
	/* every wave walks for itself (wave-uniform state) */
	const unsigned am = (lane < K) ? sh.amask[par][lane] : 0u, rm = (lane < K) ? sh.rmask[par][lane] : 0u;
	unsigned o = 0, my_o = 0;
	int m = 0;
	{ /* the common case straight-line: per gamma the first attempt at or after o that is accepted, none irregular */
		unsigned fo = 0, fmy = 0;
		bool bad = false;
#pragma unroll
		for (int mm = 0; mm < KMAX; mm++) {
			if (mm < K) {
				const unsigned A = (unsigned)__builtin_amdgcn_readlane((int)am, mm), I = (unsigned)__builtin_amdgcn_readlane((int)rm, mm);
				const unsigned rest = (fo < 32u) ? ((A | I) >> fo) : 0u;
				const unsigned e = fo + (rest ? (unsigned)__builtin_ctz(rest) : 0u);
				bad |= (rest == 0u) || (((I >> (e & 31u)) & 1u) != 0u) || (e >= (unsigned)noff);
				fmy = (lane == mm) ? e : fmy;
				fo = e + 1u;
			}
		}
		if (!bad) {
			o = fo;
			my_o = fmy;
			m = K;
		}
	}
	while (m < K) {
		const unsigned A = (unsigned)__builtin_amdgcn_readlane((int)am, m), I = (unsigned)__builtin_amdgcn_readlane((int)rm, m);
		if (o >= (unsigned)noff) break;
		const unsigned rest = (A | I) >> o; /* the attempts before the first set bit are plain rejections: 2 uniforms each */
		if (rest == 0) break;
		const unsigned e = o + (unsigned)__builtin_ctz(rest);
		unsigned step = 1;
		if ((I >> e) & 1u) { /* a retry inside the attempt (random.c:213-216) or shape 1: consumption from the table */
			const unsigned u = sh.at_used[m * noff + (int)e];
			if (u == 255u || (u & 1u)) break;
			step = u >> 1;
		}
		o = e + step;
		if ((A >> e) & 1u) {
			my_o = (lane == m) ? e : my_o;
			m++;
		}
	}
	unsigned used = 2u * o;
	const bool ok = (m == K);
	double v = (lane < m) ? sh.at_val[lane * noff + (int)my_o] : 0.0;
	if (!ok) { /* wave-uniform and the same in every wave: continue sequentially from (gamma m, offset 2 o) */
		if (t < 64) {
			if (lane < m) sh.gval[lane] = v;
			if (t == 0) {
				isg_cursor c;
				c.s = isg_wh_jump(&sh.tab, cur, dstart_off + 2ull * o);
				c.used = 0;
				c.tape = nullptr;
				for (int mm = m; mm < K; mm++) sh.gval[mm] = isg_rgamma(&c, (double)(hist[mm] + ghist[mm]) + alpha);
				sh.used_total = 2ull * o + c.used;
			}
		}
		lds_barrier();
		used = (unsigned)sh.used_total;
		if (lane < K) v = sh.gval[lane];
		lds_barrier(); /* gval / used_total are rewritten by the next fallback only after this */
	}
	STAMP(i, 5);
	if (writer) { /* qq[i] = g / sum with the sum taken in stream order (random.c:272-279) */
		double sum = 0;
		for (int k2 = 0; k2 < K; k2++) sum += readlane_f64(v, k2);
		if (lane < K) {
			d.qq[(size_t)i * K + lane] = v / sum;
			d.qqnum[(size_t)i * K + lane] = hist[lane] + ghist[lane];
		}
	}
	return used;
}

template <int KMAX>
__global__ void __launch_bounds__(256) k_zq_coop(DevView d, isg_wh base, int init_flag, double alpha, CoopBuf *cb, uint64_t *pos_out, int xcd_pack)
{
	constexpr int BLOCK = 256;
	constexpr bool PRE = (KMAX <= 8);
	__shared__ ZqShared sh;
	__shared__ unsigned same_xcd;
	if (xcd_pack && (blockIdx.x & 7)) return; /* every 8th block works: one XCD under round-robin placement */
	const int t = threadIdx.x, g = xcd_pack ? blockIdx.x >> 3 : blockIdx.x, G = xcd_pack ? gridDim.x >> 3 : gridDim.x, K = d.K;
	{
		const uint16_t *src = (const uint16_t *)d.tab;
		uint16_t *dst = (uint16_t *)&sh.tab;
		for (int k = t; k < (int)(sizeof(isg_wh_tables) / 2); k += BLOCK) dst[k] = src[k];
		if (t < 2 * ISG_KCAP) {
			(&sh.hist[0][0])[t] = 0;
			(&sh.amask[0][0])[t] = 0;
			(&sh.rmask[0][0])[t] = 0;
			(&sh.ghist[0][0])[t] = 0;
		}
	}
	__syncthreads();
	const bool local = xcd_pack && coop_same_xcd(cb, g, G, &same_xcd);
	const isg_wh cur = isg_wh_jump(&sh.tab, base, 0);
	unsigned long long off = 0; /* offset of the current individual: every workgroup derives it for itself */
	const int stride = G * BLOCK;
	const size_t rowb = (size_t)d.Lp * 2;
	const bool writer = (g == 0) && (t >= BLOCK - 64);
	const bool wmode = (G * (BLOCK / 64) * ((K + 2) / 3) <= BLOCK); /* one polling round covers a granule per wave */
	/* counts per exchanged word: 3 x 16 bits, or 4 x 12 bits when a workgroup's count of one bucket stays below 4096 */
	const int npass = (d.Lp + G * BLOCK - 1) / (G * BLOCK);
	const int pack = (!wmode && 2 * BLOCK * npass < 4096) ? 4 : 3, bits = (pack == 4) ? 12 : 16, W = (K + pack - 1) / pack;
	double icum[KMAX];
#pragma unroll
	for (int m = 0; m < KMAX; m++) icum[m] = (m < K) ? (double)(m + 1) / K : 0.0; /* mcmc.c:1144 */
	double touch = 0.0, touch2 = 0.0;
	/* position independent data of the lane's upcoming locus, loaded one step ahead */
	unsigned pa0 = 0xff, pa1 = 0xff, prw = 0;
	float pF0[KMAX], pF1[KMAX];
#pragma unroll
	for (int m = 0; m < KMAX; m++) pF0[m] = pF1[m] = 0.f;
	unsigned na0 = 0xff, na1 = 0xff, nrw = 0; /* the NEXT individual's first locus (one step further ahead) */
	auto fetch_geno = [&](int ni, int nj, unsigned &b0, unsigned &b1, unsigned &rw) {
		b0 = b1 = 0xff;
		rw = 0;
		if (ni < d.N && nj < d.Lp) {
			const unsigned short gg = *(const unsigned short *)(d.geno + (size_t)ni * rowb + (size_t)nj * 2);
			b0 = gg & 0xff;
			b1 = gg >> 8;
			rw = d.rankwave[(size_t)ni * d.nwv + (nj >> 6)];
		}
	};
	auto fetch_rows = [&](int nj) {
		if (pa0 != 0xff && PRE && !init_flag) {
			const float *P0 = d.freqf + ((size_t)nj * d.Amax + pa0) * d.KPF, *P1 = d.freqf + ((size_t)nj * d.Amax + pa1) * d.KPF;
#pragma unroll
			for (int m = 0; m < KMAX; m += 4) {
				if (m < K) {
					const float4 f0 = *(const float4 *)(P0 + m), f1 = *(const float4 *)(P1 + m);
					pF0[m] = f0.x; pF1[m] = f1.x;
					if (m + 1 < KMAX) { pF0[m + 1] = f0.y; pF1[m + 1] = f1.y; }
					if (m + 2 < KMAX) { pF0[m + 2] = f0.z; pF1[m + 2] = f1.z; }
					if (m + 3 < KMAX) { pF0[m + 3] = f0.w; pF1[m + 3] = f1.w; }
				}
			}
		}
	};
	fetch_geno(0, g * BLOCK + t, pa0, pa1, prw);
	fetch_rows(g * BLOCK + t);
	int pnvalid = d.nvalid[0];
	double pq[KMAX];
#pragma unroll
	for (int m = 0; m < KMAX; m++) pq[m] = (m < K && !init_flag) ? d.qq[m] : 0.0;
	for (int i = 0; i < d.N; i++) {
		const unsigned tag = (unsigned)(i % 65535) + 1u;
		const int slot = i & (ISG_COOP_RING - 1), par = i & 1;
		if (touch2 == -1.0 || touch == -1.0) cb->overflow_flag = 2; /* consumes last individual's cache warming loads */
		const int nvalid = pnvalid;
		double q[KMAX];
		float qf[KMAX];
#pragma unroll
		for (int m = 0; m < KMAX; m++) {
			q[m] = pq[m];
			qf[m] = (float)q[m];
		}
		int wcnt[KMAX];
#pragma unroll
		for (int m = 0; m < KMAX; m++) wcnt[m] = 0;
		STAMP(i, 0);
		const unsigned long long offi = off;
		const bool covered = offi + 2ull * (unsigned)nvalid + 1024ull <= d.tape_len;
		/* Loads return in issue order, so everything that is only needed later is issued BEHIND the loads the
		 * critical path waits for: the uniforms first, then the warm-up of the Dirichlet's stretch and the next
		 * individual's data. */
		for (int j = g * BLOCK + t; j - t < d.Lp; j += stride) { /* wave-uniform trip count */
			const unsigned a0 = pa0, a1 = pa1;
			const bool valid = (a0 != 0xff);
			const unsigned long long vm = __ballot(valid);
			const unsigned rank = prw + (unsigned)__popcll(vm & ((1ull << lane_id()) - 1ull));
			float F0[KMAX], F1[KMAX];
#pragma unroll
			for (int m = 0; m < KMAX; m++) { F0[m] = pF0[m]; F1[m] = pF1[m]; }
			double x0 = 0, x1 = 0;
			if (valid && covered) {
				x0 = d.tape[offi + 2ull * rank];
				x1 = d.tape[offi + 2ull * rank + 1];
			}
#ifdef ISG_EXP_XWAIT
			STAMP(i, 2);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			STAMP(i, 3);
#endif
			if (j - t == g * BLOCK) { /* first pass */
				if (covered && t < 10) touch = d.tape[offi + 2ull * (unsigned)nvalid + 16u * (unsigned)t]; /* one lane per 128-byte line */
				fetch_geno(i + 1, g * BLOCK + t, na0, na1, nrw);
				if (i + 1 < d.N) {
					pnvalid = d.nvalid[i + 1];
#pragma unroll
					for (int m = 0; m < KMAX; m++) pq[m] = (m < K && !init_flag) ? d.qq[(size_t)(i + 1) * K + m] : 0.0;
				}
			}
			const bool more = (j - t + stride < d.Lp);
			if (more) { /* a further pass of this individual */
				fetch_geno(i, j + stride, pa0, pa1, prw);
				fetch_rows(j + stride);
			}
			int z0 = 0xff, z1 = 0xff;
			STAMP(i, 6);
			if (valid && covered) {
				if (init_flag) {
					z0 = bucket_fast<KMAX>(x0, icum, 1.0, K);
					z1 = bucket_fast<KMAX>(x1, icum, 1.0, K);
				} else {
					bool amb0 = true, amb1 = true;
					if (PRE) {
						z0 = bucket_f32<KMAX>((float)x0, F0, qf, K, &amb0);
						z1 = bucket_f32<KMAX>((float)x1, F1, qf, K, &amb1);
					}
					if (amb0) {
						double cum[KMAX];
						double tot = weights<KMAX>(d.freq + ((size_t)j * d.Amax + a0) * d.KP, q, cum, K);
						z0 = bucket_fast<KMAX>(x0, cum, tot, K);
					}
					if (amb1) {
						double cum[KMAX];
						double tot = weights<KMAX>(d.freq + ((size_t)j * d.Amax + a1) * d.KP, q, cum, K);
						z1 = bucket_fast<KMAX>(x1, cum, tot, K);
					}
				}
			}
			STAMP(i, 7);
#pragma unroll
			for (int m = 0; m < KMAX; m++)
				if (m < K) wcnt[m] += __popcll(__ballot(z0 == m)) + __popcll(__ballot(z1 == m));
			if (j < d.Lp) *(unsigned short *)(d.z + (size_t)i * rowb + (size_t)j * 2) = (unsigned short)(z0 | (z1 << 8));
		}
		STAMP(i, 1);
		if (wmode) { /* few workgroups: every wave hands its counts over itself, no reduction inside the workgroup first */
			unsigned long long v = (unsigned long long)tag << 48;
#pragma unroll
			for (int m = 0; m < KMAX; m++)
				if (m < K && (int)lane_id() == m / 3) v |= (unsigned long long)(wcnt[m] & 0xffff) << (16 * (m % 3));
			if ((int)lane_id() < W) { if (local) st_xcd(&cb->gran[slot][(g * (BLOCK / 64) + (t >> 6)) * W + (int)lane_id()], v); else st_agent(&cb->gran[slot][(g * (BLOCK / 64) + (t >> 6)) * W + (int)lane_id()], v); }
		} else {
			if (lane_id() == 0) {
#pragma unroll
				for (int m = 0; m < KMAX; m++)
					if (m < K && wcnt[m]) atomicAdd(&sh.hist[par][m], wcnt[m]);
			}
			lds_barrier();
			/* this workgroup's counts leave; everybody else's are collected */
			if (t < W) { /* `pack` counts of `bits` bits per word */
				unsigned long long v = (unsigned long long)tag << 48;
				for (int c3 = 0; c3 < pack; c3++)
					if (pack * t + c3 < K) v |= (unsigned long long)(sh.hist[par][pack * t + c3] & ((1 << bits) - 1)) << (bits * c3);
				if (local) st_xcd(&cb->gran[slot][g * W + t], v); else st_agent(&cb->gran[slot][g * W + t], v);
			}
		}
#ifndef ISG_EXP_XWAIT
		STAMP(i, 2);
#endif
		/* while they travel: the frequency rows of the next individual's first locus (its genotype bytes arrived
		 * during the draws) and a touch of the tape lines it will read */
		pa0 = na0; pa1 = na1; prw = nrw;
		fetch_rows(g * BLOCK + t);
		if (covered && i + 1 < d.N && lane_id() < 10) touch2 = d.tape[offi + 2ull * (unsigned)nvalid + 2ull * prw + 16u * lane_id()];
		if (!covered && t == 0) cb->overflow_flag = 1;
		const int ngran = wmode ? G * (BLOCK / 64) * W : G * W;
		for (int gi = t; gi - t < ngran; gi += BLOCK) { /* wave-uniform trip count */
			if (gi < ngran && (wmode || gi / W != g)) {
				const unsigned long long v = coop_poll(&cb->gran[slot][gi], tag, cb);
				const int w = gi % W;
				for (int c3 = 0; c3 < pack; c3++) {
					const int m = pack * w + c3;
					const int c = (int)((v >> (bits * c3)) & ((1 << bits) - 1));
					if (m < K && c) atomicAdd(&sh.ghist[par][m], c);
				}
			}
		}
		lds_barrier();
#ifndef ISG_EXP_XWAIT
		STAMP(i, 3);
#endif
		const unsigned used = 2u * (unsigned)nvalid +
			dirichlet_coop<BLOCK, KMAX>(d, sh, i, cur, offi + 2ull * (unsigned)nvalid, alpha, par,
						    covered ? d.tape + offi + 2ull * (unsigned)nvalid : nullptr, writer);
		off += used;
	}
	if (g == 0 && t == 0) *pos_out = off;
}

/*
 * The Dirichlet of k_zq_spec (K <= 8): every WAVE evaluates the attempt table for itself -- lane (m, d) the attempt of
 * gamma m at stream offset m + d (gamma m cannot start before m earlier gammas took an attempt each; 64 / K offsets
 * per gamma) -- and gets the accepted / irregular masks from two ballots, so the walk needs no LDS round trip and no
 * barrier.  Walks that leave the table (1 % at K = 5) are redone sequentially by lane 0 of each wave.
 * Counts: sh.hist3[buf] + sh.ghist3[buf]; the caller's barrier made them complete.
 */
/* cntf(m) = the individual's count of cluster m; tab = the skip-ahead tables (LDS) */
template <int KMAX, class CntF>
__device__ __forceinline__ unsigned dirichlet_wave_f(const DevView &d, const isg_wh_tables *tab, CntF cntf, int i, const isg_wh &cur, unsigned long long dstart_off,
						    double alpha, const double *dtape, bool writer, bool pre = false, double ppu0 = 0.0, double ppu1 = 0.0)
{
	const int K = d.K, lane = (int)lane_id(), D = 64 / K;
	const int m = (lane < K * D) ? lane / D : K - 1, dd = lane - m * D;
	const bool act = lane < K * D;
	const int cnt = cntf(m);
	const double a = (double)cnt + alpha;
	isg_cursor c;
	c.used = 0;
	c.tape = dtape ? dtape + 2 * (m + dd) : nullptr;
	if (!dtape) c.s = isg_wh_jump(tab, cur, dstart_off + 2ull * (unsigned)(m + dd));
	double r = -1;
	if (act) {
		double pu0, pu1;
		if (pre && dtape) { /* the caller fetched this lane's first two uniforms (dtape[2 (m + dd)], [.. + 1]) ahead of time */
			pu0 = ppu0;
			pu1 = ppu1;
			c.used = 2;
		} else {
			pu0 = isg_cur_next(&c);
			pu1 = isg_cur_next(&c);
		}
		if (a < 1) r = rgamma1_try_pre(pu0, pu1, a);
		else if (a > 1) r = rgamma2_try_dev(&c, pu0, pu1, a);
		else c.used = 255; /* shape 1: odd consumption */
	}
	const unsigned long long A = __ballot(act && !(r < 0)), I = __ballot(act && c.used != 2);
	const unsigned dmask = (D >= 32) ? 0xffffffffu : ((1u << D) - 1u);
	unsigned fo = 0; /* attempts consumed so far = stream offset / 2 */
	bool bad = false;
	double v = 0.0;  /* lane mm < K ends up with gamma mm's value */
#pragma unroll
	for (int mm = 0; mm < KMAX; mm++) {
		if (mm < K) {
			const unsigned Am = (unsigned)(A >> (mm * D)) & dmask, Im = (unsigned)(I >> (mm * D)) & dmask;
			const unsigned rel = fo - (unsigned)mm; /* fo >= mm always */
			const unsigned rest = (rel < (unsigned)D) ? ((Am | Im) >> rel) : 0u;
			const unsigned e = rel + (rest ? (unsigned)__builtin_ctz(rest) : 0u);
			bad |= (rest == 0u) || (((Im >> (e & 31u)) & 1u) != 0u);
			const double val = readlane_f64(r, (mm * D + (int)(e < (unsigned)D ? e : 0u)) & 63);
			v = (lane == mm) ? val : v;
			fo = (unsigned)mm + e + 1u;
		}
	}
	unsigned used = 2u * fo;
	if (bad) { /* an irregular attempt on the path (a retry inside it: more than two uniforms): the general walk */
		const unsigned ul = (unsigned)c.used;
		unsigned go = 0;
		int gm = 0;
		bool ok = true;
		while (gm < K) {
			const unsigned Am = (unsigned)(A >> (gm * D)) & dmask, Im = (unsigned)(I >> (gm * D)) & dmask;
			const unsigned rel = go - (unsigned)gm;
			if (rel >= (unsigned)D) { ok = false; break; }
			const unsigned rest = (Am | Im) >> rel;
			if (rest == 0u) { ok = false; break; }
			const unsigned e = rel + (unsigned)__builtin_ctz(rest);
			const int idx = gm * D + (int)e;
			unsigned step = 1;
			if ((Im >> e) & 1u) {
				const unsigned u = (unsigned)__builtin_amdgcn_readlane((int)ul, idx);
				if (u == 255u || (u & 1u)) { ok = false; break; }
				step = u >> 1;
			}
			go = (unsigned)gm + e + step;
			if ((Am >> e) & 1u) {
				const double val = readlane_f64(r, idx);
				v = (lane == gm) ? val : v;
				gm++;
			}
		}
		bad = !ok;
		used = 2u * go;
	}
	if (bad) { /* wave-uniform: the plain loop from the Dirichlet's first position */
		double g[KMAX];
		unsigned u = 0;
		if (lane == 0) {
			isg_cursor q;
			q.used = 0;
			q.tape = nullptr;
			q.s = isg_wh_jump(tab, cur, dstart_off);
#pragma unroll
			for (int mm = 0; mm < KMAX; mm++) g[mm] = (mm < K) ? isg_rgamma(&q, (double)cntf(mm) + alpha) : 0.0;
			u = q.used;
		} else {
#pragma unroll
			for (int mm = 0; mm < KMAX; mm++) g[mm] = 0.0;
		}
		used = (unsigned)__builtin_amdgcn_readfirstlane((int)u);
#pragma unroll
		for (int mm = 0; mm < KMAX; mm++) {
			const double val = readlane_f64(g[mm], 0);
			v = (lane == mm) ? val : v;
		}
	}
	if (writer) { /* qq[i] = g / sum with the sum taken in stream order (random.c:272-279) */
		double sum = 0;
		for (int k2 = 0; k2 < K; k2++) sum += readlane_f64(v, k2);
		if (lane < K) {
			d.qq[(size_t)i * K + lane] = v / sum;
			d.qqnum[(size_t)i * K + lane] = cntf(lane);
		}
	}
	return used;
}
template <int KMAX>
__device__ __forceinline__ unsigned dirichlet_wave(const DevView &d, ZqShared &sh, int i, const isg_wh &cur, unsigned long long dstart_off,
						  double alpha, int buf, const double *dtape, bool writer, bool pre = false, double ppu0 = 0.0, double ppu1 = 0.0)
{
	return dirichlet_wave_f<KMAX>(d, &sh.tab, [&](int m) { return sh.hist3[buf][m] + sh.ghist3[buf][m]; }, i, cur, dstart_off, alpha, dtape, writer, pre, ppu0, ppu1);
}

/*
 * k_zq_spec: k_zq_coop with the Z draws taken off the critical path (single pass: one locus per lane, K <= 8).
 * Individual i+1 starts used_i uniforms behind the end of individual i's draws, and used_i = 2 K + 2 c where c is
 * the number of rejected gamma attempts of i's Dirichlet -- almost always 0..3.  While the counts of individual i
 * travel between the workgroups, every lane draws its two Z of individual i+1 for each of these ISG_SPEC_C start
 * positions (the running sums are shared, a candidate costs a multiply and K-1 compares per copy).  When the
 * Dirichlet of i is done the matching candidate is picked; any other consumption takes the plain path of
 * k_zq_coop for that individual.  Same Z, same counts, same consumption in every case.
 */
#ifndef ISG_SPEC_C
#define ISG_SPEC_C 6
#endif
template <int KMAX>
__global__ void __launch_bounds__(256) k_zq_spec(DevView d, isg_wh base, double alpha, CoopBuf *cb, uint64_t *pos_out, int xcd_pack)
{
	constexpr int BLOCK = 256, C = ISG_SPEC_C;
	static_assert(KMAX <= 8 && C <= 8, "pre-filter rows in registers; candidates packed 4 bits each");
	__shared__ ZqShared sh;
	__shared__ unsigned same_xcd;
	if (xcd_pack && (blockIdx.x & 7)) return; /* every 8th block works: one XCD under round-robin placement */
	const int t = threadIdx.x, g = xcd_pack ? blockIdx.x >> 3 : blockIdx.x, G = xcd_pack ? gridDim.x >> 3 : gridDim.x, K = d.K, lane = (int)lane_id();
	{
		const uint16_t *src = (const uint16_t *)d.tab;
		uint16_t *dst = (uint16_t *)&sh.tab;
		for (int k = t; k < (int)(sizeof(isg_wh_tables) / 2); k += BLOCK) dst[k] = src[k];
		if (t < 2 * ISG_KCAP) {
			(&sh.hist[0][0])[t] = 0;
			(&sh.amask[0][0])[t] = 0;
			(&sh.rmask[0][0])[t] = 0;
			(&sh.ghist[0][0])[t] = 0;
		}
		if (t < 3 * ISG_KCAP) {
			(&sh.hist3[0][0])[t] = 0;
			(&sh.ghist3[0][0])[t] = 0;
		}
	}
	__syncthreads();
	const bool local = xcd_pack && coop_same_xcd(cb, g, G, &same_xcd);
	const isg_wh cur = isg_wh_jump(&sh.tab, base, 0);
	unsigned long long off = 0;
	const size_t rowb = (size_t)d.Lp * 2;
	const bool writer = (g == 0) && (t >= BLOCK - 64);
	const bool wmode = (G * (BLOCK / 64) * ((K + 2) / 3) <= BLOCK);
	/* counts per exchanged word: 3 x 16 bits, or 4 x 12 bits when a workgroup's count of one bucket stays below 4096 */
	const int npass = (d.Lp + G * BLOCK - 1) / (G * BLOCK);
	const int pack = (!wmode && 2 * BLOCK * npass < 4096) ? 4 : 3, bits = (pack == 4) ? 12 : 16, W = (K + pack - 1) / pack;
	const int j = g * BLOCK + t; /* this lane's locus */
	double touch = 0.0;
	/* three individuals in flight per lane: cl = the one being finished, nl = the one whose candidates are drawn,
	 * ml = the one whose data is being fetched */
	/* the individual's qq row and locus count travel as ONE vector load each (lane m holds qq[m]; they are spread
	 * with v_readlane when used): a scalar load here would sit in the same counter as the LDS traffic and stall the
	 * next LDS wait for a cold HBM access */
	struct Loc {
		unsigned a0, a1, rw;
		float F0[KMAX], F1[KMAX];
		double qv;
		int nvv;
	};
	Loc cl, nl, ml;
	auto fetch_geno = [&](int ni, Loc &L) {
		L.a0 = L.a1 = 0xff;
		L.rw = 0;
		L.nvv = 0;
		L.qv = 0.0;
		if (ni < d.N) {
			L.nvv = d.nvalid[min(ni + lane, d.N - 1)];
			if (lane < K) L.qv = d.qq[(size_t)ni * K + lane];
			if (j < d.Lp) {
				const unsigned short gg = *(const unsigned short *)(d.geno + (size_t)ni * rowb + (size_t)j * 2);
				L.a0 = gg & 0xff;
				L.a1 = gg >> 8;
				L.rw = d.rankwave[(size_t)ni * d.nwv + (j >> 6)];
			}
		}
	};
	auto fetch_rows = [&](Loc &L) {
#pragma unroll
		for (int m = 0; m < KMAX; m++) L.F0[m] = L.F1[m] = 0.f;
		if (L.a0 != 0xff) {
			const float *P0 = d.freqf + ((size_t)j * d.Amax + L.a0) * d.KPF, *P1 = d.freqf + ((size_t)j * d.Amax + L.a1) * d.KPF;
#pragma unroll
			for (int m = 0; m < KMAX; m += 4) {
				if (m < K) {
					const float4 f0 = *(const float4 *)(P0 + m), f1 = *(const float4 *)(P1 + m);
					L.F0[m] = f0.x; L.F1[m] = f1.x;
					if (m + 1 < KMAX) { L.F0[m + 1] = f0.y; L.F1[m + 1] = f1.y; }
					if (m + 2 < KMAX) { L.F0[m + 2] = f0.z; L.F1[m + 2] = f1.z; }
					if (m + 3 < KMAX) { L.F0[m + 3] = f0.w; L.F1[m + 3] = f1.w; }
				}
			}
		}
	};
	fetch_geno(0, cl);
	fetch_rows(cl);
	fetch_geno(1, nl);
	ml = nl;
	/* candidates of the individual about to be finished: buckets packed 4 bits each, ambiguity bits, base offset */
	unsigned cz0 = 0, cz1 = 0, camb = 0;
	unsigned long long cbase = 0;
	bool cvalid = false;
	for (int i = 0; i < d.N; i++) {
		const unsigned tag = (unsigned)(i % 65535) + 1u;
		const int slot = i & (ISG_COOP_RING - 1), buf = i % 3;
		if (touch == -1.0) cb->overflow_flag = 2;
		const int nvalid = __builtin_amdgcn_readfirstlane(cl.nvv), nnvalid = __builtin_amdgcn_readfirstlane(nl.nvv);
		const unsigned long long offi = off;
		const bool covered = offi + 2ull * (unsigned)nvalid + 1024ull <= d.tape_len;
		const bool valid = (cl.a0 != 0xff);
		const unsigned rank = cl.rw + (unsigned)__popcll(__ballot(valid) & ((1ull << lane) - 1ull));
		STAMP(i, 0);
		/* ---- first in the queue: the uniforms of the NEXT individual's candidates and its frequency rows ---- */
		const unsigned long long nbase = offi + 2ull * (unsigned)nvalid + 2ull * (unsigned)K; /* every gamma: >= one attempt of two uniforms */
		const bool nvalidc = (i + 1 < d.N) && (nbase + 2ull * (C - 1) + 2ull * (unsigned)nnvalid + 1024ull <= d.tape_len);
		const bool nvalidl = (nl.a0 != 0xff);
		const unsigned nrank = nl.rw + (unsigned)__popcll(__ballot(nvalidl) & ((1ull << lane) - 1ull));
		double xs[2 * C];
#pragma unroll
		for (int k = 0; k < 2 * C; k++) xs[k] = 0.5;
		if (nvalidc && nvalidl) {
			const double *tp = d.tape + nbase + 2ull * nrank;
#pragma unroll
			for (int k = 0; k < 2 * C; k++) xs[k] = tp[k];
		}
		fetch_rows(nl);
		if (covered && lane < 10) touch = d.tape[offi + 2ull * (unsigned)nvalid + 16u * (unsigned)lane]; /* this Dirichlet's stretch */
		/* ---- this individual's Z: a candidate drawn earlier, or the plain path ---- */
		const unsigned long long dc = offi - cbase;
		const bool hit = cvalid && offi >= cbase && !(dc & 1ull) && dc < 2ull * C;
		int z0 = 0xff, z1 = 0xff;
		bool amb0 = false, amb1 = false;
		if (hit) {
			const unsigned c = (unsigned)(dc >> 1);
			if (valid) {
				z0 = (int)((cz0 >> (4 * c)) & 0xfu);
				z1 = (int)((cz1 >> (4 * c)) & 0xfu);
				amb0 = (camb >> c) & 1u;
				amb1 = (camb >> (8 + c)) & 1u;
			}
		} else if (valid && covered) {
			float qf[KMAX];
#pragma unroll
			for (int m = 0; m < KMAX; m++) qf[m] = (m < K) ? (float)readlane_f64(cl.qv, m) : 0.f;
			const double x0 = d.tape[offi + 2ull * rank], x1 = d.tape[offi + 2ull * rank + 1];
			z0 = bucket_f32<KMAX>((float)x0, cl.F0, qf, K, &amb0);
			z1 = bucket_f32<KMAX>((float)x1, cl.F1, qf, K, &amb1);
		}
		if (__ballot(valid && covered && (amb0 || amb1))) { /* rare: the draw in double */
			double cum[KMAX], q[KMAX];
#pragma unroll
			for (int m = 0; m < KMAX; m++) q[m] = (m < K) ? readlane_f64(cl.qv, m) : 0.0;
			if (valid && covered && amb0) {
				const double tot = weights<KMAX>(d.freq + ((size_t)j * d.Amax + cl.a0) * d.KP, q, cum, K);
				z0 = bucket_fast<KMAX>(d.tape[offi + 2ull * rank], cum, tot, K);
			}
			if (valid && covered && amb1) {
				const double tot = weights<KMAX>(d.freq + ((size_t)j * d.Amax + cl.a1) * d.KP, q, cum, K);
				z1 = bucket_fast<KMAX>(d.tape[offi + 2ull * rank + 1], cum, tot, K);
			}
		}
		if (!covered) { z0 = z1 = 0xff; }
		int wcnt[KMAX];
#pragma unroll
		for (int m = 0; m < KMAX; m++) wcnt[m] = (m < K) ? __popcll(__ballot(z0 == m)) + __popcll(__ballot(z1 == m)) : 0;
		STAMP(i, 1);
		/* ---- counts leave ---- */
		if (wmode) {
			unsigned long long v = (unsigned long long)tag << 48;
#pragma unroll
			for (int m = 0; m < KMAX; m++)
				if (m < K && lane == m / 3) v |= (unsigned long long)(wcnt[m] & 0xffff) << (16 * (m % 3));
			if (lane < W) { if (local) st_xcd(&cb->gran[slot][(g * (BLOCK / 64) + (t >> 6)) * W + lane], v); else st_agent(&cb->gran[slot][(g * (BLOCK / 64) + (t >> 6)) * W + lane], v); }
		} else {
			if (lane == 0) {
#pragma unroll
				for (int m = 0; m < KMAX; m++)
					if (m < K && wcnt[m]) atomicAdd(&sh.hist3[buf][m], wcnt[m]);
			}
			lds_barrier();
			if (t < W) { /* `pack` counts of `bits` bits per word */
				unsigned long long v = (unsigned long long)tag << 48;
				for (int c3 = 0; c3 < pack; c3++)
					if (pack * t + c3 < K) v |= (unsigned long long)(sh.hist3[buf][pack * t + c3] & ((1 << bits) - 1)) << (bits * c3);
				if (local) st_xcd(&cb->gran[slot][g * W + t], v); else st_agent(&cb->gran[slot][g * W + t], v);
			}
		}
		if (j < d.Lp) *(unsigned short *)(d.z + (size_t)i * rowb + (size_t)j * 2) = (unsigned short)(z0 | (z1 << 8));
		STAMP(i, 2);
		if (!covered && t == 0) cb->overflow_flag = 1;
		/* ---- while they travel: the data of the individual after the next is requested (cold: it arrives under
		 * the arithmetic below, before the polling loads queue up behind it), then the next one's candidates ---- */
		fetch_geno(i + 2, ml);
		cbase = nbase;
		cvalid = nvalidc;
		cz0 = cz1 = camb = 0;
		if (nvalidc && nvalidl) {
			float qf[KMAX], cum0[KMAX], cum1[KMAX];
#pragma unroll
			for (int m = 0; m < KMAX; m++) qf[m] = (m < K) ? (float)readlane_f64(nl.qv, m) : 0.f;
			const float run0 = prefix_f32<KMAX>(nl.F0, qf, K, cum0), run1 = prefix_f32<KMAX>(nl.F1, qf, K, cum1);
#pragma unroll
			for (int c = 0; c < C; c++) {
				bool a0, a1;
				const int b0 = bucket_from_cum<KMAX>((float)xs[2 * c], cum0, run0, K, &a0);
				const int b1 = bucket_from_cum<KMAX>((float)xs[2 * c + 1], cum1, run1, K, &a1);
				cz0 |= (unsigned)b0 << (4 * c);
				cz1 |= (unsigned)b1 << (4 * c);
				camb |= (a0 ? 1u : 0u) << c;
				camb |= (a1 ? 1u : 0u) << (8 + c);
			}
		}
		STAMP(i, 6);
		/* ---- everybody's counts ---- */
		const int ngran = wmode ? G * (BLOCK / 64) * W : G * W;
		for (int gi = t; gi - t < ngran; gi += BLOCK) { /* wave-uniform trip count */
			if (gi < ngran && (wmode || gi / W != g)) {
				const unsigned long long v = coop_poll(&cb->gran[slot][gi], tag, cb);
				const int w = gi % W;
				for (int c3 = 0; c3 < pack; c3++) {
					const int m = pack * w + c3;
					const int c = (int)((v >> (bits * c3)) & ((1 << bits) - 1));
					if (m < K && c) atomicAdd(&sh.ghist3[buf][m], c);
				}
			}
		}
		lds_barrier();
		STAMP(i, 3);
		/* the buffer of individual i + 2 (last read before this barrier, next written after the next one) */
		if (t < KMAX) {
			sh.hist3[(i + 2) % 3][t] = 0;
			sh.ghist3[(i + 2) % 3][t] = 0;
		}
		const unsigned used = 2u * (unsigned)nvalid +
			dirichlet_wave<KMAX>(d, sh, i, cur, offi + 2ull * (unsigned)nvalid, alpha, buf,
					     covered ? d.tape + offi + 2ull * (unsigned)nvalid : nullptr, writer);
		STAMP(i, 5);
		off += used;
		cl = nl;
		nl = ml;
	}
	if (g == 0 && t == 0) *pos_out = off;
}

/* the same sum in every lane: rotations inside the rows of 16 lanes, then the rows through two cross-lane exchanges */
__device__ __forceinline__ unsigned wave_allsum_u32(unsigned x)
{
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xf, 0xf, false); /* row_ror:1 */
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xf, 0xf, false); /* row_ror:2 */
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xf, 0xf, false); /* row_ror:4 */
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false); /* row_ror:8 */
	x += (unsigned)__shfl_xor((int)x, 16, 64);
	x += (unsigned)__shfl_xor((int)x, 32, 64);
	return x;
}

/*
 * k_zq_pipe: the replay update_ZQ chain with the exchange of an individual's counts taken off the critical path.
 * Every workgroup has DW draw waves (one locus per lane) and ONE control wave; nothing but LDS words joins them.
 *   draw waves:   individual i starts where the control wave says (LDS).  Its Z is the candidate drawn earlier for
 *                 that start position (92 %), or the plain path.  Then the C candidates of individual i+1 are drawn
 *                 (start = end of i's draws + 2 K + 2 c, c = rejected attempts of i's Dirichlet), the few draws the
 *                 single precision filter could not decide are redone in double, and the wave's counts of EVERY
 *                 candidate leave as tagged words -- before anybody knows which candidate it will be.
 *   control wave: knows c when its Dirichlet of i-1 is done; the counts of individual i for that candidate were
 *                 published while that Dirichlet and the exchange before it ran, so they have arrived or are about
 *                 to; it sums them (packed 16-bit fields add without carries: a total stays below 2 Lp < 65536), runs
 *                 the Dirichlet of i (dirichlet_wave) and posts the next start position.
 * A start position no candidate was drawn for (8 %: c >= C, a shape of exactly 1, the first individual) takes the
 * plain path: the draw waves draw Z then, publish the counts in a set of their own, and the control wave waits for
 * those.  Same Z, same counts, same consumption as k_zq_spec / k_zq_coop in every case.
 * Granules: pipe_gran(slot of the individual, publishing wave, set * W + w); set = candidate, or C for the plain path.
 */
#ifndef ISG_PIPE_DW
#define ISG_PIPE_DW 3
#endif
#define ISG_PIPE_RMAX 6
#ifndef ISG_PIPE_C
#define ISG_PIPE_C 5 /* candidates per individual: 4 / 5 / 6 / 7 measured 9.78 / 9.52 / 9.70 / 9.96 k cycles per individual at config 3 */
#endif
#ifdef ISG_STAMPS
#define STAMPC(i, k) do { if (threadIdx.x == 64 * DW && blockIdx.x == 0 && (i) < 4096) g_stamps[(i) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMPC(i, k) do { } while (0)
#endif
/* granules of k_zq_pipe: every publishing wave has lines of its own, 4352 bytes from the next wave's (4 KiB + 256 B:
 * consecutive publishers fall on different HBM stacks AND channels). */
#define ISG_PIPE_STRIDE 544 /* 8-byte words */
__device__ __forceinline__ unsigned long long *pipe_gran(unsigned long long *pg, int slot, int NP, int p, int w)
{
	return pg + ((size_t)slot * NP + p) * ISG_PIPE_STRIDE + w;
}
template <int KMAX, int DW>
__global__ void __launch_bounds__(64 * (DW + 1)) k_zq_pipe(DevView d, isg_wh base, double alpha, CoopBuf *cb, unsigned long long *pg, uint64_t *pos_out, int xcd_pack)
{
	constexpr int BLOCK = 64 * (DW + 1), C = ISG_PIPE_C, LW = 64 * DW;
	static_assert(KMAX <= 8 && C <= 8, "pre-filter rows in registers; candidates packed 4 bits each");
	__shared__ ZqShared sh;
	__shared__ unsigned long long off_sh[4]; /* start position of individual i in off_sh[i & 3] ... */
	__shared__ unsigned seq_sh;              /* ... valid once seq_sh >= i */
	__shared__ unsigned same_xcd;
	if (xcd_pack && (blockIdx.x & 7)) return; /* every 8th block works: one XCD under round-robin placement */
	const int t = threadIdx.x, g = xcd_pack ? blockIdx.x >> 3 : blockIdx.x, G = xcd_pack ? gridDim.x >> 3 : gridDim.x, K = d.K, lane = (int)lane_id();
	const bool ctrl = (t >= LW);
	{
		const uint16_t *src = (const uint16_t *)d.tab;
		uint16_t *dst = (uint16_t *)&sh.tab;
		for (int k = t; k < (int)(sizeof(isg_wh_tables) / 2); k += BLOCK) dst[k] = src[k];
		if (t < 3 * ISG_KCAP) {
			(&sh.hist3[0][0])[t] = 0;
			(&sh.ghist3[0][0])[t] = 0;
		}
		if (t < 4) off_sh[t] = 0;
		if (t == 4) seq_sh = 0;
	}
	__syncthreads();
	/* all workgroups on one XCD (checked through the hardware register): the counts are handed over in its L2 */
	const bool local = xcd_pack && coop_same_xcd(cb, g, G, &same_xcd);
	const isg_wh cur = isg_wh_jump(&sh.tab, base, 0);
	const int W = (K + 2) / 3, NP = G * DW;
	const size_t rowb = (size_t)d.Lp * 2;
	if (ctrl) {
		/* ------------------------------- control wave ------------------------------- */
		const bool writer = (g == 0);
		unsigned long long off = 0, pcbase = 0;
		bool pcvalid = false;
		int nvv = d.nvalid[min(lane, d.N - 1)], nvn = d.nvalid[min(64 + lane, d.N - 1)];
		for (int i = 0; i < d.N; i++) {
			if (i && !(i & 63)) {
				nvv = nvn;
				nvn = d.nvalid[min(i + 64 + lane, d.N - 1)];
			}
			const int nvalid = __builtin_amdgcn_readlane(nvv, i & 63);
			const int nvnext = ((i + 1) & 63) ? __builtin_amdgcn_readlane(nvv, (i + 1) & 63) : __builtin_amdgcn_readlane(nvn, 0);
			const unsigned tag = (unsigned)(i % 65535) + 1u;
			const int slot = i & (ISG_COOP_RING - 1);
			const unsigned long long offi = off, dpos = offi + 2ull * (unsigned)nvalid;
			const bool covered = dpos + 1024ull <= d.tape_len;
			/* which set of counts: the draw waves decide the same way from the same numbers */
			const unsigned long long dc = offi - pcbase;
			const bool hit = pcvalid && offi >= pcbase && !(dc & 1ull) && dc < 2ull * C;
			const int set = hit ? (int)(dc >> 1) : C;
			STAMPC(i, 4);
			/* this lane's attempt of the Dirichlet (gamma m at offset m + dd, as dirichlet_wave lays them out): its two
			 * uniforms are fetched now, under the exchange (the draw waves touched these lines an individual ago) */
			double ppu0 = 0.0, ppu1 = 0.0;
			if (covered) {
				const int D = 64 / K, m = (lane < K * D) ? lane / D : K - 1, dd = lane - m * D;
				ppu0 = d.tape[dpos + 2 * (m + dd)]; /* (a shape of exactly 1 makes the position odd: no 16-byte load) */
				ppu1 = d.tape[dpos + 2 * (m + dd) + 1];
			}
			/* everybody's counts: the probes of all missing granules go out together; every spin is bounded and watches
			 * the common abort word */
			unsigned long long v[3][ISG_PIPE_RMAX];
#pragma unroll
			for (int w = 0; w < 3; w++)
#pragma unroll
				for (int r = 0; r < ISG_PIPE_RMAX; r++) {
					const int p = r * 64 + lane;
					v[w][r] = (w < W && r * 64 < NP && p < NP) ? 0ull : ((unsigned long long)tag << 48);
				}
			for (unsigned spin = 0;; spin++) {
#pragma unroll
				for (int w = 0; w < 3; w++)
#pragma unroll
					for (int r = 0; r < ISG_PIPE_RMAX; r++) {
						const int p = r * 64 + lane;
						if (w < W && r * 64 < NP && p < NP && (unsigned)(v[w][r] >> 48) != tag) v[w][r] = ld_agent(pipe_gran(pg, slot, NP, p, set * W + w));
					}
				bool miss = false;
#pragma unroll
				for (int w = 0; w < 3; w++)
#pragma unroll
					for (int r = 0; r < ISG_PIPE_RMAX; r++)
						miss |= ((unsigned)(v[w][r] >> 48) != tag);
				if (!__ballot(miss)) break;
				if ((spin & 1023u) == 1023u) {
					if (__hip_atomic_load(&cb->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
					if (spin > (1u << 22)) {
						__hip_atomic_store(&cb->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						break;
					}
				}
			}
			unsigned lo[3] = {0u, 0u, 0u}, hi[3] = {0u, 0u, 0u};
#pragma unroll
			for (int w = 0; w < 3; w++)
#pragma unroll
				for (int r = 0; r < ISG_PIPE_RMAX; r++)
					if (w < W && r * 64 < NP) {
						lo[w] += (unsigned)v[w][r];
						hi[w] += (unsigned)(v[w][r] >> 32) & 0xffffu;
					}
			unsigned cnt = 0;
#pragma unroll
			for (int w = 0; w < 3; w++)
				if (w < W) {
					const unsigned Ls = wave_sum_u32(lo[w]), Hs = wave_sum_u32(hi[w]);
					cnt = (lane == 3 * w) ? (Ls & 0xffffu) : (lane == 3 * w + 1) ? (Ls >> 16) : (lane == 3 * w + 2) ? Hs : cnt;
				}
			if (lane < K) sh.ghist3[0][lane] = (int)cnt;
			STAMPC(i, 3);
			const unsigned used = 2u * (unsigned)nvalid +
				dirichlet_wave<KMAX>(d, sh, i, cur, dpos, alpha, 0, covered ? d.tape + dpos : nullptr, writer, true, ppu0, ppu1);
			off += used;
			if (lane == 0) {
				off_sh[(i + 1) & 3] = off;
				__hip_atomic_store(&seq_sh, (unsigned)(i + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
			/* the candidates the draw waves are drawing for individual i + 1 */
			pcbase = dpos + 2ull * (unsigned)K;
			pcvalid = (i + 1 < d.N) && (pcbase + 2ull * (C - 1) + 2ull * (unsigned)nvnext + 1024ull <= d.tape_len);
			STAMPC(i, 5);
		}
		if (g == 0 && lane == 0) *pos_out = off;
#ifdef ISG_STAMPS
		if (g == 0 && lane == 0) g_stamps[4095 * 8 + 4] = (local ? 1ull : 0ull) | ((unsigned long long)xcc_id() << 8) | ((unsigned long long)G << 16);
#endif
		return;
	}
	/* --------------------------------- draw waves --------------------------------- */
	const int j = g * LW + t; /* this lane's locus */
	const int pub = g * DW + (t >> 6);
	const int myc = lane / W, myw = lane % W; /* the candidate and the word this lane publishes */
	struct Loc {
		unsigned a0, a1, rw;
		float F0[KMAX], F1[KMAX];
		double qv;
		int nvv;
	};
	Loc cl, nl, ml;
	auto fetch_geno = [&](int ni, Loc &L) {
		L.a0 = L.a1 = 0xff;
		L.rw = 0;
		L.nvv = 0;
		L.qv = 0.0;
		if (ni < d.N) {
			L.nvv = d.nvalid[min(ni + lane, d.N - 1)];
			if (lane < K) L.qv = d.qq[(size_t)ni * K + lane];
			if (j < d.Lp) {
				const unsigned short gg = *(const unsigned short *)(d.geno + (size_t)ni * rowb + (size_t)j * 2);
				L.a0 = gg & 0xff;
				L.a1 = gg >> 8;
				L.rw = d.rankwave[(size_t)ni * d.nwv + (j >> 6)];
			}
		}
	};
	auto fetch_rows = [&](Loc &L) {
#pragma unroll
		for (int m = 0; m < KMAX; m++) L.F0[m] = L.F1[m] = 0.f;
		if (L.a0 != 0xff) {
			const float *P0 = d.freqf + ((size_t)j * d.Amax + L.a0) * d.KPF, *P1 = d.freqf + ((size_t)j * d.Amax + L.a1) * d.KPF;
#pragma unroll
			for (int m = 0; m < KMAX; m += 4) {
				if (m < K) {
					const float4 f0 = *(const float4 *)(P0 + m), f1 = *(const float4 *)(P1 + m);
					L.F0[m] = f0.x; L.F1[m] = f1.x;
					if (m + 1 < KMAX) { L.F0[m + 1] = f0.y; L.F1[m + 1] = f1.y; }
					if (m + 2 < KMAX) { L.F0[m + 2] = f0.z; L.F1[m + 2] = f1.z; }
					if (m + 3 < KMAX) { L.F0[m + 3] = f0.w; L.F1[m + 3] = f1.w; }
				}
			}
		}
	};
	fetch_geno(0, cl);
	fetch_rows(cl);
	fetch_geno(1, nl);
	ml = nl;
	unsigned cz0 = 0, cz1 = 0;
	unsigned long long cbase = 0;
	bool cvalid = false;
	double touch = 0.0;
	for (int i = 0; i < d.N; i++) {
		const unsigned tag = (unsigned)(i % 65535) + 1u, ntag = (unsigned)((i + 1) % 65535) + 1u;
		const int slot = i & (ISG_COOP_RING - 1), nslot = (i + 1) & (ISG_COOP_RING - 1);
		/* the start position, posted by the control wave when its Dirichlet of individual i - 1 was done */
		while (__hip_atomic_load(&seq_sh, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)i) __builtin_amdgcn_s_sleep(1);
		const unsigned long long offi = off_sh[i & 3];
		const int nvalid = __builtin_amdgcn_readfirstlane(cl.nvv), nnvalid = __builtin_amdgcn_readfirstlane(nl.nvv);
		const bool covered = offi + 2ull * (unsigned)nvalid + 1024ull <= d.tape_len;
		const bool valid = (cl.a0 != 0xff);
		STAMP(i, 0);
		/* ---- first in the queue: the uniforms of the next individual's candidates and its frequency rows ---- */
		const unsigned long long nbase = offi + 2ull * (unsigned)nvalid + 2ull * (unsigned)K; /* every gamma: >= one attempt of two uniforms */
		const bool nvalidc = (i + 1 < d.N) && (nbase + 2ull * (C - 1) + 2ull * (unsigned)nnvalid + 1024ull <= d.tape_len);
		const bool nvalidl = (nl.a0 != 0xff);
		const unsigned nrank = nl.rw + (unsigned)__popcll(__ballot(nvalidl) & ((1ull << lane) - 1ull));
		double xs[2 * C];
#pragma unroll
		for (int k = 0; k < 2 * C; k++) xs[k] = 0.5;
		if (nvalidc && nvalidl) {
			const double *tp = d.tape + nbase + 2ull * nrank;
#pragma unroll
			for (int k = 0; k < 2 * C; k++) xs[k] = tp[k];
		}
		fetch_rows(nl);
		/* ---- this individual's Z: a candidate drawn earlier (its counts left then), or the plain path ---- */
		const unsigned long long dc = offi - cbase;
		const bool hit = cvalid && offi >= cbase && !(dc & 1ull) && dc < 2ull * C;
		int z0 = 0xff, z1 = 0xff;
		if (hit) {
			const unsigned c = (unsigned)(dc >> 1);
			if (valid) {
				z0 = (int)((cz0 >> (4 * c)) & 0xfu);
				z1 = (int)((cz1 >> (4 * c)) & 0xfu);
			}
		} else {
			const unsigned rank = cl.rw + (unsigned)__popcll(__ballot(valid) & ((1ull << lane) - 1ull));
			bool amb0 = false, amb1 = false;
			if (valid && covered) {
				float qf[KMAX];
#pragma unroll
				for (int m = 0; m < KMAX; m++) qf[m] = (m < K) ? (float)readlane_f64(cl.qv, m) : 0.f;
				const double x0 = d.tape[offi + 2ull * rank], x1 = d.tape[offi + 2ull * rank + 1];
				z0 = bucket_f32<KMAX>((float)x0, cl.F0, qf, K, &amb0);
				z1 = bucket_f32<KMAX>((float)x1, cl.F1, qf, K, &amb1);
			}
			if (__ballot(valid && covered && (amb0 || amb1))) { /* rare: the draw in double */
				double cum[KMAX], q[KMAX];
#pragma unroll
				for (int m = 0; m < KMAX; m++) q[m] = (m < K) ? readlane_f64(cl.qv, m) : 0.0;
				if (valid && covered && amb0) {
					const double tot = weights<KMAX>(d.freq + ((size_t)j * d.Amax + cl.a0) * d.KP, q, cum, K);
					z0 = bucket_fast<KMAX>(d.tape[offi + 2ull * rank], cum, tot, K);
				}
				if (valid && covered && amb1) {
					const double tot = weights<KMAX>(d.freq + ((size_t)j * d.Amax + cl.a1) * d.KP, q, cum, K);
					z1 = bucket_fast<KMAX>(d.tape[offi + 2ull * rank + 1], cum, tot, K);
				}
			}
			if (!covered) { z0 = z1 = 0xff; }
			/* the plain path's counts: set C */
			unsigned long long v = (unsigned long long)tag << 48;
#pragma unroll
			for (int m = 0; m < KMAX; m++)
				if (m < K) {
					const int wc = __popcll(__ballot(z0 == m)) + __popcll(__ballot(z1 == m));
					if (lane == m / 3) v |= (unsigned long long)(wc & 0xffff) << (16 * (m % 3));
				}
			if (lane < W) { if (local) st_xcd(pipe_gran(pg, slot, NP, pub, C * W + lane), v); else st_agent(pipe_gran(pg, slot, NP, pub, C * W + lane), v); }
		}
		STAMP(i, 1);
		if (j < d.Lp) *(unsigned short *)(d.z + (size_t)i * rowb + (size_t)j * 2) = (unsigned short)(z0 | (z1 << 8));
		if (!covered && t == 0) cb->overflow_flag = 1;
		STAMP(i, 2);
		/* ---- the next individual's candidates ---- */
		fetch_geno(i + 2, ml);
		/* the next Dirichlet's stretch of the tape (its start is known up to this Dirichlet's rejected attempts: 160
		 * uniforms cover that), so that the control wave finds it in L2 */
		if (touch == -1.0) cb->overflow_flag = 2;
		if (t < 10 && nvalidc) touch = d.tape[nbase + 2ull * (unsigned)nnvalid + 16u * (unsigned)t];
		cbase = nbase;
		cvalid = nvalidc;
		cz0 = cz1 = 0;
		if (nvalidc) { /* wave-uniform */
			unsigned camb = 0;
			if (nvalidl) {
				float qf[KMAX], cum0[KMAX], cum1[KMAX];
#pragma unroll
				for (int m = 0; m < KMAX; m++) qf[m] = (m < K) ? (float)readlane_f64(nl.qv, m) : 0.f;
				const float run0 = prefix_f32<KMAX>(nl.F0, qf, K, cum0), run1 = prefix_f32<KMAX>(nl.F1, qf, K, cum1);
#pragma unroll
				for (int c = 0; c < C; c++) {
					bool a0, a1;
					const int b0 = bucket_from_cum<KMAX>((float)xs[2 * c], cum0, run0, K, &a0);
					const int b1 = bucket_from_cum<KMAX>((float)xs[2 * c + 1], cum1, run1, K, &a1);
					cz0 |= (unsigned)b0 << (4 * c);
					cz1 |= (unsigned)b1 << (4 * c);
					camb |= (a0 ? 1u : 0u) << c;
					camb |= (a1 ? 1u : 0u) << (8 + c);
				}
			}
			STAMP(i, 6);
			if (__ballot(camb != 0u)) { /* rare (about 4 % of the waves): the draws the filter could not decide, in double */
				double cum[KMAX], q[KMAX];
#pragma unroll
				for (int m = 0; m < KMAX; m++) q[m] = (m < K) ? readlane_f64(nl.qv, m) : 0.0;
				if (camb & 0xffu) {
					const double tot = weights<KMAX>(d.freq + ((size_t)j * d.Amax + nl.a0) * d.KP, q, cum, K);
#pragma unroll
					for (int c = 0; c < C; c++)
						if ((camb >> c) & 1u) cz0 = (cz0 & ~(0xfu << (4 * c))) | ((unsigned)bucket_fast<KMAX>(xs[2 * c], cum, tot, K) << (4 * c));
				}
				if (camb >> 8) {
					const double tot = weights<KMAX>(d.freq + ((size_t)j * d.Amax + nl.a1) * d.KP, q, cum, K);
#pragma unroll
					for (int c = 0; c < C; c++)
						if ((camb >> (8 + c)) & 1u) cz1 = (cz1 & ~(0xfu << (4 * c))) | ((unsigned)bucket_fast<KMAX>(xs[2 * c + 1], cum, tot, K) << (4 * c));
				}
			}
			/* the wave's counts of every candidate: lane c * W + w carries word w of candidate c.  A lane's two draws as
			 * 8-bit fields (one per cluster), summed over the wave with DPP adds (a field stays <= 128: no carries) --
			 * 2 K ballots and scalar popcounts per candidate took 5 k cycles here */
			unsigned mylo = 0, myhi = 0;
#pragma unroll
			for (int c = 0; c < C; c++) {
				unsigned long long P = 0;
				if (nvalidl) P = (1ull << (8 * ((cz0 >> (4 * c)) & 0xfu))) + (1ull << (8 * ((cz1 >> (4 * c)) & 0xfu)));
				const unsigned plo = wave_allsum_u32((unsigned)P), phi = (KMAX > 4) ? wave_allsum_u32((unsigned)(P >> 32)) : 0u;
				if (myc == c) {
					mylo = plo;
					myhi = phi;
				}
			}
			const unsigned long long my = ((unsigned long long)myhi << 32) | mylo;
			unsigned long long v = (unsigned long long)ntag << 48;
#pragma unroll
			for (int k = 0; k < 3; k++)
				if (3 * myw + k < 8) v |= ((my >> (8 * (3 * myw + k))) & 0xffull) << (16 * k);
			if (lane < C * W) { if (local) st_xcd(pipe_gran(pg, nslot, NP, pub, lane), v); else st_agent(pipe_gran(pg, nslot, NP, pub, lane), v); }
		}
		STAMP(i, 7);
		cl = nl;
		nl = ml;
	}
}

/* the uniforms at positions [0, n) after `base`, in stream order (8 per lane: one skip-ahead, then stepping) */
__global__ void __launch_bounds__(256) k_tape(const isg_wh_tables *tab, isg_wh base, unsigned long long n, double *tape)
{
	const unsigned long long p0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 8ull;
	if (p0 >= n) return;
	isg_wh s = isg_wh_jump(tab, base, p0);
	double v[8];
#pragma unroll
	for (int k = 0; k < 8; k++) v[k] = isg_wh_next(&s);
	if (p0 + 8 <= n) {
		double2 *o = (double2 *)(tape + p0);
#pragma unroll
		for (int k = 0; k < 4; k++) o[k] = make_double2(v[2 * k], v[2 * k + 1]);
	} else {
		for (int k = 0; k < 8 && p0 + k < n; k++) tape[p0 + k] = v[k];
	}
}

/* single precision copy of the frequency table for the Z-draw pre-filter */
__global__ void k_freqf(DevView d)
{
	const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t rows = (size_t)d.L * d.Amax;
	if (id >= rows * d.KPF) return;
	const size_t r = id / d.KPF;
	const int m = (int)(id - r * d.KPF);
	d.freqf[id] = (m < d.K) ? (float)d.freq[r * d.KP + m] : 0.f;
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
			c.tape = nullptr;
	double sum = 0;
	for (int a = 0; a < A; a++) {
		double g = isg_rgamma(&c, (double)d.cnt[((size_t)j * d.Amax + a) * d.K + k] + 1.0);
		d.freq[((size_t)j * d.Amax + a) * d.KP + k] = g;
		sum += g;
	}
	for (int a = 0; a < A; a++) d.freq[((size_t)j * d.Amax + a) * d.KP + k] /= sum;
}

/* ------------------------------------------------------------------------------------------ */
/* k_spop: update_S_POP (mcmc.c:913-983) -- one workgroup, K sequential MH steps                 */
/* ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ double dev_q_trans(int a, int b) /* mcmc.c:1566-1593 */
{
	if (a == 0) return (b == 0 || b == 1) ? 0.5 : 0.0;
	if (a == 2) return (b == 2 || b == 1) ? 0.5 : 0.0;
	if (a == 1) return (b == 0 || b == 2) ? 0.05 : (b == 1 ? 0.90 : 0.0);
	return 0.0;
}
/* proposal() (mcmc.c:1630-1648): sum over individuals of log(s_i^(g_i-1) (1-s_i)), order-independent sum */
template <int BLOCK>
__device__ __forceinline__ double dev_proposal(const DevView &d, const double *s, unsigned long long *smr)
{
	isg_acc a;
	isg_acc_zero(&a);
	for (int i = threadIdx.x; i < d.N; i += BLOCK) {
		double temp = 0;
		for (int j = 0; j < d.K; j++) temp += d.qq[(size_t)i * d.K + j] * s[j];
		isg_acc_add(&a, isg_log(isg_pow(temp, (double)(d.gen[i] - 1)) * (1 - temp)));
	}
	const isg_acc r = block_reduce_acc<BLOCK>(a, smr);
	return isg_acc_value(&r);
}
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_spop(DevView d, double *S, int *state, isg_wh start, int back_refl, uint64_t *used_out)
{
	__shared__ unsigned long long smr[(BLOCK / 64) * 5];
	__shared__ double Scur[ISG_KCAP], Stmp[ISG_KCAP];
	__shared__ int stc[ISG_KCAP], stt[ISG_KCAP];
	__shared__ double ld_sh;
	const int t = threadIdx.x, K = d.K;
	if (t < K) { Scur[t] = S[t]; stc[t] = state[t]; }
	__syncthreads();
	/* proposal(self_rates) only changes when a move is accepted: carried instead of recomputed */
	double cur_ld = dev_proposal<BLOCK>(d, Scur, smr);
	isg_cursor c;
	c.s = start;
	c.used = 0;
	c.tape = nullptr;
	for (int j = 0; j < K; j++) {
		if (t == 0) {
			for (int i = 0; i < K; i++) { Stmp[i] = Scur[i]; stt[i] = stc[i]; }
			if (back_refl == 1) { /* mcmc.c:939-945 */
				double v = isg_cur_next(&c) * 2 * 0.05 - 0.05;
				v += Scur[j];
				if (v <= 0.0) v = 0.0 - v;
				else if (v >= 1.0) v = 1.0 - (v - 1.0);
				Stmp[j] = v;
			} else { /* adpt_indp, mcmc.c:1461-1520 */
				const int st = stc[j];
				double v;
				int ns;
				if (st == 0) {
					if (isg_cur_next(&c) < 0.50) { v = 0.0; ns = 0; } else { v = isg_cur_next(&c); ns = 1; }
				} else if (st == 2) {
					if (isg_cur_next(&c) < 0.5) { v = 1.0; ns = 2; } else { v = isg_cur_next(&c); ns = 1; }
				} else {
					const double tt = isg_cur_next(&c);
					if (tt <= 0.05) { v = 0.0; ns = 0; }
					else if (tt >= 0.95) { v = 1.0; ns = 2; }
					else { v = isg_cur_next(&c); ns = 1; }
				}
				Stmp[j] = v;
				stt[j] = ns;
			}
		}
		__syncthreads();
		const double new_ld = dev_proposal<BLOCK>(d, Stmp, smr);
		if (t == 0) {
			double mh = isg_exp(new_ld - cur_ld);
			if (back_refl == 0) {
				double h = 1.0;
				for (int i = 0; i < K; i++) h *= dev_q_trans(stc[i], stt[i]) / dev_q_trans(stt[i], stc[i]);
				mh *= h;
			}
			const double thr = (1 > mh) ? mh : 1;
			double keep = cur_ld;
			if (isg_cur_next(&c) < thr) {
				Scur[j] = Stmp[j];
				if (back_refl == 0) stc[j] = stt[j];
				keep = new_ld;
			}
			ld_sh = keep;
		}
		__syncthreads();
		cur_ld = ld_sh;
	}
	if (t < K) { S[t] = Scur[t]; state[t] = stc[t]; }
	if (t == 0) *used_out = c.used;
}

/*
 * update_S_POP for -e 1 and K <= 6, spread over the chip.  The proposal of step j is a reflected +-0.05 step from the
 * cluster's OWN current rate (mcmc.c:939-945), and a step consumes exactly two uniforms: all K proposed values are
 * known before the first decision.  The vector step j evaluates proposal() at depends on which earlier steps were
 * accepted -- 2^j possibilities -- so all 2^K - 1 of them (+ the current vector) are evaluated at once, each an
 * order-independent exact sum over the individuals (same terms, same accumulator as dev_proposal), and k_spop_decide
 * then walks the K Metropolis steps in order on one lane.  Evaluation id: 0 = current rates; 2^j + h = step j with
 * acceptance history h (bit m = step m accepted).
 */
#define ISG_SPOP_TREE_K 6
__device__ __forceinline__ double spop_proposed(double cur, double u)
{
	double v = u * 2 * 0.05 - 0.05;
	v += cur;
	if (v <= 0.0) v = 0.0 - v;
	else if (v >= 1.0) v = 1.0 - (v - 1.0);
	return v;
}
__global__ void __launch_bounds__(256) k_spop_tree(DevView d, const double *S, isg_wh start, unsigned long long *limbs /* [2^K][5], zeroed */)
{
	__shared__ double Scur[ISG_SPOP_TREE_K], V[ISG_SPOP_TREE_K];
	const int K = d.K, t = threadIdx.x;
	if (t < K) {
		Scur[t] = S[t];
		isg_wh s = isg_wh_jump(d.tab, start, 2ull * (unsigned)t);
		V[t] = spop_proposed(Scur[t], isg_wh_next(&s));
	}
	__syncthreads();
	const int i = blockIdx.x * 256 + t;
	double q[ISG_SPOP_TREE_K];
	double gm1 = 0.0;
#pragma unroll
	for (int m = 0; m < ISG_SPOP_TREE_K; m++) q[m] = (i < d.N && m < K) ? d.qq[(size_t)i * K + m] : 0.0;
	if (i < d.N) gm1 = (double)(d.gen[i] - 1);
	{
		const int e = blockIdx.y; /* one evaluation per workgroup row */
		int j = -1, h = 0;
		if (e) {
			j = 31 - __builtin_clz((unsigned)e);
			h = e - (1 << j);
		}
		isg_acc a;
		isg_acc_zero(&a);
		if (i < d.N) {
			double temp = 0;
#pragma unroll
			for (int m = 0; m < ISG_SPOP_TREE_K; m++)
				if (m < K) temp += q[m] * ((m == j || (m < j && ((h >> m) & 1))) ? V[m] : Scur[m]);
			isg_acc_add(&a, isg_log(isg_pow(temp, gm1) * (1 - temp)));
		}
		AccLimbs sl = acc_to_limbs(a);
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			sl.l0 += __shfl_down(sl.l0, o, 64);
			sl.l1 += __shfl_down(sl.l1, o, 64);
			sl.l2 += __shfl_down(sl.l2, o, 64);
			sl.l3 += __shfl_down(sl.l3, o, 64);
			sl.flags |= __shfl_down(sl.flags, o, 64);
		}
		if (lane_id() == 0) { /* 32-bit limbs in 64-bit words: the chip-wide sums cannot carry out */
			atomicAdd(&limbs[e * 5 + 0], sl.l0);
			atomicAdd(&limbs[e * 5 + 1], sl.l1);
			atomicAdd(&limbs[e * 5 + 2], sl.l2);
			atomicAdd(&limbs[e * 5 + 3], sl.l3);
			if (sl.flags) atomicOr(&limbs[e * 5 + 4], (unsigned long long)sl.flags);
		}
	}
}
__global__ void k_spop_decide(DevView d, double *S, isg_wh start, const unsigned long long *limbs)
{
	if (threadIdx.x || blockIdx.x) return;
	const int K = d.K;
	auto value = [&](int e) {
		AccLimbs sl;
		sl.l0 = limbs[e * 5 + 0];
		sl.l1 = limbs[e * 5 + 1];
		sl.l2 = limbs[e * 5 + 2];
		sl.l3 = limbs[e * 5 + 3];
		sl.flags = (unsigned)limbs[e * 5 + 4];
		const isg_acc a = limbs_to_acc(sl);
		return isg_acc_value(&a);
	};
	double cur_ld = value(0);
	int h = 0;
	isg_cursor c;
	c.s = isg_wh_jump(d.tab, start, 0);
	c.used = 0;
	c.tape = nullptr;
	for (int j = 0; j < K; j++) {
		const double v = spop_proposed(S[j], isg_cur_next(&c)); /* the same value k_spop_tree formed */
		const double new_ld = value((1 << j) + h);
		const double mh = isg_exp(new_ld - cur_ld);
		const double thr = (1 > mh) ? mh : 1;
		if (isg_cur_next(&c) < thr) {
			S[j] = v;
			h |= 1 << j;
			cur_ld = new_ld;
		}
	}
}

/* update_alpha: the N*K factors pow(q, ralpha + n) / pow(q, n + alpha) (mcmc.c:1258); their ORDERED
 * product (overflow / NaN behaviour of the reference included) is formed on the host */
__global__ void k_alpha_ratios(DevView d, double ralpha, double alpha, double *ratios)
{
	const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= (size_t)d.N * d.K) return;
	const double q = d.qq[id], n = (double)d.qqnum[id];
	ratios[id] = isg_pow(q, ralpha + n) / isg_pow(q, n + alpha);
}

/* cal_lkh: totallkh = order-independent sum of indvlkh (mcmc.c:1940) */
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_lkh_total(DevView d, double *total)
{
	__shared__ unsigned long long smr[(BLOCK / 64) * 5];
	isg_acc a;
	isg_acc_zero(&a);
	for (int i = threadIdx.x; i < d.N; i += BLOCK) isg_acc_add(&a, d.indvlkh[i]);
	const isg_acc r = block_reduce_acc<BLOCK>(a, smr);
	if (threadIdx.x == 0) *total = isg_acc_value(&r);
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

/* the device error word, d_err[0] (bit 1: dt_stat saw a selfing rate outside [0,1], its value in d_err[2..3]); `e` was
 * copied back by the caller together with whatever else it was waiting for */
static int report_dev_err(const unsigned e[4])
{
	if (e[0] & 2u) {
		double v;
		char msg[160];
		memcpy(&v, e + 2, sizeof(v));
		snprintf(msg, sizeof(msg), "The value of selfing rate or inbreeding coefficient %f is beyond [0,1]!", v); /* mcmc.c:1540 */
		return fail(msg);
	}
	if (e[0]) return fail("device error flag set");
	return 0;
}

static int refresh_freqf(isg_ctx *c);
static void keyed_layout(isg_ctx *c)
{
	uint64_t N = c->cfg.N, L = c->cfg.L, P = c->cfg.P, K = c->cfg.K, A = c->Amax;
	uint64_t SP = 16 * A + 16, SZ = P * L + 16 * K + 16, ZI0 = 1 + 2 * N, B0 = ZI0 + N * SZ;
	uint64_t offS = K * L * SP, offG = offS + 4 * K, offZ = offG + 2 * N, offA = offZ + N * SZ, BLK = offA + 4;
	if (c->cfg.mode == 3) BLK += 2 * N; /* update_S_IND of individual i at offA + 4 + 2 i */
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

/*
 * Host side of update_P in the replay schedule (the K L Dirichlets are drawn in stream order): rdirich
 * (random.c:264-280) with the accept / reject test of rgamma2 (random.c:195-231) pre-decided in single precision,
 * exactly as rgamma2_try_dev does on the device -- the two logarithms only feed the comparison
 * c3 log(u1) - log(w) + w >= 1; outside a band of 2e-6 (1 + |terms|) the float value decides, inside it the double
 * expression.  Same values, same consumption as isg_rdirich.
 */
static inline double host_rgamma2_try(isg_cursor *c, double alpha)
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
		const float l1 = logf((float)u1), lw = logf((float)w);
		const double al1 = fabs((double)l1), alw = fabs((double)lw);
		const double dlt = (c3 * (double)l1 - (double)lw + w) - 1;
		const double tol = 2e-6 * (fabs(c3) * (1.0 + al1) + 1.0 + alw) + 1e-12 * fabs(w);
		bool rej;
		if (dlt > tol) rej = true;
		else if (dlt < -tol) rej = false;
		else rej = (c3 * isg_log(u1) - isg_log(w) + w) >= 1;
		if (rej) return -1;
	}
	return c1 * w;
}
/* rgamma2's constants only depend on the shape (random.c:199-203): the sequential loop has three divisions and a square root
 * less per attempt when they are formed beforehand -- for all gammas of the sweep at once, by a few threads (the shapes are
 * the counts + 1, known before the first draw).  Same expressions, same values. */
struct HostGammaCoef { double c1, c2, c3, c4, c5; };
static inline void host_gamma_coef(double alpha, HostGammaCoef *o)
{
	o->c1 = alpha - 1;
	o->c2 = (alpha - 1 / (6 * alpha)) / o->c1;
	o->c3 = 2 / o->c1;
	o->c4 = o->c3 + 2;
	o->c5 = 1 / isg_sqrt(alpha);
}
static void host_gamma_coefs(const double *shape, size_t n, HostGammaCoef *out)
{
	unsigned nt = std::thread::hardware_concurrency();
	nt = nt > 4 ? 4 : (nt < 1 ? 1 : nt);
	if (n < 16384) nt = 1;
	auto work = [&](size_t a, size_t b) { for (size_t g = a; g < b; g++) host_gamma_coef(shape[g], &out[g]); };
	std::vector<std::thread> th;
	const size_t per = (n + nt - 1) / nt;
	for (unsigned k = 1; k < nt; k++) th.emplace_back(work, k * per < n ? k * per : n, (k + 1) * per < n ? (k + 1) * per : n);
	work(0, per < n ? per : n);
	for (auto &t : th) t.join();
}
static inline double host_rgamma2_try_pre(isg_cursor *c, double alpha, const HostGammaCoef &k)
{
	double u1, u2, w;
	do {
		u1 = isg_cur_next(c);
		u2 = isg_cur_next(c);
		if (alpha > 2.5) u1 = u2 + k.c5 * (1 - 1.86 * u1);
	} while ((u1 >= 1) || (u1 <= 0));
	w = k.c2 * u2 / u1;
	if ((k.c3 * u1 + w + 1 / w) > k.c4) {
		const float l1 = logf((float)u1), lw = logf((float)w);
		const double al1 = fabs((double)l1), alw = fabs((double)lw);
		const double dlt = (k.c3 * (double)l1 - (double)lw + w) - 1;
		const double tol = 2e-6 * (fabs(k.c3) * (1.0 + al1) + 1.0 + alw) + 1e-12 * fabs(w);
		bool rej;
		if (dlt > tol) rej = true;
		else if (dlt < -tol) rej = false;
		else rej = (k.c3 * isg_log(u1) - isg_log(w) + w) >= 1;
		if (rej) return -1;
	}
	return k.c1 * w;
}
/* rdirich over shapes given directly (count + 1 already formed) with their constants */
static void host_rdirich_pre(isg_cursor *c, const double *shape, const HostGammaCoef *coef, int n, double *out)
{
	double sum = 0;
	for (int k = 0; k < n; k++) {
		const double a = shape[k];
		double g = 0;
		if (a > 1) {
			do { g = host_rgamma2_try_pre(c, a, coef[k]); } while (g < 0);
		} else {
			g = isg_rgamma(c, a);
		}
		out[k] = g;
		sum += g;
	}
	for (int k = 0; k < n; k++) out[k] /= sum;
}
static void host_rdirich(isg_cursor *c, const double *count, int n, double *out, double add)
{
	double sum = 0;
	for (int k = 0; k < n; k++) {
		const double a = count[k] + add;
		double g = 0;
		if (a > 1) {
			do { g = host_rgamma2_try(c, a); } while (g < 0);
		} else {
			g = isg_rgamma(c, a);
		}
		out[k] = g;
		sum += g;
	}
	for (int k = 0; k < n; k++) out[k] /= sum;
}

/*
 * The uniforms of the host's Dirichlet loop come from the device: half of that loop's time was the generator
 * (21 ns per uniform, ~3 per gamma of ~118 ns).  k_tape writes the next `need` uniforms of the stream, the loop reads
 * them through the cursor's tape; if they run out (they are budgeted at 3 per gamma) the loop continues with the
 * generator from the position reached -- same values either way.
 */
__global__ void __launch_bounds__(256) k_tape(const isg_wh_tables *tab, isg_wh base, unsigned long long n, double *tape);
static int host_tape_begin(isg_ctx *c, uint64_t ngamma, isg_cursor *cur)
{
	cur->s = c->rng;
	cur->used = 0;
	cur->tape = nullptr;
	c->htape_len = 0;
	if (!c->host_tape || ngamma < 4096) return 0;
	const uint64_t need = 3 * ngamma + 65536;
	if (need > c->tape_cap) {
		if (c->d_tape) HIPCHK(hipFree(c->d_tape));
		c->d_tape = nullptr;
		c->tape_cap = 0;
		HIPCHK(hipMalloc((void **)&c->d_tape, sizeof(double) * need));
		c->tape_cap = need;
	}
	if (c->htape_cap < need) {
		if (c->htape) (void)hipHostFree(c->htape);
		c->htape = nullptr;
		c->htape_cap = 0;
		HIPCHK(hipHostMalloc((void **)&c->htape, sizeof(double) * need, hipHostMallocDefault));
		c->htape_cap = need;
	}
	prof_begin(c);
	hipLaunchKernelGGL(k_tape, dim3((unsigned)((need + 2047) / 2048)), dim3(256), 0, c->stream, c->d.tab, c->rng, (unsigned long long)need, c->d_tape);
	prof_end(c, "k_tape_host");
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(c->htape, c->d_tape, sizeof(double) * need, hipMemcpyDeviceToHost, c->stream));
	c->htape_len = need;
	return 0;
}
/* The per-sweep host copies of replay update_P (counts down, frequencies up) go through vectors of fixed size: registered with the
 * runtime they are copied by DMA at PCIe rate instead of through a staging buffer.  Failure to register only costs that speed. */
static void pin_host(isg_ctx *c, void *ptr, size_t bytes)
{
	if (ptr && bytes && hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess) c->pinned.push_back(ptr);
	else (void)hipGetLastError();
}
/* replay update_P: marks / waits for the point of the stream where the counts' copy ends, so that the host can form the shapes and
 * their constants while the uniform tape is generated and copied */
#define HOST_T(c, k, t0) do { if ((c)->host_timing) { const auto t1_ = std::chrono::steady_clock::now(); (c)->host_t[k] += std::chrono::duration<double>(t1_ - (t0)).count(); (t0) = t1_; } } while (0)
static int counts_mark(isg_ctx *c)
{
	if (!c->ev_cnt) HIPCHK(hipEventCreateWithFlags(&c->ev_cnt, hipEventDisableTiming));
	HIPCHK(hipEventRecord(c->ev_cnt, c->stream));
	return 0;
}
static int counts_wait(isg_ctx *c)
{
	HIPCHK(hipEventSynchronize(c->ev_cnt));
	return 0;
}
/* call after the stream has been synchronised, before the loop */
static void host_tape_attach(isg_ctx *c, isg_cursor *cur) { if (c->htape_len) cur->tape = c->htape; }
/* before each Dirichlet of n gammas: leave the tape while a comfortable margin remains (32 attempts per gamma) */
static inline void host_tape_guard(isg_ctx *c, isg_cursor *cur, int n)
{
	if (cur->tape && (uint64_t)cur->used + 64ull * (unsigned)n + 64 > c->htape_len) {
		cur->s = isg_wh_jump(&c->tab_h, c->rng, cur->used);
		cur->tape = nullptr;
	}
}
static void host_tape_end(isg_ctx *c, isg_cursor *cur)
{
	c->rng = cur->tape ? isg_wh_jump(&c->tab_h, c->rng, cur->used) : cur->s;
	c->raw_valid = false;
}

extern "C" const char *isg_last_error(void) { return g_err.c_str(); }

/* The cooperative replay kernels hand data between workgroups by polling, so every workgroup of the launch must be
 * resident at the same time: `blocks` of `threads` threads have to fit on the device's CUs at this kernel's occupancy. */
template <class Kern>
static bool fits_resident(Kern kern, int threads, long blocks, int device)
{
	int per_cu = 0, cus = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, 0) != hipSuccess) return false;
	if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return false;
	return (long)per_cu * cus >= blocks;
}
/* contexts alive per device in THIS process: the one-launch block resolver wants the whole chip to itself (all its workgroups
 * resident); with several chains on one GPU the per-block launches interleave much better (8 chains at config 3: 72 vs 179
 * chain-iterations/s), so it is only used by a device's sole context */
static std::atomic<int> g_live_ctx[64];
static void ctx_count(isg_ctx *c, int delta);
#include "isg_resolve_hip.inc"
#include "isg_walk_hip.inc"
#include "isg_spec_hip.inc"
static void ctx_count(isg_ctx *c, int delta)
{
	if (c->cfg.device < 0) return;
	if (delta > 0 && !c->counted) { g_live_ctx[c->cfg.device & 63].fetch_add(1); c->counted = true; }
	if (delta < 0 && c->counted) { g_live_ctx[c->cfg.device & 63].fetch_sub(1); c->counted = false; }
}

/* ploidy 4 (isg_poly_hip.inc, included further down) */
static int poly_ctx_create(const isg_config *cfg, const int32_t *allelenum, const int32_t *seq, isg_ctx **out);
static void poly_ctx_destroy(isg_ctx *c);
static int poly_update_P(isg_ctx *c);
static int poly_update_S_POP(isg_ctx *c);
static int poly_update_ZQ(isg_ctx *c, int init_flag);
static int poly_cal_lkh(isg_ctx *c);
static int poly_count_alleles(isg_ctx *c, int32_t *counts);
extern "C" void isg_ctx_destroy(isg_ctx *c);
static int noadm_update_Z(isg_ctx *c, int init_flag);
static int noadm_cal_lkh(isg_ctx *c);
static int indiv_update_S_IND(isg_ctx *c);
static int indiv_update_F_IND(isg_ctx *c);
static int indiv_cal_lkh_F(isg_ctx *c);
static int inbreed_update_F_POP(isg_ctx *c);
static int inbreed_cal_lkh(isg_ctx *c);
static int inbreed_alloc(isg_ctx *c);
static void inbreed_free(isg_ctx *c);
static void store_free(isg_ctx *c);
#define NOT_POLY(c, what) if ((c)->poly) return fail(what ": not part of the ploidy 4 chain (poly_geno.c:98-116)")

/* a context under construction: destroyed (device allocations, stream and all) unless construction reaches its end */
struct CtxGuard {
	isg_ctx *c;
	explicit CtxGuard(isg_ctx *p) : c(p) {}
	~CtxGuard() { if (c) isg_ctx_destroy(c); }
	void release() { c = nullptr; }
};
extern "C" int isg_ctx_create(const isg_config *cfg, const int32_t *allelenum, const int32_t *geno, const int32_t *missindx, isg_ctx **out)
{
	*out = nullptr;
	if (cfg->P == 4) return fail("isg_ctx_create: ploidy 4 data goes through isg_ctx_create_poly");
	if (cfg->P != 2) return fail("isg_ctx_create: only ploidy 2 and 4 are supported");
	if (cfg->K < 1 || cfg->K > ISG_KCAP) return fail("isg_ctx_create: K must be in 1..32");
	if (cfg->mode < 0 || cfg->mode > 5) return fail("isg_ctx_create: mode must be 0 .. 5");
	if (cfg->N < 1 || cfg->L < 1) return fail("isg_ctx_create: empty problem");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("isg_ctx_create: no HIP device available (the MI355X path has no CPU fallback)");
	if (cfg->device < 0 || cfg->device >= ndev) return fail("isg_ctx_create: bad device ordinal");
	HIPCHK(hipSetDevice(cfg->device));
	isg_ctx *c = new isg_ctx(); /* value-initialised: every pointer starts null */
	c->cfg = *cfg;
	ctx_count(c, +1);
	{ const char *e_ = getenv("INSTRUCT_HOST_TIMING"); c->host_timing = e_ && atoi(e_) == 1; }
	memset(&c->d, 0, sizeof(c->d));
	CtxGuard guard(c); /* any early return below releases what has been allocated so far */
	const int N = cfg->N, L = cfg->L, K = cfg->K;
	int Amax = 0;
	for (int j = 0; j < L; j++) Amax = allelenum[j] > Amax ? allelenum[j] : Amax;
	if (Amax > 254) return fail("isg_ctx_create: more than 254 alleles at a locus");
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
			if (a0 < 0 || a1 < 0 || a0 >= allelenum[j] || a1 >= allelenum[j]) return fail("isg_ctx_create: allele code out of range at a non-missing locus");
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
	d.KPF = (K + 3) & ~3;
	DALLOC(d.freqf, float, (size_t)Lp * Amax * d.KPF);
	d.lftab = d.lltab = nullptr;
	d.lli = nullptr;
	d.lli_F = 0;
	if (cfg->type_freq == 1 || cfg->mode == 0) { /* (-y 0 mixes the frequencies with the individual's qq: no tables) */
		const char *e = getenv("INSTRUCT_LL_TABLES");
		if (!(e && atoi(e) == 0)) {
			DALLOC(d.lftab, double, (size_t)L * Amax * K);
			const size_t ent = (size_t)50 * L * Amax * Amax * K;
			const char *ei = getenv("INSTRUCT_LL_INT");
			const size_t enti = 1 + (size_t)50 * Lp * Amax * Amax * K + (size_t)Lp * Amax * K; /* the integer form, locus innermost (k_loglik_int) */
			if (cfg->mode == 2 && !(ei && atoi(ei) == 0) && enti * sizeof(int2) <= ((size_t)1 << 30)) {
				DALLOC(d.lli, int2, enti);
				d.lli_F = (unsigned)(1 + (size_t)50 * Lp * Amax * Amax * K);
			} else if (cfg->mode == 2 && ent * sizeof(double) <= ((size_t)1 << 30)) { DALLOC(d.lltab, double, ent); } /* (mode 3: unclamped initial generations) */
			if (cfg->mode == 4) { /* one slot: log genofreq_inbreedcoff.  Mode 4 has no table-free path: up to half of what the device has free */
				size_t fr = 0, tot = 0;
				if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = (size_t)1 << 31;
				if ((ent / 50) * sizeof(double) <= fr / 2) { DALLOC(d.lltab, double, ent / 50); }
			}
		}
	}
	c->d_tape = nullptr;
	c->tape_cap = 0;
	c->nvalid_total = 0;
	for (int i = 0; i < N; i++) c->nvalid_total += (uint64_t)nvalid[i];
	if (resolve_alloc(c, nvalid)) return 1;
	if (cfg->rng_sched == ISG_SCHED_REPLAY && pdev_create(c, &c->pdev, L, K, Amax, allelenum, 1)) return 1;
	if (cfg->rng_sched == ISG_SCHED_REPLAY && cfg->mode != 0 && spec_create(c, &c->zspec, nvalid, 2)) return 1;
	{
		const int nwv = (Lp + 63) / 64 + 1;
		std::vector<unsigned> rw((size_t)N * nwv, 0);
		for (int i = 0; i < N; i++) {
			unsigned run = 0;
			for (int w = 0; w < nwv; w++) {
				rw[(size_t)i * nwv + w] = run;
				for (int j = 64 * w; j < 64 * (w + 1) && j < Lp; j++) run += (pk[((size_t)i * Lp + j) * 2] != 0xff) ? 1u : 0u;
			}
		}
		unsigned *drw;
		DALLOC(drw, unsigned, (size_t)N * nwv);
		HIPCHK(hipMemcpy(drw, rw.data(), sizeof(unsigned) * rw.size(), hipMemcpyHostToDevice));
		d.rankwave = drw;
		d.nwv = nwv;
	}
	{
		CoopBuf *cbp;
		DALLOC(cbp, CoopBuf, 1);
		c->d_coop = cbp;
		const char *e = getenv("INSTRUCT_ZQ_COOP");
		c->coop = (e && atoi(e) == 0) ? 0 : 1;
		e = getenv("INSTRUCT_ZQ_SPEC");
		c->spec = (e && atoi(e) == 0) ? 0 : 1;
		e = getenv("INSTRUCT_ZQ_XCD"); /* experimental, off by default: measured gain at config 3 is within noise */
		c->xcd = (e && atoi(e) == 1) ? 1 : 0;
		e = getenv("INSTRUCT_ZQ_PIPE");
		c->pipe = (e && atoi(e) == 0) ? 0 : 1;
		e = getenv("INSTRUCT_HOST_TAPE");
		c->host_tape = (e && atoi(e) == 0) ? 0 : 1;
		e = getenv("INSTRUCT_SPOP_TREE");
		c->spop_tree = (e && atoi(e) == 0) ? 0 : 1;
		e = getenv("INSTRUCT_ZQ_PIPE_XCD");
		c->pipe_xcd = (e && atoi(e) == 0) ? 0 : 1;
		e = getenv("INSTRUCT_ZQ_TEST_ABORT");
		c->test_abort = e ? atoi(e) : 0;
	}
	DALLOC(d.cnt, int, (size_t)Lp * Amax * K);
	DALLOC(d.qq, double, (size_t)N * K);
	DALLOC(c->d_qqsave, double, (size_t)N * K);
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
	DALLOC(c->d_err, unsigned, 4);
	DALLOC(c->d_S, double, ((cfg->mode == 3 || cfg->mode == 5) && N > ISG_KCAP) ? N : ISG_KCAP); /* modes 3, 5: one value per individual */
	c->d_Fprop = nullptr;
	if (cfg->mode == 5) { DALLOC(c->d_Fprop, double, N); }
	DALLOC(c->d_state, int, ISG_KCAP);
	DALLOC(c->d_ratios, double, (size_t)N * K);
	DALLOC(c->d_total, double, 1);
	c->ratios_h.assign((size_t)N * K, 0.0);
	c->h_qq = c->h_gen = c->h_S = c->h_lkh = true;
	d.err = c->d_err;
#undef DALLOC
	c->freq.assign((size_t)K * L * Amax, 0.0);
	c->freq_stage.assign((size_t)Lp * Amax * KP, 0.0);
	c->qq.assign((size_t)N * K, 0.0);
	c->qqnum.assign((size_t)N * K, 0);
	c->gen.assign(N, 0);
	c->S.assign((cfg->mode == 3 || cfg->mode == 5) ? N : K, 0.0); /* modes 3, 5: one value per individual */
	c->state.assign(K, 0);
	c->indvlkh.assign(N, 0.0);
	c->cnt_h.assign((size_t)L * Amax * K, 0);
	pin_host(c, c->cnt_h.data(), sizeof(int) * c->cnt_h.size());
	pin_host(c, c->freq_stage.data(), sizeof(double) * c->freq_stage.size());
	c->alpha = 0;
	c->totallkh = 0;
	c->iter = 0;
	c->rng.s1 = 13; c->rng.s2 = 4; c->rng.s3 = 1972; /* random.c:10-12 */
	c->origin = c->rng;
	c->raw_seed[0] = 13; c->raw_seed[1] = 4; c->raw_seed[2] = 1972;
	c->raw_valid = true;
	c->prof = false;
	keyed_layout(c);
	if (cfg->mode == 0 && !d.lftab) return fail("isg_ctx_create: mode 0 needs its log frequency table (INSTRUCT_LL_TABLES must not be 0)");
	if (cfg->mode == 4) {
		if (!d.lltab) return fail("isg_ctx_create: mode 4 needs its log-likelihood table (INSTRUCT_LL_TABLES must not be 0; L * Amax^2 * K doubles, at most half of the free device memory)");
		if (inbreed_alloc(c)) return 1;
	}
	guard.release();
	*out = c;
	return 0;
}

extern "C" void isg_ctx_destroy(isg_ctx *c)
{
	if (!c) return;
	ctx_count(c, -1);
	if (c->host_timing && c->host_n > 0)
		fprintf(stderr, "replay update_P host side, ms per sweep over %ld sweeps: counts %.3f  shapes %.3f  constants %.3f  tape %.3f  draws %.3f  upload %.3f\n", c->host_n,
			1e3 * c->host_t[0] / c->host_n, 1e3 * c->host_t[1] / c->host_n, 1e3 * c->host_t[2] / c->host_n, 1e3 * c->host_t[3] / c->host_n, 1e3 * c->host_t[4] / c->host_n, 1e3 * c->host_t[5] / c->host_n);
	(void)hipSetDevice(c->cfg.device);
	(void)hipStreamSynchronize(c->stream);
	if (c->htape) (void)hipHostFree(c->htape);
	c->htape = nullptr;
	if (c->ev_cnt) (void)hipEventDestroy(c->ev_cnt);
	c->ev_cnt = nullptr;
	if (c->ev_tape) (void)hipEventDestroy(c->ev_tape);
	c->ev_tape = nullptr;
	for (void *q : c->pinned) (void)hipHostUnregister(q);
	c->pinned.clear();
	store_free(c);
	resolve_free(c);
	pdev_free(c->pdev);
	c->pdev = nullptr;
	spec_free(c->zspec);
	c->zspec = nullptr;
	if (c->poly) {
		poly_ctx_destroy(c);
		prof_collect(c);
		for (auto e : c->prof_free) (void)hipEventDestroy(e);
		(void)hipStreamDestroy(c->stream);
		delete c;
		return;
	}
	inbreed_free(c);
	DevView &d = c->d;
	(void)hipFree((void *)d.geno); (void)hipFree(d.z); (void)hipFree((void *)d.allelenum); (void)hipFree((void *)d.nvalid); (void)hipFree(d.freq); (void)hipFree(d.freqf); (void)hipFree(d.lftab); (void)hipFree(d.lltab); (void)hipFree(d.lli); (void)hipFree(c->d_tape); (void)hipFree((void *)d.rankwave); (void)hipFree(c->d_coop); (void)hipFree(c->d_pipe); (void)hipFree(c->d_spop); (void)hipFree(d.cnt);
	(void)hipFree(d.qq); (void)hipFree(c->d_qqsave); (void)hipFree(d.qqnum); (void)hipFree(d.gen); (void)hipFree(d.genprop); (void)hipFree(d.uacc); (void)hipFree(d.indvlkh);
	(void)hipFree((void *)d.tab); (void)hipFree(c->d_pos); (void)hipFree(c->d_err); (void)hipFree(c->d_S); (void)hipFree(c->d_Fprop); (void)hipFree(c->d_state); (void)hipFree(c->d_ratios); (void)hipFree(c->d_total);
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
	return refresh_freqf(c);
}
static int refresh_freqf(isg_ctx *c)
{
	DevView &d = c->d;
	size_t n = (size_t)d.L * d.Amax * d.KPF;
	prof_begin(c);
	hipLaunchKernelGGL(k_freqf, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d);
	prof_end(c, "k_freqf");
	if (d.lli) { /* mode 2: the integer tables are all the likelihood sweeps read */
		const size_t per = (size_t)d.L * d.Amax * d.Amax * d.K;
		prof_begin(c);
		hipLaunchKernelGGL(k_lltab_int, dim3((unsigned)((per + 127) / 128)), dim3(128), 0, c->stream, d);
		prof_end(c, "k_lltab");
	} else if (d.lftab) {
		const size_t per = (size_t)d.L * d.Amax * d.Amax * d.K;
		prof_begin(c);
		hipLaunchKernelGGL(k_lltab, dim3((unsigned)((per + 127) / 128)), dim3(128), 0, c->stream, d);
		prof_end(c, "k_lltab");
	}
	HIPCHK(hipGetLastError());
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
/* host mirrors are refreshed on demand only (getters, the host-side steps of the replay schedule) */
static int ensure_qq(isg_ctx *c)
{
	if (c->h_qq) return 0;
	HIPCHK(hipMemcpyAsync(c->qq.data(), c->d.qq, sizeof(double) * c->qq.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(c->qqnum.data(), c->d.qqnum, sizeof(int) * c->qqnum.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	c->h_qq = true;
	return 0;
}
static int ensure_gen(isg_ctx *c)
{
	if (c->h_gen) return 0;
	HIPCHK(hipMemcpyAsync(c->gen.data(), c->d.gen, sizeof(int) * c->gen.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	c->h_gen = true;
	return 0;
}
static int ensure_S(isg_ctx *c)
{
	if (c->h_S) return 0;
	HIPCHK(hipMemcpyAsync(c->S.data(), c->d_S, sizeof(double) * c->S.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(c->state.data(), c->d_state, sizeof(int) * c->cfg.K, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	c->h_S = true;
	return 0;
}
static int ensure_lkh(isg_ctx *c)
{
	if (c->h_lkh) return 0;
	unsigned e[4] = {0, 0, 0, 0};
	HIPCHK(hipMemcpyAsync(c->indvlkh.data(), c->d.indvlkh, sizeof(double) * c->indvlkh.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(&c->totallkh, c->d_total, sizeof(double), hipMemcpyDeviceToHost, c->stream));
	if (!c->poly) HIPCHK(hipMemcpyAsync(e, c->d_err, sizeof(e), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (report_dev_err(e)) return 1;
	c->h_lkh = true;
	return 0;
}

/* ---- counts ---- */
template <int BLOCK, int LPT>
static int launch_count_tile(isg_ctx *c)
{
	DevView &d = c->d;
	const size_t lds = (size_t)LPT * d.Amax * d.K * BLOCK * sizeof(unsigned);
	int tiles = (d.Lp + BLOCK * LPT - 1) / (BLOCK * LPT);
	int rb = (1024 + tiles - 1) / tiles; /* about a thousand workgroups */
	if (rb > d.N) rb = d.N;
	if (rb < 1) rb = 1;
	int rows = (d.N + rb - 1) / rb;
	rb = (d.N + rows - 1) / rows;
	HIPCHK(hipFuncSetAttribute((const void *)k_count<BLOCK, LPT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	prof_begin(c);
	hipLaunchKernelGGL((k_count<BLOCK, LPT>), dim3(tiles, rb), dim3(BLOCK), lds, c->stream, d, rows);
	prof_end(c, "k_count");
	return 0;
}
static int launch_count(isg_ctx *c)
{
	DevView &d = c->d;
	HIPCHK(hipMemsetAsync(d.cnt, 0, sizeof(int) * (size_t)d.L * d.Amax * d.K, c->stream));
	const size_t per_lane = (size_t)d.Amax * d.K * sizeof(unsigned); /* LDS bytes per lane and locus */
	const size_t cap = 150 * 1024;
	if (4 * per_lane * 256 <= cap) { if (launch_count_tile<256, 4>(c)) return 1; }
	else if (per_lane * 256 <= cap) { if (launch_count_tile<256, 1>(c)) return 1; }
	else if (per_lane * 64 <= cap) { if (launch_count_tile<64, 1>(c)) return 1; }
	else {
		const size_t n = (size_t)d.N * d.Lp;
		prof_begin(c);
		hipLaunchKernelGGL(k_count_atomic, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d);
		prof_end(c, "k_count_atomic");
	}
	HIPCHK(hipGetLastError());
	return 0;
}

extern "C" int isg_count_alleles(isg_ctx *c, int32_t *counts)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (c->poly) return poly_count_alleles(c, counts);
	if (launch_count(c)) return 1;
	const int L = c->cfg.L, K = c->cfg.K, A = c->Amax;
	HIPCHK(hipMemcpyAsync(c->cnt_h.data(), c->d.cnt, sizeof(int) * c->cnt_h.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (int k = 0; k < K; k++)
		for (int j = 0; j < L; j++)
			for (int a = 0; a < A; a++) counts[((size_t)k * L + j) * A + a] = c->cnt_h[((size_t)j * A + a) * K + k];
	return 0;
}

/* End of update_alpha (replay schedule): the stream position of the next update_P is known and Z is final, while the likelihood
 * sweep of cal_lkh is still to run on the device.  The next update_P's counts and uniform tape are requested NOW, ahead of cal_lkh
 * in the stream, so that its host loop runs while cal_lkh does (0.3 ms at config 3) instead of waiting for it.  update_P uses them
 * only if it starts at exactly this position and nothing has written Z in between. */
static int update_P_ahead(isg_ctx *c)
{
	c->ahead_valid = false;
	if (c->poly || is_keyed(c) || !c->host_tape) return 0;
	if (c->pdev && c->pdev->usable) return 0; /* update_P runs on the device: nothing for the host to get ahead with */
	const int L = c->cfg.L, K = c->cfg.K;
	if (launch_count(c)) return 1;
	HIPCHK(hipMemcpyAsync(c->cnt_h.data(), c->d.cnt, sizeof(int) * c->cnt_h.size(), hipMemcpyDeviceToHost, c->stream));
	if (counts_mark(c)) return 1;
	uint64_t ngamma = 0;
	for (int j = 0; j < L; j++) ngamma += (c->allelenum[j] > 1) ? (uint64_t)c->allelenum[j] * K : 0;
	isg_cursor cur;
	if (host_tape_begin(c, ngamma, &cur)) return 1;
	if (!c->ev_tape) HIPCHK(hipEventCreateWithFlags(&c->ev_tape, hipEventDisableTiming));
	HIPCHK(hipEventRecord(c->ev_tape, c->stream));
	c->ahead_rng = c->rng;
	c->ahead_ngamma = ngamma;
	c->ahead_valid = true;
	return 0;
}

/* ---- update_P ---- */
extern "C" int isg_update_P(isg_ctx *c)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (c->poly) return poly_update_P(c);
	DevView &d = c->d;
	const int L = c->cfg.L, K = c->cfg.K, A = c->Amax;
	/* requested at the end of the previous iteration's update_alpha (update_P_ahead), for exactly this stream position? */
	const bool ahead = c->ahead_valid && !is_keyed(c) && c->ahead_rng.s1 == c->rng.s1 && c->ahead_rng.s2 == c->rng.s2 && c->ahead_rng.s3 == c->rng.s3;
	c->ahead_valid = false;
	if (!ahead && launch_count(c)) return 1;
	if (is_keyed(c)) {
		int n = K * L, B = 256;
		prof_begin(c);
		hipLaunchKernelGGL(k_pdirich, dim3((n + B - 1) / B), dim3(B), 0, c->stream, d, c->origin, iter_base(c), c->ky[KY_SP]);
		prof_end(c, "k_pdirich");
		HIPCHK(hipGetLastError());
		return refresh_freqf(c);
	}
	/* replay: the K*L Dirichlets consume the stream in (k, j) order with data-dependent length (random.c:167-250).  Their start
	 * positions are resolved on the device (isg_walk_hip.inc) and all of them drawn at once; if that cannot be done (a window
	 * missed, more alleles at a locus than the engine's groups hold) they are drawn sequentially on the host from the counts */
	if (!ahead && c->pdev && c->pdev->usable) {
		bool done = false;
		if (pdev_update_P(c, c->pdev, &done)) return 1;
		if (done) return refresh_freqf(c);
	}
	auto ht0 = std::chrono::steady_clock::now();
	isg_cursor cur;
	uint64_t ngamma = 0;
	if (ahead) {
		ngamma = c->ahead_ngamma;
		cur.s = c->rng;
		cur.used = 0;
		cur.tape = nullptr;
	} else {
		HIPCHK(hipMemcpyAsync(c->cnt_h.data(), d.cnt, sizeof(int) * c->cnt_h.size(), hipMemcpyDeviceToHost, c->stream));
		if (counts_mark(c)) return 1;
		for (int j = 0; j < L; j++) ngamma += (c->allelenum[j] > 1) ? (uint64_t)c->allelenum[j] * K : 0;
		if (host_tape_begin(c, ngamma, &cur)) return 1;
	}
	if (counts_wait(c)) return 1;
	HOST_T(c, 0, ht0); /* launches + wait for the counts */
	/* the shapes (count + 1.0, rdirich's `add`) of all gammas in stream order and their constants, then the draws */
	c->pshape.resize(ngamma);
	c->pcoef.resize(ngamma);
	{
		size_t g = 0;
		for (int k = 0; k < K; k++)
			for (int j = 0; j < L; j++) {
				const int Aj = c->allelenum[j];
				if (Aj <= 1) continue;
				for (int a = 0; a < Aj; a++) c->pshape[g++] = (double)c->cnt_h[((size_t)j * A + a) * K + k] + 1.0;
			}
	}
	HOST_T(c, 1, ht0); /* shapes */
	host_gamma_coefs(c->pshape.data(), (size_t)ngamma, (HostGammaCoef *)c->pcoef.data());
	HOST_T(c, 2, ht0); /* constants */
	if (ahead) HIPCHK(hipEventSynchronize(c->ev_tape)); /* (not the stream: the previous iteration's cal_lkh may still be running) */
	else HIPCHK(hipStreamSynchronize(c->stream)); /* the tape */
	HOST_T(c, 3, ht0); /* wait for the tape */
	host_tape_attach(c, &cur);
	{
		size_t g = 0;
		for (int k = 0; k < K; k++)
			for (int j = 0; j < L; j++) {
				const int Aj = c->allelenum[j];
				if (Aj <= 1) continue;
				host_tape_guard(c, &cur, Aj);
				host_rdirich_pre(&cur, &c->pshape[g], (const HostGammaCoef *)c->pcoef.data() + g, Aj, &c->freq[((size_t)k * L + j) * A]);
				g += (size_t)Aj;
			}
	}
	host_tape_end(c, &cur);
	HOST_T(c, 4, ht0); /* the sequential draws */
	const int rc_up = upload_freq(c);
	HOST_T(c, 5, ht0); /* transposition + upload launch */
	c->host_n++;
	return rc_up;
}

/* ---- update_S_POP ---- */
extern "C" int isg_update_S_POP(isg_ctx *c)
{
	if (c->cfg.mode == 4 && !c->poly) {
		HIPCHK(hipSetDevice(c->cfg.device));
		return inbreed_update_F_POP(c); /* the population coefficients of mode 4 (update_inbreedcoff_POP) */
	}
	if (c->cfg.mode != 2) return 0;
	HIPCHK(hipSetDevice(c->cfg.device));
	if (c->poly) return poly_update_S_POP(c);
	const int K = c->cfg.K;
	isg_wh start = is_keyed(c) ? isg_wh_jump(&c->tab_h, c->origin, iter_base(c) + c->ky[KY_OFFS]) : c->rng;
	if (c->cfg.back_refl == 1 && K <= ISG_SPOP_TREE_K && c->spop_tree) {
		const size_t nb = sizeof(unsigned long long) * 5 * ((size_t)1 << K);
		if (!c->d_spop) HIPCHK(hipMalloc((void **)&c->d_spop, sizeof(unsigned long long) * 5 * ((size_t)1 << ISG_SPOP_TREE_K)));
		HIPCHK(hipMemsetAsync(c->d_spop, 0, nb, c->stream));
		prof_begin(c);
		hipLaunchKernelGGL(k_spop_tree, dim3((c->cfg.N + 255) / 256, 1u << K), dim3(256), 0, c->stream, c->d, (const double *)c->d_S, start, c->d_spop);
		hipLaunchKernelGGL(k_spop_decide, dim3(1), dim3(64), 0, c->stream, c->d, c->d_S, start, (const unsigned long long *)c->d_spop);
		prof_end(c, "k_spop");
	} else {
		prof_begin(c);
		hipLaunchKernelGGL(k_spop<1024>, dim3(1), dim3(1024), 0, c->stream, c->d, c->d_S, c->d_state, start, c->cfg.back_refl, c->d_pos + 1);
		prof_end(c, "k_spop");
	}
	HIPCHK(hipGetLastError());
	c->h_S = false;
	if (!is_keyed(c)) {
		uint64_t used = 2ull * (uint64_t)K; /* -e 1: one proposal + one acceptance uniform per cluster */
		if (c->cfg.back_refl == 0) {
			HIPCHK(hipMemcpyAsync(&used, c->d_pos + 1, sizeof(used), hipMemcpyDeviceToHost, c->stream));
			HIPCHK(hipStreamSynchronize(c->stream));
		}
		host_advance(c, used);
	}
	return 0;
}

/* ---- update_G ---- */
extern "C" int isg_update_G(isg_ctx *c)
{
	NOT_POLY(c, "isg_update_G");
	if (c->cfg.mode != 2 && c->cfg.mode != 3) return 0;
	HIPCHK(hipSetDevice(c->cfg.device));
	DevView &d = c->d;
	double *d_S = c->d_S;
	isg_wh base = is_keyed(c) ? isg_wh_jump(&c->tab_h, c->origin, iter_base(c) + c->ky[KY_OFFG]) : c->rng;
	prof_begin(c);
	hipLaunchKernelGGL(k_gprop<1024>, dim3(1), dim3(1024), 0, c->stream, d, (const double *)d_S, base, is_keyed(c) ? 1 : 0, c->d_pos);
	prof_end(c, "k_gprop");
	prof_begin(c);
	if (d.lli && d.mode == 2) hipLaunchKernelGGL((k_loglik_int<256, true>), dim3(d.N), dim3(256), 0, c->stream, d);
	else if (d.lltab) hipLaunchKernelGGL((k_loglik_tab<256, true>), dim3(d.N), dim3(256), 0, c->stream, d);
	else hipLaunchKernelGGL((k_loglik<256, true>), dim3(d.N), dim3(256), 0, c->stream, d);
	prof_end(c, "k_loglik_pair");
	HIPCHK(hipGetLastError());
	c->h_gen = false;
	if (!is_keyed(c)) {
		uint64_t used = 0;
		unsigned e[4] = {0, 0, 0, 0};
		HIPCHK(hipMemcpyAsync(&used, c->d_pos, sizeof(used), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipMemcpyAsync(e, c->d_err, sizeof(e), hipMemcpyDeviceToHost, c->stream)); /* same wait: dt_stat out of range (mcmc.c:1524-1546) */
		HIPCHK(hipStreamSynchronize(c->stream));
		if (report_dev_err(e)) return 1;
		host_advance(c, used);
	}
	return 0; /* keyed schedule: no host wait here; the flag is read with the next likelihood download (ensure_lkh) */
}

/* ---- update_ZQ ---- */
/* A cooperative sweep that did not complete (a hand-off timed out because the GPU is shared and a workgroup was not
 * scheduled, or an individual's Dirichlet ran past the uniform tape) has overwritten part of Z, qq and qqnum.  Z and qqnum
 * are outputs only; qq is restored from the copy taken before the launch and the sweep is redone from the same stream
 * position by the single-workgroup kernel, which has no hand-offs and no tape budget -- same results, bit for bit. */
static bool coop_sweep_failed(isg_ctx *c, const unsigned flags[2])
{
	bool bad = flags[0] || flags[1];
	if (c->test_abort > 0 && --c->test_abort == 0) bad = true;
	return bad;
}
template <int KMAX>
static void launch_zq(isg_ctx *c, bool chain, isg_wh base, uint64_t pos0, uint64_t stride, int init_flag)
{
	DevView &d = c->d;
	if (chain)
		hipLaunchKernelGGL((k_zq<512, KMAX, true>), dim3(1), dim3(512), 0, c->stream, d, base, pos0, stride, init_flag, c->alpha, c->d_pos);
	else
		hipLaunchKernelGGL((k_zq<256, KMAX, false>), dim3(d.N), dim3(256), 0, c->stream, d, base, pos0, stride, init_flag, c->alpha, c->d_pos);
}
extern "C" int isg_update_ZQ(isg_ctx *c, int init_flag)
{
	c->ahead_valid = false; /* Z changes: counts requested ahead are stale */
	HIPCHK(hipSetDevice(c->cfg.device));
	if (c->poly) return poly_update_ZQ(c, init_flag);
	const int K = c->cfg.K;
	bool chain = !is_keyed(c);
	isg_wh base = chain ? c->rng : c->origin;
	uint64_t pos0 = chain ? 0 : (init_flag ? c->ky[KY_ZI0] : iter_base(c) + c->ky[KY_OFFZ]);
	uint64_t stride = c->ky[KY_SZ];
	c->d.tape = nullptr;
	c->d.tape_len = 0;
	if (chain && !init_flag && c->zspec) { /* the start positions resolved from intervals of shapes, then one parallel sweep (isg_spec_hip.inc) */
		bool done = false;
		if (spec_update_ZQ(c, c->zspec, base, &done)) return 1;
		if (done) return 0;
	}
	if (chain && !init_flag && c->rs) { /* the start positions resolved block-wise, then one parallel sweep (isg_resolve_hip.inc) */
		bool done = false;
		if (resolve_update_ZQ(c, base, &done)) return 1;
		if (done) return 0;
	}
	if (chain) {
		/* replay schedule: the phase consumes a contiguous run of the stream (2 uniforms per used
		 * locus, plus each individual's Dirichlet).  Its uniforms do not depend on where the
		 * individual boundaries fall, so they are generated up front by the whole chip; the serial
		 * chain kernel then only reads them. */
		uint64_t need = 2 * c->nvalid_total + (uint64_t)(8 * K + 32 > 96 ? 8 * K + 32 : 96) * (uint64_t)c->cfg.N + 4096;
		if (need > c->tape_cap) {
			if (c->d_tape) HIPCHK(hipFree(c->d_tape));
			c->d_tape = nullptr;
			c->tape_cap = 0;
			HIPCHK(hipMalloc((void **)&c->d_tape, sizeof(double) * need));
			c->tape_cap = need;
		}
		prof_begin(c);
		hipLaunchKernelGGL(k_tape, dim3((unsigned)((need + 2047) / 2048)), dim3(256), 0, c->stream, c->d.tab, base, (unsigned long long)need, c->d_tape);
		prof_end(c, "k_tape");
		HIPCHK(hipGetLastError());
		c->d.tape = c->d_tape;
		c->d.tape_len = need;
	}
	const bool coop = chain && c->coop;
	if (coop) {
		HIPCHK(hipMemsetAsync(c->d_coop, 0, sizeof(CoopBuf), c->stream));
		int G = (c->d.Lp + 255) / 256;
		if (G > ISG_COOP_GMAX) G = ISG_COOP_GMAX;
		CoopBuf *cb = (CoopBuf *)c->d_coop;
		/* few enough workgroups for one XCD (32 CUs): start 8 G blocks, every 8th works (see coop_same_xcd) */
		const int pack = (G <= 32 && c->xcd) ? 1 : 0;
		const bool spec = !init_flag && K <= 8 && G * 256 >= c->d.Lp && c->spec;
		/* draw waves + control wave: single pass over the loci, packed 16-bit totals, every workgroup resident */
		const int GP = (c->d.Lp + 64 * ISG_PIPE_DW - 1) / (64 * ISG_PIPE_DW);
		const bool pipe = spec && c->pipe && !pack && 2 * c->d.Lp < 65536 && GP * ISG_PIPE_DW <= 64 * ISG_PIPE_RMAX && GP <= 192;
		const int ppack = (GP <= 32 && c->pipe_xcd) ? 1 : 0;
		if (pipe) {
			const size_t need_words = (size_t)ISG_COOP_RING * GP * ISG_PIPE_DW * ISG_PIPE_STRIDE;
			if (need_words > c->pipe_cap) {
				if (c->d_pipe) HIPCHK(hipFree(c->d_pipe));
				c->d_pipe = nullptr;
				c->pipe_cap = 0;
				HIPCHK(hipMalloc((void **)&c->d_pipe, need_words * sizeof(unsigned long long)));
				c->pipe_cap = need_words;
			}
			HIPCHK(hipMemsetAsync(c->d_pipe, 0, need_words * sizeof(unsigned long long), c->stream));
		}
		HIPCHK(hipMemcpyAsync(c->d_qqsave, c->d.qq, sizeof(double) * (size_t)c->cfg.N * K, hipMemcpyDeviceToDevice, c->stream));
		bool launched = true;
		prof_begin(c);
#define COOP_LAUNCH(KM) do { if (fits_resident(k_zq_coop<KM>, 256, G, c->cfg.device)) hipLaunchKernelGGL((k_zq_coop<KM>), dim3(pack ? 8 * G : G), dim3(256), 0, c->stream, c->d, base, init_flag, c->alpha, cb, c->d_pos, pack); else launched = false; } while (0)
#define SPEC_LAUNCH(KM) do { if (fits_resident(k_zq_spec<KM>, 256, G, c->cfg.device)) hipLaunchKernelGGL((k_zq_spec<KM>), dim3(pack ? 8 * G : G), dim3(256), 0, c->stream, c->d, base, c->alpha, cb, c->d_pos, pack); else launched = false; } while (0)
#define PIPE_LAUNCH(KM) do { if (fits_resident(k_zq_pipe<KM, ISG_PIPE_DW>, 64 * (ISG_PIPE_DW + 1), GP, c->cfg.device)) hipLaunchKernelGGL((k_zq_pipe<KM, ISG_PIPE_DW>), dim3(ppack ? 8 * GP : GP), dim3(64 * (ISG_PIPE_DW + 1)), 0, c->stream, c->d, base, c->alpha, cb, c->d_pipe, c->d_pos, ppack); else launched = false; } while (0)
		if (pipe) {
			switch (K) {
			case 1: case 2: PIPE_LAUNCH(2); break;
			case 3: PIPE_LAUNCH(3); break;
			case 4: PIPE_LAUNCH(4); break;
			case 5: PIPE_LAUNCH(5); break;
			case 6: PIPE_LAUNCH(6); break;
			default: PIPE_LAUNCH(8); break;
			}
		} else if (spec) {
			switch (K) {
			case 1: case 2: SPEC_LAUNCH(2); break;
			case 3: SPEC_LAUNCH(3); break;
			case 4: SPEC_LAUNCH(4); break;
			case 5: SPEC_LAUNCH(5); break;
			case 6: SPEC_LAUNCH(6); break;
			default: SPEC_LAUNCH(8); break;
			}
		} else
		switch (K) {
		case 1: case 2: COOP_LAUNCH(2); break;
		case 3: COOP_LAUNCH(3); break;
		case 4: COOP_LAUNCH(4); break;
		case 5: COOP_LAUNCH(5); break;
		case 6: COOP_LAUNCH(6); break;
		case 7: case 8: COOP_LAUNCH(8); break;
		default:
			if (K <= 12) COOP_LAUNCH(12);
			else if (K <= 16) COOP_LAUNCH(16);
			else if (K <= 24) COOP_LAUNCH(24);
			else COOP_LAUNCH(32);
		}
#undef COOP_LAUNCH
		prof_end(c, pipe ? "k_zq_pipe" : spec ? "k_zq_spec" : "k_zq_coop");
		HIPCHK(hipGetLastError());
		uint64_t used = 0;
		unsigned flags[2] = {0, 0};
		HIPCHK(hipMemcpyAsync(&used, c->d_pos, sizeof(used), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipMemcpyAsync(flags, &cb->abort_flag, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		if (launched && !coop_sweep_failed(c, flags)) {
			host_advance(c, used);
			c->h_qq = false;
			return 0;
		}
		/* redo with the single-workgroup kernel below (see coop_sweep_failed) */
		HIPCHK(hipMemcpyAsync(c->d.qq, c->d_qqsave, sizeof(double) * (size_t)c->cfg.N * K, hipMemcpyDeviceToDevice, c->stream));
		c->zq_fallbacks++;
	}
	prof_begin(c);
	switch (K) { /* small K: exact-size register arrays; larger K: rounded up */
	case 1: case 2: launch_zq<2>(c, chain, base, pos0, stride, init_flag); break;
	case 3: launch_zq<3>(c, chain, base, pos0, stride, init_flag); break;
	case 4: launch_zq<4>(c, chain, base, pos0, stride, init_flag); break;
	case 5: launch_zq<5>(c, chain, base, pos0, stride, init_flag); break;
	case 6: launch_zq<6>(c, chain, base, pos0, stride, init_flag); break;
	case 7: case 8: launch_zq<8>(c, chain, base, pos0, stride, init_flag); break;
	default:
		if (K <= 12) launch_zq<12>(c, chain, base, pos0, stride, init_flag);
		else if (K <= 16) launch_zq<16>(c, chain, base, pos0, stride, init_flag);
		else if (K <= 24) launch_zq<24>(c, chain, base, pos0, stride, init_flag);
		else launch_zq<32>(c, chain, base, pos0, stride, init_flag);
	}
	prof_end(c, chain ? "k_zq_chain" : "k_zq_keyed");
	HIPCHK(hipGetLastError());
	if (chain) {
		uint64_t used = 0;
		HIPCHK(hipMemcpyAsync(&used, c->d_pos, sizeof(used), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		host_advance(c, used);
	}
	c->h_qq = false;
	return 0;
}

/* ---- update_alpha (mcmc.c:1244-1263): factors on the device, ordered product on the host ---- */
extern "C" int isg_update_alpha(isg_ctx *c)
{
	NOT_POLY(c, "isg_update_alpha");
	HIPCHK(hipSetDevice(c->cfg.device));
	const size_t NK = (size_t)c->cfg.N * c->cfg.K;
	if (is_keyed(c)) host_seek(c, iter_base(c) + c->ky[KY_OFFA]);
	isg_cursor cur;
	cur.s = c->rng;
	cur.used = 0;
	cur.tape = nullptr;
	double ralpha = isg_rnormal(&cur, c->alpha, 1.0);
	c->rng = cur.s;
	c->raw_valid = false;
	if (ralpha > 0) {
		prof_begin(c);
		hipLaunchKernelGGL(k_alpha_ratios, dim3((unsigned)((NK + 255) / 256)), dim3(256), 0, c->stream, c->d, ralpha, c->alpha, c->d_ratios);
		prof_end(c, "k_alpha_ratios");
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(c->ratios_h.data(), c->d_ratios, sizeof(double) * NK, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		double mh = 1.0;
		for (size_t k = 0; k < NK; k++) mh *= c->ratios_h[k]; /* in the reference's order: overflow to inf, 0 and NaN included */
		double thr = (1 > mh) ? mh : 1;
		c->alpha = (host_next(c) < thr) ? ralpha : c->alpha;
	}
	return update_P_ahead(c);
}

/* ---- cal_lkh ---- */
extern "C" int isg_cal_lkh(isg_ctx *c)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (c->poly) return poly_cal_lkh(c);
	if (c->cfg.mode == 0) return noadm_cal_lkh(c);
	if (c->cfg.mode == 4) return inbreed_cal_lkh(c);
	if (c->cfg.mode == 5) return indiv_cal_lkh_F(c);
	DevView &d = c->d;
	prof_begin(c);
	if (d.lli && d.mode == 2) hipLaunchKernelGGL((k_loglik_int<256, false>), dim3(d.N), dim3(256), 0, c->stream, d);
	else if (d.lltab || (d.lftab && d.mode == 1)) hipLaunchKernelGGL((k_loglik_tab<256, false>), dim3(d.N), dim3(256), 0, c->stream, d);
	else hipLaunchKernelGGL((k_loglik<256, false>), dim3(d.N), dim3(256), 0, c->stream, d);
	prof_end(c, "k_loglik_lkh");
	prof_begin(c);
	hipLaunchKernelGGL(k_lkh_total<1024>, dim3(1), dim3(1024), 0, c->stream, d, c->d_total);
	prof_end(c, "k_lkh_total");
	HIPCHK(hipGetLastError());
	c->h_lkh = false;
	return 0;
}

#include "isg_poly_hip.inc"
#include "isg_modes_hip.inc"

extern "C" int isg_update_Z(isg_ctx *c, int init_flag) /* mode 0: update_Z, mcmc.c:1094-1120 (zz[i] is returned by isg_get_generation) */
{
	c->ahead_valid = false; /* Z changes: counts requested ahead are stale */
	if (c->poly || c->cfg.mode != 0) return fail("isg_update_Z: mode 0 (-v 0) only");
	HIPCHK(hipSetDevice(c->cfg.device));
	return noadm_update_Z(c, init_flag);
}

extern "C" int isg_update_S_IND(isg_ctx *c) /* mode 3: update_S_IND, mcmc.c:864-884; mode 5: update_F_IND, mcmc.c:888-910 */
{
	if (c->poly || (c->cfg.mode != 3 && c->cfg.mode != 5)) return fail("isg_update_S_IND: modes 3 and 5 (-v 3, -v 5) only");
	HIPCHK(hipSetDevice(c->cfg.device));
	return c->cfg.mode == 3 ? indiv_update_S_IND(c) : indiv_update_F_IND(c);
}

extern "C" int isg_iteration(isg_ctx *c)
{
	/* keyed positions grow with the iteration count; the device's skip-ahead reduces them through a double (isg_mod_u64: exact below 2^52) */
	if (is_keyed(c) && iter_base(c) + c->ky[KY_BLK] >= (1ull << 52)) return fail("isg_iteration: the keyed schedule's stream positions reach 2^52 (the generator's period is 6.95e12 anyway)");
	if (c->poly) {
		HIPCHK(hipSetDevice(c->cfg.device));
		return poly_iteration(c);
	}
	if (isg_update_P(c)) return 1;
	if (c->cfg.mode == 0) { /* mcmc.c:113-115 */
		if (isg_update_Z(c, 0)) return 1;
		if (isg_cal_lkh(c)) return 1;
		c->iter++;
		return 0;
	}
	if (c->cfg.mode == 2) {
		if (isg_update_S_POP(c)) return 1;
		if (isg_update_G(c)) return 1;
	}
	if (c->cfg.mode == 4 && isg_update_S_POP(c)) return 1;
	if (c->cfg.mode == 3) {
		if (isg_update_S_IND(c)) return 1;
		if (isg_update_G(c)) return 1;
	}
	if (c->cfg.mode == 5 && isg_update_S_IND(c)) return 1;
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
	c->ahead_valid = false; /* Z changes: counts requested ahead are stale */
	HIPCHK(hipSetDevice(c->cfg.device));
	if (c->poly) return poly_chain_init(c, initd);
	const int N = c->cfg.N, K = c->cfg.K;
	c->origin = c->rng;
	c->iter = 0;
	HIPCHK(hipMemsetAsync(c->d_err, 0, 4 * sizeof(unsigned), c->stream));
	if (c->cfg.mode == 0) return isg_update_Z(c, 1); /* mcmc_POP_no_admixture does not draw alpha (no initial_chn) */
	c->alpha = host_next(c) * 10;
	if (c->cfg.mode == 2) {
		isg_cursor cur;
		cur.s = c->rng;
		cur.used = 0;
	cur.tape = nullptr;
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
		HIPCHK(hipMemcpyAsync(c->d_S, c->S.data(), sizeof(double) * K, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(c->d_state, c->state.data(), sizeof(int) * K, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		c->h_gen = c->h_S = true;
	}
	if (c->cfg.mode == 3) { /* mcmc_INDV_selfing, mcmc.c:324-331 (prior_flag 0); sic: the generations are not clamped here */
		if (is_keyed(c)) {
			for (int i = 0; i < N; i++) {
				host_seek(c, 1 + 2 * (uint64_t)i);
				c->S[i] = host_next(c);
				c->gen[i] = isg_rgeom_u(host_next(c), 1 - c->S[i]);
			}
		} else {
			for (int i = 0; i < N; i++) c->S[i] = host_next(c);
			for (int i = 0; i < N; i++) c->gen[i] = isg_rgeom_u(host_next(c), 1 - c->S[i]);
		}
		HIPCHK(hipMemcpyAsync(c->d.gen, c->gen.data(), sizeof(int) * N, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(c->d_S, c->S.data(), sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		c->h_gen = c->h_S = true;
	}
	if (c->cfg.mode == 5) { /* mcmc_INDV_inbreedcoff, mcmc.c:412-415 (prior_flag 0) */
		for (int i = 0; i < N; i++) {
			if (is_keyed(c)) host_seek(c, 1 + 2 * (uint64_t)i);
			c->S[i] = host_next(c);
		}
		HIPCHK(hipMemcpyAsync(c->d_S, c->S.data(), sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		c->h_S = true;
	}
	if (c->cfg.mode == 4) { /* mcmc_POP_inbreedcoff, mcmc.c:255-259: coefficients and their states, no generations */
		for (int k = 0; k < K; k++) {
			c->S[k] = (double)initd[k];
			if (c->cfg.back_refl == 0) {
				int st = isg_dt_stat(c->S[k]);
				if (st < 0) return fail("The value of selfing rate or inbreeding coefficient is beyond [0,1]!");
				c->state[k] = st;
			}
		}
		HIPCHK(hipMemcpyAsync(c->d_S, c->S.data(), sizeof(double) * K, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(c->d_state, c->state.data(), sizeof(int) * K, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		c->h_S = true;
	}
	return isg_update_ZQ(c, 1);
}

/* ---- getters / setters ---- */
extern "C" int isg_get_amax(isg_ctx *c, int32_t *a) { *a = c->Amax; return 0; }
extern "C" int isg_get_z(isg_ctx *c, int32_t *z)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (c->poly) return poly_get_bytes(c, c->poly->p.z, z);
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
	c->ahead_valid = false; /* Z changes: counts requested ahead are stale */
	NOT_POLY(c, "isg_set_z");
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
	if (c->poly && c->poly->freq_host) { memcpy(f, c->freq.data(), sizeof(double) * c->freq.size()); return 0; } /* drawn on the host */
	if (download_freq(c)) return 1;
	memcpy(f, c->freq.data(), sizeof(double) * c->freq.size());
	return 0;
}
extern "C" int isg_set_freq(isg_ctx *c, const double *f)
{
	NOT_POLY(c, "isg_set_freq");
	HIPCHK(hipSetDevice(c->cfg.device));
	memcpy(c->freq.data(), f, sizeof(double) * c->freq.size());
	if (upload_freq(c)) return 1;
	HIPCHK(hipStreamSynchronize(c->stream));
	return 0;
}
extern "C" int isg_get_qq(isg_ctx *c, double *q)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (ensure_qq(c)) return 1;
	memcpy(q, c->qq.data(), sizeof(double) * c->qq.size());
	return 0;
}
extern "C" int isg_set_qq(isg_ctx *c, const double *q)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (ensure_qq(c)) return 1; /* keeps qqnum */
	memcpy(c->qq.data(), q, sizeof(double) * c->qq.size());
	HIPCHK(hipMemcpy(c->d.qq, q, sizeof(double) * c->qq.size(), hipMemcpyHostToDevice));
	return 0;
}
extern "C" int isg_get_qqnum(isg_ctx *c, double *q)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (ensure_qq(c)) return 1;
	for (size_t i = 0; i < c->qqnum.size(); i++) q[i] = (double)c->qqnum[i];
	return 0;
}
extern "C" int isg_get_generation(isg_ctx *c, int32_t *g)
{
	NOT_POLY(c, "isg_get_generation");
	HIPCHK(hipSetDevice(c->cfg.device));
	if (ensure_gen(c)) return 1;
	memcpy(g, c->gen.data(), sizeof(int) * c->gen.size());
	return 0;
}
extern "C" int isg_set_generation(isg_ctx *c, const int32_t *g)
{
	NOT_POLY(c, "isg_set_generation");
	HIPCHK(hipSetDevice(c->cfg.device));
	memcpy(c->gen.data(), g, sizeof(int) * c->gen.size());
	HIPCHK(hipMemcpy(c->d.gen, g, sizeof(int) * c->gen.size(), hipMemcpyHostToDevice));
	c->h_gen = true;
	return 0;
}
extern "C" int isg_get_self_rates(isg_ctx *c, double *s)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (ensure_S(c)) return 1;
	memcpy(s, c->S.data(), sizeof(double) * c->S.size());
	return 0;
}
extern "C" int isg_set_self_rates(isg_ctx *c, const double *s)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (ensure_S(c)) return 1;
	memcpy(c->S.data(), s, sizeof(double) * c->S.size());
	HIPCHK(hipMemcpy(c->d_S, c->S.data(), sizeof(double) * c->S.size(), hipMemcpyHostToDevice));
	return 0;
}
extern "C" int isg_get_state(isg_ctx *c, int32_t *s)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (ensure_S(c)) return 1;
	memcpy(s, c->state.data(), sizeof(int) * c->state.size());
	return 0;
}
extern "C" int isg_get_indvlkh(isg_ctx *c, double *v)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (ensure_lkh(c)) return 1;
	memcpy(v, c->indvlkh.data(), sizeof(double) * c->indvlkh.size());
	return 0;
}
extern "C" int isg_get_alpha(isg_ctx *c, double *a) { *a = c->alpha; return 0; }
extern "C" int isg_set_alpha(isg_ctx *c, double a) { c->alpha = a; return 0; }
extern "C" int isg_get_totallkh(isg_ctx *c, double *t)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (ensure_lkh(c)) return 1;
	*t = c->totallkh;
	return 0;
}

/* ---- ploidy 4 ---- */
extern "C" int isg_ctx_create_poly(const isg_config *cfg, const int32_t *allelenum, const int32_t *seqdata, const int32_t *alleleid, isg_ctx **out)
{
	*out = nullptr;
	if (cfg->P != 4) return fail("isg_ctx_create_poly: ploidy must be 4 (autotetraploid, -ap 1)");
	if (cfg->N < 1 || cfg->L < 1) return fail("isg_ctx_create_poly: empty problem");
	std::vector<int32_t> seq((size_t)cfg->N * cfg->L * 4);
	for (size_t e = 0; e < (size_t)cfg->N * cfg->L; e++)
		for (int k = 0; k < 4; k++) seq[e * 4 + k] = k < alleleid[e] ? seqdata[e * 4 + k] : -1;
	return poly_ctx_create(cfg, allelenum, seq.data(), out);
}
extern "C" int isg_poly_update_geno(isg_ctx *c)
{
	if (!c->poly) return fail("isg_poly_update_geno: not a ploidy 4 context");
	HIPCHK(hipSetDevice(c->cfg.device));
	return poly_geno_sweep(c, 0);
}
extern "C" int isg_get_poly_geno(isg_ctx *c, int32_t *g)
{
	if (!c->poly) return fail("isg_get_poly_geno: not a ploidy 4 context");
	HIPCHK(hipSetDevice(c->cfg.device));
	return poly_get_bytes(c, c->poly->p.geno, g);
}
extern "C" int isg_get_poly_gs(isg_ctx *c, int32_t *gs, int32_t *gcount /* [L] genotypes per locus */)
{
	if (!c->poly) return fail("isg_get_poly_gs: not a ploidy 4 context");
	*gs = c->poly->p.GS;
	if (gcount)
		for (int j = 0; j < c->cfg.L; j++) gcount[j] = c->poly->p.allo ? isg_allo_G(c->allelenum[j]) : isg_poly_G(c->allelenum[j]);
	return 0;
}
extern "C" int isg_get_poly_freq2(isg_ctx *c, double *f) /* [K][L][Amax]: the second subgenome's allele frequencies (UPMCMC.freq2, -ap 0) */
{
	if (!c->poly || !c->poly->p.allo) return fail("isg_get_poly_freq2: not an allotetraploid context");
	if (!c->poly->freq_host) { /* drawn on the device (keyed schedule) */
		const int L = c->cfg.L, K = c->cfg.K, A = c->Amax, KP = c->poly->p.KP;
		HIPCHK(hipSetDevice(c->cfg.device));
		HIPCHK(hipMemcpyAsync(c->freq_stage.data(), c->poly->p.freq2, sizeof(double) * (size_t)L * A * KP, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		for (int k = 0; k < K; k++)
			for (int j = 0; j < L; j++)
				for (int a = 0; a < A; a++) c->poly->freq2_h[((size_t)k * L + j) * A + a] = c->freq_stage[((size_t)j * A + a) * KP + k];
	}
	memcpy(f, c->poly->freq2_h.data(), sizeof(double) * c->poly->freq2_h.size());
	return 0;
}
extern "C" int isg_get_poly_table(isg_ctx *c, int which, float *out)
{
	if (!c->poly) return fail("isg_get_poly_table: not a ploidy 4 context");
	HIPCHK(hipSetDevice(c->cfg.device));
	const PolyDev &p = c->poly->p;
	HIPCHK(hipMemcpyAsync(out, which ? p.genofreq : p.exfreq, sizeof(float) * (size_t)p.K * p.L * p.GS, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return 0;
}

/* ---- profiling ---- */
/* ---- store_chn on the device (mcmc.c:1320-1456; allocate_chn + initialize_chn, mcmc.c:588-738) ----
 * The reference keeps multiplicative running means  m <- m * ((n + x / m) / (n + 1)), seeded with 1  (mcmc.c:1327-1332 and
 * every block after it).  The same two divisions and one multiplication in double, no contraction: the same bits as
 * the host loop (tests/test_gpu_parity.py::test_store_chn_on_device...).  Only what is O(N K) or O(K L A) lives here;
 * the scalars and the per-cluster rates stay with the caller. */
__device__ __forceinline__ void runmean_dev(double *m, double x, double n, double n1)
{
	const double v = *m;
	*m = (v != 0) ? v * ((n + x / v) / n1) : x / n1;
}
__global__ void k_store_fill(double *p, size_t n, double v)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = v;
}
__global__ void k_store_chn(const double *qq, double *mqq, double *mqq2, size_t nqq, const double *lkh, double *mlkh, const int *gen, double *mgen,
			    double *mgen2, size_t nind, const double *freq, double *mfreq, double *mfreq2, size_t nfreq, double n, double n1)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (mqq && i < nqq) {
		const double x = qq[i];
		runmean_dev(&mqq[i], x, n, n1);
		runmean_dev(&mqq2[i], x * x, n, n1);
	}
	if (i < nind) {
		runmean_dev(&mlkh[i], lkh[i], n, n1);
		if (mgen) {
			const int g = gen[i];
			runmean_dev(&mgen[i], (double)g, n, n1);
			runmean_dev(&mgen2[i], (double)(g * g), n, n1);
		}
	}
	if (mfreq && i < nfreq) {
		const double x = freq[i];
		runmean_dev(&mfreq[i], x, n, n1);
		runmean_dev(&mfreq2[i], x * x, n, n1);
	}
}
static void store_free(isg_ctx *c)
{
	(void)hipFree(c->st_qq); (void)hipFree(c->st_qq2); (void)hipFree(c->st_lkh); (void)hipFree(c->st_gen); (void)hipFree(c->st_gen2);
	(void)hipFree(c->st_freq); (void)hipFree(c->st_freq2);
	c->st_qq = c->st_qq2 = c->st_lkh = c->st_gen = c->st_gen2 = c->st_freq = c->st_freq2 = nullptr;
	c->st_on = false;
	c->st_step = 0;
}
static int store_alloc_ones(isg_ctx *c, double **p, size_t n)
{
	HIPCHK(hipMalloc((void **)p, sizeof(double) * n));
	hipLaunchKernelGGL(k_store_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, *p, n, 1.0);
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int isg_store_begin(isg_ctx *c, int with_freq)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	store_free(c);
	const size_t N = (size_t)c->cfg.N, K = (size_t)c->cfg.K;
	const bool has_qq = c->poly || c->cfg.mode != 0, has_gen = !c->poly && (c->cfg.mode == 2 || c->cfg.mode == 3);
	if (with_freq && c->poly) return fail("isg_store_begin: allele frequencies are only accumulated for ploidy 2 (mcmc.c:1436)");
	if (has_qq && (store_alloc_ones(c, &c->st_qq, N * K) || store_alloc_ones(c, &c->st_qq2, N * K))) return 1;
	if (store_alloc_ones(c, &c->st_lkh, N)) return 1;
	if (has_gen && (store_alloc_ones(c, &c->st_gen, N) || store_alloc_ones(c, &c->st_gen2, N))) return 1;
	if (with_freq) {
		const size_t nf = (size_t)c->cfg.L * c->Amax * c->d.KP;
		if (store_alloc_ones(c, &c->st_freq, nf) || store_alloc_ones(c, &c->st_freq2, nf)) return 1;
	}
	c->st_on = true;
	c->st_step = 0;
	return 0;
}
extern "C" int isg_store_step(isg_ctx *c)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (!c->st_on) return fail("isg_store_step: isg_store_begin first");
	if (c->poly && c->poly->freq_host && c->st_freq) return fail("isg_store_step: ploidy 4 keeps no frequency means");
	const size_t N = (size_t)c->cfg.N, K = (size_t)c->cfg.K, nqq = c->st_qq ? N * K : 0, nf = c->st_freq ? (size_t)c->cfg.L * c->Amax * c->d.KP : 0;
	size_t n = nqq > N ? nqq : N;
	if (nf > n) n = nf;
	prof_begin(c);
	hipLaunchKernelGGL(k_store_chn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const double *)c->d.qq, c->st_qq, c->st_qq2, nqq,
			   (const double *)c->d.indvlkh, c->st_lkh, (const int *)c->d.gen, c->st_gen, c->st_gen2, N, (const double *)c->d.freq, c->st_freq, c->st_freq2,
			   nf, (double)c->st_step, (double)(1 + c->st_step));
	prof_end(c, "k_store_chn");
	HIPCHK(hipGetLastError());
	c->st_step++;
	return 0;
}
/* null pointers are skipped; freq / freq2 come back in the reference's order [K][L][Amax] */
extern "C" int isg_store_fetch(isg_ctx *c, double *qq, double *qq2, double *indvlkh, double *gen, double *gen2, double *freq, double *freq2, long *steps)
{
	HIPCHK(hipSetDevice(c->cfg.device));
	if (!c->st_on) return fail("isg_store_fetch: isg_store_begin first");
	const size_t N = (size_t)c->cfg.N, K = (size_t)c->cfg.K;
	if ((qq || qq2) && !c->st_qq) return fail("isg_store_fetch: this mode keeps no qq means");
	if ((gen || gen2) && !c->st_gen) return fail("isg_store_fetch: this mode keeps no generation means");
	if ((freq || freq2) && !c->st_freq) return fail("isg_store_fetch: isg_store_begin was called without frequencies");
	if (qq) HIPCHK(hipMemcpyAsync(qq, c->st_qq, sizeof(double) * N * K, hipMemcpyDeviceToHost, c->stream));
	if (qq2) HIPCHK(hipMemcpyAsync(qq2, c->st_qq2, sizeof(double) * N * K, hipMemcpyDeviceToHost, c->stream));
	if (indvlkh) HIPCHK(hipMemcpyAsync(indvlkh, c->st_lkh, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
	if (gen) HIPCHK(hipMemcpyAsync(gen, c->st_gen, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
	if (gen2) HIPCHK(hipMemcpyAsync(gen2, c->st_gen2, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (int which = 0; which < 2; which++) {
		double *dst = which ? freq2 : freq;
		if (!dst) continue;
		const int L = c->cfg.L, A = c->Amax, KP = c->d.KP;
		HIPCHK(hipMemcpyAsync(c->freq_stage.data(), which ? c->st_freq2 : c->st_freq, sizeof(double) * (size_t)L * A * KP, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		for (int k = 0; k < (int)K; k++)
			for (int j = 0; j < L; j++)
				for (int a = 0; a < A; a++) dst[((size_t)k * L + j) * A + a] = c->freq_stage[((size_t)j * A + a) * KP + k];
	}
	if (steps) *steps = c->st_step;
	return 0;
}

extern "C" long isg_zq_fallbacks(isg_ctx *c) { return c->zq_fallbacks; }
extern "C" int isg_zq_spec_stats(isg_ctx *c, long out[10])
{
	for (int k = 0; k < 10; k++) out[k] = 0;
	if (!c->zspec) return 0;
	out[0] = c->zspec->sweeps;
	out[1] = c->zspec->done;
	out[2] = c->zspec->lost;
	out[3] = (long)c->zspec->last_probes;
	out[4] = (long)c->zspec->last_fail;
	out[5] = (long)c->zspec->last_rounds;
	out[6] = (long)c->zspec->walk.plan.table_bytes;
	out[7] = (long)c->zspec->walk.plan.seg.size();
	out[8] = (long)c->zspec->lost_fail;
	out[9] = c->zspec->retried;
	return 0;
}
extern "C" int isg_p_device_stats(isg_ctx *c, long out[10])
{
	for (int k = 0; k < 10; k++) out[k] = 0;
	if (!c->pdev) return 0;
	const WalkRun &w = c->pdev->walk;
	out[0] = c->pdev->sweeps - c->pdev->fallbacks;
	out[1] = c->pdev->fallbacks;
	out[2] = (long)w.plan.seg.size();
	out[3] = (long)w.plan.blk.size();
	out[4] = (long)w.plan.table_bytes;
	out[5] = (long)(1000.0 * w.sigma);
	out[6] = (long)(1000.0 * w.scale);
	out[7] = (long)(1000.0 * w.kwin);
	out[8] = c->pdev->retried;
	out[9] = w.fails;
	return 0;
}
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

/* ---- host-side self test of the integer/float shortcuts used on the device (no GPU needed) ---- */
/* device-to-device copy bandwidth of this GPU with 16-byte loads and stores (the microarchitecture guide's float4 copy): what a
 * streaming kernel can at best reach here; bench.py reports roofline fractions against it next to the 8 TB/s specification */
__global__ void __launch_bounds__(256) k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
	const size_t stride = (size_t)gridDim.x * 256;
	for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) dst[k] = src[k];
}
extern "C" int isg_copy_bandwidth(int device, size_t bytes, int reps, double *gbs)
{
	*gbs = 0;
	HIPCHK(hipSetDevice(device));
	const size_t n = bytes / 16;
	uint4 *a = nullptr, *b = nullptr;
	HIPCHK(hipMalloc((void **)&a, n * 16));
	if (hipMalloc((void **)&b, n * 16) != hipSuccess) { (void)hipFree(a); return fail("isg_copy_bandwidth: out of device memory"); }
	HIPCHK(hipMemset(a, 1, n * 16));
	hipEvent_t e0, e1;
	HIPCHK(hipEventCreate(&e0));
	HIPCHK(hipEventCreate(&e1));
	int cus = 256;
	(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
	const dim3 grid((unsigned)(cus * 8));
	for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_copy16, grid, dim3(256), 0, 0, (const uint4 *)a, b, n);
	HIPCHK(hipEventRecord(e0, 0));
	for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_copy16, grid, dim3(256), 0, 0, (const uint4 *)a, b, n);
	HIPCHK(hipEventRecord(e1, 0));
	HIPCHK(hipEventSynchronize(e1));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, e0, e1));
	*gbs = 2.0 * (double)(n * 16) * reps / (ms * 1e-3) / 1e9; /* read + written */
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	(void)hipFree(a); (void)hipFree(b);
	return 0;
}

extern "C" int isg_selftest(void)
{
	static const uint32_t M[3] = {ISG_M1, ISG_M2, ISG_M3}, A[3] = {ISG_A1, ISG_A2, ISG_A3};
	static const double MD[3] = {30269.0, 30307.0, 30323.0};
	for (int g = 0; g < 3; g++) {
		volatile float inv = 1.0f / (float)M[g];
		volatile double md = MD[g];
		const double invd = 1.0 / md;
		for (uint32_t s = 0; s < M[g] + 2000; s++) {
			if (isg_lcg_fast(s, A[g], M[g], inv) != (A[g] * s) % M[g]) return fail("isg_selftest: isg_lcg_fast differs from (a*s) % m");
			if (isg_wh_div(s, md, invd) != (double)s / md) return fail("isg_selftest: isg_wh_div differs from s / m");
		}
	}
	isg_wh_tables tab;
	isg_wh_tables_init(&tab);
	isg_wh s0 = {13, 4, 1972}, s = s0;
	for (uint64_t n = 1; n <= 70000; n++) {
		isg_wh_step(&s);
		if (n % 997 == 0 || n < 600) {
			isg_wh j = isg_wh_jump(&tab, s0, n), j32 = isg_wh_jump32(&tab, s0, (uint32_t)n);
			if (j.s1 != s.s1 || j.s2 != s.s2 || j.s3 != s.s3 || j32.s1 != s.s1 || j32.s2 != s.s2 || j32.s3 != s.s3) return fail("isg_selftest: skip-ahead differs from stepping");
		}
	}
	return 0;
}

#include "isg_rccl.inc"

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
