/*
 * instruct_hip.h -- C ABI of the MI355X-native InStruct MCMC hot path (libinstruct_hip.so).
 *
 * This is the drop-in boundary for the per-iteration sampler of slowkoni/InStruct: every entry
 * point below replaces one function of the reference's mcmc.c (cited per declaration) and takes
 * plain pointers and sizes only.  The reference driver keeps calling
 *     CHAIN mcmc_updating(SEQDATA, INIT, int chn, CONVG *)   (mcmc.h:56, call sites InStruct.c:184,565)
 *     void  free_chain(CHAIN *, SEQDATA)                     (mcmc.h:57)
 * which instruct_amd/host/mcmc_hip.c implements on top of this ABI with the reference's struct
 * layouts (see INTEGRATION.md for the link line a maintainer adds).
 *
 * Data layout handed over at context creation (the reference's in-memory genotype layout,
 * data_interface.h:10-56): allele codes int32 [N][L][P] (0..allelenum[j]-1), missindx int32
 * [N][L] (1 = locus missing for that individual), allelenum int32 [L].  On the device genotypes
 * and ancestry assignments Z are packed to one byte per allele copy, N x L-major.
 *
 * Errors: every function returns 0 on success, nonzero on failure; isg_last_error() returns the
 * message (the mcmc_updating shim turns it into the reference's nrerror() convention,
 * nrutil.c:9-16).  There is no CPU fallback: without a usable gfx950 device isg_ctx_create fails.
 *
 * RNG schedules (cfg.rng_sched):
 *   ISG_SCHED_REPLAY  the single Wichmann-Hill stream of random.c:14-47 is consumed at exactly
 *                     the positions the reference consumes it (Z assignments, allele counts,
 *                     generations and seeds are bit-identical to the reference run);
 *   ISG_SCHED_KEYED   same generator, same samplers, but each consumer starts at a stream
 *                     position that only depends on (iteration, phase, index) -- "keyed layout"
 *                     below -- so that all consumers run concurrently.
 */
#ifndef INSTRUCT_HIP_H
#define INSTRUCT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISG_SCHED_REPLAY 0
#define ISG_SCHED_KEYED 1

typedef struct isg_ctx isg_ctx;

typedef struct {
	int32_t N;         /* individuals            SEQDATA.totalsize */
	int32_t L;         /* polymorphic loci       SEQDATA.locinum   */
	int32_t P;         /* ploidy (2)             SEQDATA.ploid     */
	int32_t K;         /* clusters (<= 32)       SEQDATA.popnum    */
	int32_t mode;      /* 0 no admixture, 1 admixture, 2 population selfing rates, 3 individual selfing rates, 4 / 5 population / individual inbreeding coefficients (InStruct.c:58, mcmc.c:63-87) */
	int32_t type_freq; /* -y (InStruct.c:46) */
	int32_t back_refl; /* -e (InStruct.c:45) */
	int32_t rng_sched; /* ISG_SCHED_* */
	int32_t device;    /* HIP device ordinal */
	int32_t reserved[7]; /* reserved[0] (ploidy 4 only): 1 = allotetraploid, -ap 0 (SEQDATA.autopoly == 0); 0 = autotetraploid */
} isg_config;

/* problem upload + device state allocation (replaces allocate_node, mcmc.c:506-546) */
int isg_ctx_create(const isg_config *cfg, const int32_t *allelenum, const int32_t *geno, const int32_t *missindx, isg_ctx **out);
void isg_ctx_destroy(isg_ctx *ctx);
const char *isg_last_error(void);

/* RNG stream shared with the driver (random.c:50-63 setseeds/printseeds; ran1 random.c:34-47) */
int isg_set_seeds(isg_ctx *ctx, long s1, long s2, long s3);
int isg_get_seeds(isg_ctx *ctx, long seeds[3]);
double isg_ran1(isg_ctx *ctx);

/* chain set-up: initial_chn's alpha draw (mcmc.c:479), generation / selfing-rate initialisation
 * (mcmc.c:196-205) and update_ZQ(init_flag = 1) (mcmc.c:206).  initd: K floats (INIT.initd[chn]) */
int isg_chain_init(isg_ctx *ctx, const float *initd);

/* one call per reference sweep */
int isg_update_P(isg_ctx *ctx);                 /* mcmc.c:799-861  */
int isg_update_S_POP(isg_ctx *ctx);             /* mcmc.c:913-983; mode 4: update_inbreedcoff_POP mcmc.c:986-1051 (coefficients in the self_rates slots) */
int isg_update_Z(isg_ctx *ctx, int init_flag);  /* mode 0: update_Z mcmc.c:1094-1120 (zz[i] via isg_get_generation, mirrored into z) */
int isg_update_S_IND(isg_ctx *ctx);             /* mode 3: update_S_IND mcmc.c:864-884; mode 5: update_F_IND mcmc.c:888-910 (N values in the self_rates slots) */
int isg_update_G(isg_ctx *ctx);                 /* mcmc.c:1053-1091 (+ log_ld_indv :1726-1773) */
int isg_update_ZQ(isg_ctx *ctx, int init_flag); /* mcmc.c:1122-1203 */
int isg_update_alpha(isg_ctx *ctx);             /* mcmc.c:1244-1263 */
int isg_cal_lkh(isg_ctx *ctx);                  /* mcmc.c:1916-1942 */
int isg_iteration(isg_ctx *ctx);                /* the loop body mcmc.c:210-215 (mode 2) / 152-155 (mode 1) */
int isg_run(isg_ctx *ctx, long n_iterations);
int isg_iter_advance(isg_ctx *ctx);             /* sweep-by-sweep drivers: counts an iteration (keyed positions depend on it) */

/* allele counts seqpop[K][L][Amax] of the current Z (the count nest mcmc.c:810-845) */
int isg_count_alleles(isg_ctx *ctx, int32_t *counts);

/* state download, reference layouts (UPMCMC, mcmc.h:12-27) */
int isg_get_z(isg_ctx *ctx, int32_t *z);                 /* [N][L][P]; -1 where the locus is unused */
int isg_get_freq(isg_ctx *ctx, double *freq);            /* [K][L][Amax] */
int isg_get_qq(isg_ctx *ctx, double *qq);                /* [N][K] */
int isg_get_qqnum(isg_ctx *ctx, double *qqnum);          /* [N][K] */
int isg_get_generation(isg_ctx *ctx, int32_t *gen);      /* [N] */
int isg_get_self_rates(isg_ctx *ctx, double *s);         /* [K] */
int isg_get_state(isg_ctx *ctx, int32_t *state);         /* [K] (-e 0) */
int isg_get_indvlkh(isg_ctx *ctx, double *indvlkh);      /* [N] */
int isg_get_alpha(isg_ctx *ctx, double *alpha);
int isg_get_totallkh(isg_ctx *ctx, double *totallkh);
int isg_get_amax(isg_ctx *ctx, int32_t *amax);

/* state upload (tests; restarting a chain from saved state) */
int isg_set_z(isg_ctx *ctx, const int32_t *z);
int isg_set_freq(isg_ctx *ctx, const double *freq);
int isg_set_qq(isg_ctx *ctx, const double *qq);
int isg_set_generation(isg_ctx *ctx, const int32_t *gen);
int isg_set_self_rates(isg_ctx *ctx, const double *s);
int isg_set_alpha(isg_ctx *ctx, double alpha);

/*
 * keyed layout, positions counted in uniforms from the chain origin (the stream state when
 * isg_chain_init is entered):
 *   0                     alpha
 *   1 + 2 i               generation init of individual i
 *   ZI0 + i SZ            ZQ(init) of individual i           ZI0 = 1 + 2N, SZ = P L + 16 K + 16
 *   iteration t base B = B0 + t BLK,  B0 = ZI0 + N SZ
 *   B + (k L + j) SP      update_P Dirichlet (k, j)          SP = 16 Amax + 16
 *   B + offS              update_S_POP                        offS = K L SP
 *   B + offG + 2 i        update_G individual i              offG = offS + 4 K
 *   B + offZ + i SZ       update_ZQ individual i             offZ = offG + 2 N
 *   B + offA              update_alpha                        offA = offZ + N SZ;  BLK = offA + 4
 * out[9] = {SP, SZ, ZI0, B0, offS, offG, offZ, offA, BLK}
 */
int isg_keyed_layout(isg_ctx *ctx, uint64_t out[9]);

/*
 * Ploidy 4 (autotetraploid, `-p 4 -ap 1`): the chain of poly_geno.c:85-116 (mcmc_POP_tetra_selfing).
 * seqdata int32 [N][L][4]: the sorted distinct allele codes observed for (individual, locus) in the first
 * alleleid[i][j] entries (SEQDATA.seqdata / SEQDATA.alleleid as transform_data2 leaves them,
 * data_interface.c:571-669; alleleid 0 = missing).  cfg.P = 4; both schedules.  Keyed layout for ploidy 4
 * (amb = number of (individual, locus) pairs with 2 or 3 distinct alleles -- allotetraploid: 2, 3 or 4 --, rank = their order i-major):
 *   0 alpha | 1 + rank  initial_geno | ZI0 = 1 + amb, ZQ(init) of i at ZI0 + i SZ, SZ = 4 L + 16 K + 16
 *   B = B0 + t BLK, B0 = ZI0 + N SZ:  B + (k L + j) SP  update_P_auto (SP = 16 Amax + 16; allotetraploid: both subgenomes' Dirichlets, SP = 32 Amax + 32) | B + offS  update_S_POP
 *   (offS = K L SP) | B + offZ + i SZ  update_ZQ (offZ = offS + 4 K) | B + offGE + rank  update_geno
 *   (offGE = offZ + N SZ) | BLK = offGE + amb + 4;  isg_keyed_layout: {SP, SZ, ZI0, B0, offS, offGE, offZ, offGE, BLK}
 * The generic entry points then run the ploidy-4 sweeps:
 *   isg_chain_init      initial_chn alpha + initial_geno + update_ZQ(1)   poly_geno.c:85-96, 316-369
 *   isg_update_P        update_P_auto + calc_exfreq_auto                  poly_geno.c:390-438, 1515-1590
 *   isg_update_S_POP    update_S_POP (tables auto_genfreq :1803-2028)     poly_geno.c:584-643
 *   isg_update_ZQ       update_ZQ                                         poly_geno.c:750-836
 *   isg_poly_update_geno update_geno (choose_two/tri_auto :854-960)       poly_geno.c:520-580
 *   isg_cal_lkh         cal_lkd                                           poly_geno.c:715-735
 *   isg_iteration       P, S_POP, ZQ, geno, lkd in the reference's order  poly_geno.c:98-116
 * isg_update_G / isg_update_alpha fail: the reference has no such sweep on this path.
 *
 * Allotetraploid (`-p 4 -ap 0`, cfg.reserved[0] = 1; replay schedule): a genotype is a pair of diploid genotypes, copies
 * 0, 1 from the first subgenome (allele frequencies freq), copies 2, 3 from the second (freq2).  Same entry points:
 *   isg_update_P = update_P_allo + calc_exfreq_allo (poly_geno.c:441-518, 1592-1670; per (cluster, locus) the first
 *   subgenome's Dirichlet, then the second's); update_S_POP with allo_genfreq (:2122-2304); update_ZQ unchanged (it uses
 *   freq for all four copies, poly_geno.c:772); isg_poly_update_geno = choose_two / tri / tetra_allo (:962-1215: 7, 12, 6
 *   candidates, one uniform per locus with two or more observed alleles); cal_lkd with the two-subgenome terms (:1262-1282).
 */
int isg_ctx_create_poly(const isg_config *cfg, const int32_t *allelenum, const int32_t *seqdata, const int32_t *alleleid, isg_ctx **out);
int isg_poly_update_geno(isg_ctx *ctx);
int isg_get_poly_geno(isg_ctx *ctx, int32_t *geno);               /* [N][L][4] imputed genotypes (UPMCMC.geno), -1 unused */
int isg_get_poly_gs(isg_ctx *ctx, int32_t *gs, int32_t *gcount);  /* table row stride; genotypes per locus [L] (may be NULL) */
int isg_get_poly_table(isg_ctx *ctx, int which, float *out);      /* 0 exfreq, 1 genofreq: float [K][L][gs] */
int isg_get_poly_freq2(isg_ctx *ctx, double *freq2);              /* [K][L][Amax], allotetraploid contexts */

/* CHAIN running means on the device: allocate_chn + initialize_chn (mcmc.c:588-738) and store_chn (mcmc.c:1320-1456)
 * for everything that is O(N K) or O(K L A) -- qq, qq2, indvlkh, gen, gen2 and, with_freq (-pf 1, ploidy 2), freq,
 * freq2.  Same multiplicative running mean, same bits as the reference's host loop; a stored step moves nothing
 * over PCIe.  The scalars (totallkh, totallkh2) and the rate vectors stay with the caller (instruct_amd/host).
 * isg_store_fetch: null pointers are skipped; freq / freq2 come back as [K][L][Amax]; *steps = stored steps. */
int isg_store_begin(isg_ctx *ctx, int with_freq);
int isg_store_step(isg_ctx *ctx);
int isg_store_fetch(isg_ctx *ctx, double *qq, double *qq2, double *indvlkh, double *gen, double *gen2, double *freq, double *freq2, long *steps);

/* Replay-schedule update_ZQ runs as several cooperating workgroups that hand counts to each other by polling.  If such a
 * sweep cannot complete (the GPU is shared and a workgroup was not scheduled in time, or a Dirichlet ran past the uniform
 * budget) it is redone from the same stream position by the single-workgroup kernel: same results, only slower.  This
 * returns how many sweeps of this context took that path (diagnostics; INSTRUCT_ZQ_TEST_ABORT=n forces the n-th one). */
long isg_zq_fallbacks(isg_ctx *ctx);
/* Replay-schedule update_ZQ (K <= 8) first RESOLVES the start position of every individual, a block of individuals at a
 * time on the whole chip (all blocks in one launch, or one launch per block), then runs the sweep as one parallel pass (instruct_amd/csrc/isg_resolve_hip.inc;
 * INSTRUCT_ZQ_RESOLVE=0 selects the chain kernels).  Diagnostics of the last sweep:
 * out = {blocks, blocks ended early by a window miss, kernel launches, individuals per block, units per block, 1000 mu,
 * 1000 sigma, draws redone exactly (diagnostic builds)} */
int isg_zq_resolve_stats(isg_ctx *ctx, long out[8]);
/* Host only (no device needed): the resolver's plan for `units` workgroups per block given the rejection statistics (mu, sigma: mean and
 * spread of a Dirichlet's rejected attempts) and the window parameters (a: half-width in sigma, shape: taper along the block).
 * lo, w: per individual of the block its first candidate and number of candidates; ur, uo, cslot: per unit its individual, first
 * candidate and slot in the walk's table image (arrays of 64 / 64 / 512 / 512 / 512 shorts).  Returns individuals per block, *nunits =
 * units used.  For tests of the plan's invariants. */
int isg_zq_resolve_plan(int units, double mu, double sigma, double a, double shape, short *lo, short *w, short *ur, short *uo, short *cslot, int *nunits);

/* Replay-schedule update_ZQ, first choice (instruct_amd/csrc/isg_spec_hip.inc): the start positions resolved from INTERVALS of the
 * Dirichlets' shapes -- expected cluster counts +- k sigma per individual, accept bits that are the same at both ends of the
 * interval are certain without any Z draw, the uncertain candidates near the trajectory are probed exactly -- then the sweep in
 * one parallel pass with every Dirichlet's consumption checked.  A sweep this path cannot settle (many small clusters: every
 * candidate uncertain) goes to the block resolver / chain kernels above; INSTRUCT_ZQ_SPEC_RESOLVE=0 disables it.  Diagnostics:
 * out = {sweeps tried, settled, lost (handed on), probes of the last sweep, fail bits of the last sweep (1 window missed, 2 irregular
 * byte, 16 probe list full, 32 consumption check, 64 rounds did not settle), probe rounds of the last sweep, bytes of the accept-bit tables, segments,
 * fail bits of all lost sweeps together, sweeps redone at once with wider windows after a missed window} */
int isg_zq_spec_stats(isg_ctx *ctx, long out[10]);
/* Replay-schedule update_P on the device (instruct_amd/csrc/isg_walk_hip.inc): the start position of every Dirichlet of
 * mcmc.c:846-857 / poly_geno.c:426-434 is resolved by the walk engine (accept bits of every (gamma, stream position) on the
 * whole chip, hierarchical composition of the blocks' offset maps), then all Dirichlets are drawn at once and each is checked
 * to have consumed what the resolution said.  A sweep whose windows were missed is resolved once more with wider windows and, if
 * that fails too, drawn by the sequential host loop instead (same values).  INSTRUCT_P_DEVICE=0 selects the host loop always.
 * Diagnostics: out = {sweeps on the device, sweeps that fell back to the host loop, segments, blocks, table bytes, 1000 sigma (spread
 * of the rejections per gamma the windows assume), 1000 scale (measured / predicted drift), 1000 window half-width in sigma, sweeps
 * resolved a second time with wider windows, runs of the engine that failed} */
int isg_p_device_stats(isg_ctx *ctx, long out[10]);

/* per-kernel device timing with HIP events on the launch stream (bench.py roofline) */
int isg_profile_enable(isg_ctx *ctx, int on);
int isg_profile_count(isg_ctx *ctx);
int isg_profile_get(isg_ctx *ctx, int idx, char *name, int name_cap, double *total_ms, long *launches);
int isg_profile_reset(isg_ctx *ctx);

/* measured device-to-device copy bandwidth (read + written GB/s) of `bytes` bytes with 16-byte loads and stores, `reps` timed
 * repetitions: the ceiling a streaming kernel has on this GPU, reported by bench.py beside the 8 TB/s specification */
int isg_copy_bandwidth(int device, size_t bytes, int reps, double *gbs);

/* host-only exhaustive check of the device's integer/float shortcuts (LCG step without division,
 * quotient by reciprocal + fma, table skip-ahead) against the plain formulas of random.c:19-47 */
int isg_selftest(void);

/* Chains sharded one per GPU (the reference runs them back to back, InStruct.c:182-193): the only exchange is each chain's
 * n stored log-likelihood samples (CONVG.convg_ld, mcmc.c:223-224).  isg_gather_convg: ONE ncclAllGather over RCCL / xGMI
 * on device buffers; all[r * n + k] = sample k of rank r.  id_path: a file every rank can reach (rank 0 publishes the
 * ncclUniqueId there).  Called by instruct_amd/host/mcmc_hip.c when instruct_amd/host/instruct_mgpu.c launched it.
 * isg_gelman_rubin: GelmanRubin (check_converg.c:100-153) on the gathered vector, including the reference's indexing
 * (repperchain = totrep / numchains, check_converg.c:121-137). */
int isg_gather_convg(isg_ctx *ctx, int rank, int world, const char *id_path, const double *mine, int n, double *all);
double isg_gelman_rubin(const double *vec, int numchains, int totrep);

#ifdef __cplusplus
}
#endif
#endif
