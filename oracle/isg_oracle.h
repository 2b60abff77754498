/*
 * isg_oracle.h -- TEST INFRASTRUCTURE.  CPU restatement of the InStruct per-iteration MCMC hot
 * path (reference mcmc.c:182-239 and the functions it calls; random.c samplers).
 *
 * NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load liborc.so.  The HIP product path never calls into it.
 *
 * Parity status: PINNED.  In its reference configuration (math=ORC_MATH_LIBM, accum=ORC_ACC_SEQ,
 * sched=ORC_SCHED_REPLAY) this restatement reproduces, byte for byte, the trajectory files under
 * tests/golden/<case>.golden that oracle/ref_dump.c generated from the REAL reference sweeps
 * (hashes of z, allele counts, generation, freq, qq; RNG seed triples; hex-float scalars after
 * every sweep), and the known-answer vectors of tests/golden/unit_random.golden.
 *
 * Two further switches define the "canonical" configuration that the MI355X kernels are
 * bit-identical to (all state, not only the discrete part):
 *   math  = ORC_MATH_ISG : log/exp/pow/cos from instruct_amd/csrc/isg_math.h instead of glibc
 *   accum = ORC_ACC_EXACT: order-independent fixed-point sums instead of sequential `+=`
 * and one switch selects the RNG position schedule:
 *   sched = ORC_SCHED_REPLAY: one sequential Wichmann-Hill stream, as the reference consumes it
 *   sched = ORC_SCHED_KEYED : same generator and samplers, but every consumer (a Dirichlet, an
 *                             individual's Z draws, ...) seeks to a position that depends only on
 *                             (iteration, phase, index) -- see include/instruct_hip.h "keyed layout".
 */
#ifndef ISG_ORACLE_H
#define ISG_ORACLE_H
#include <stdint.h>

#define ORC_MATH_LIBM 0
#define ORC_MATH_ISG 1
#define ORC_ACC_SEQ 0
#define ORC_ACC_EXACT 1
#define ORC_SCHED_REPLAY 0
#define ORC_SCHED_KEYED 1

typedef struct orc_chain orc_chain;

typedef struct {
	int N, L, P, K, Amax;
	int mode;      /* 1 = admixture (mcmc.c:135), 2 = population selfing rates (mcmc.c:182) */
	int type_freq; /* -y */
	int back_refl; /* -e */
	int print_freq;
	int nstep_check_empty_cluster;
	int math, accum, sched;
} orc_params;

/* geno: int [N][L][P] allele codes (missing: any value, flagged by miss); miss: int [N][L] (1 = missing) */
orc_chain *orc_create(const orc_params *p, const int *allelenum, const int *geno, const int *miss);
void orc_destroy(orc_chain *c);

void orc_set_seeds(orc_chain *c, long s1, long s2, long s3);
void orc_get_seeds(const orc_chain *c, long *s);
uint64_t orc_rng_count(const orc_chain *c); /* uniforms drawn so far */
double orc_ran1(orc_chain *c);

/* chain set-up: initial_chn + generation/selfing-rate init + update_ZQ(init) (mcmc.c:193-206) */
void orc_chain_init(orc_chain *c, const float *initd_row);
void orc_update_P(orc_chain *c);
void orc_update_S_POP(orc_chain *c);
void orc_update_S_IND(orc_chain *c); /* mode 3: per-individual selfing rates (self_rates has N entries then) */
void orc_update_F_IND(orc_chain *c); /* mode 5: per-individual inbreeding coefficients (self_rates has N entries) */
void orc_update_F_POP(orc_chain *c); /* mode 4: update_inbreedcoff_POP (the coefficients live in self_rates) */
void orc_update_Z(orc_chain *c, int init_flag); /* mode 0: update_Z; zz[i] is returned by orc_generation() and mirrored into z */
void orc_update_G(orc_chain *c);
void orc_update_ZQ(orc_chain *c, int init_flag);
void orc_update_alpha(orc_chain *c);
void orc_cal_lkh(orc_chain *c);
void orc_iteration(orc_chain *c); /* the loop body mcmc.c:210-215 */

/* state access (pointers into the chain object) */
int *orc_z(orc_chain *c);            /* [N][L][P] */
double *orc_freq(orc_chain *c);      /* [K][L][Amax] */
double *orc_qq(orc_chain *c);        /* [N][K] */
double *orc_qqnum(orc_chain *c);     /* [N][K] */
int *orc_generation(orc_chain *c);   /* [N] */
double *orc_self_rates(orc_chain *c);/* [K] */
int *orc_state(orc_chain *c);        /* [K] (-e 0) */
double *orc_indvlkh(orc_chain *c);   /* [N] */
double orc_alpha(const orc_chain *c);
void orc_set_alpha(orc_chain *c, double a);
double orc_totallkh(const orc_chain *c);
const int *orc_valid(const orc_chain *c); /* [N][L] */
void orc_count_alleles(orc_chain *c, int *counts /* [K][L][Amax] */);
int orc_error(const orc_chain *c);   /* nonzero once the reference would have called nrerror/exit */

/* running posterior means (CHAIN, mcmc.h:29-53; store_chn mcmc.c:1320-1456) */
typedef struct {
	long steps, step;
	int flag_empty_cluster;
	double totallkh, totallkh2;
	double *indvlkh, *self_rates, *self_rates2, *qq, *qq2, *gen, *gen2, *freq, *freq2;
} orc_result;

/* full chain: mcmc.c:182-239 (mode 2) / 135-179 (mode 1).  convg: ckrep doubles or NULL */
void orc_run_chain(orc_chain *c, const float *initd_row, long update, long burnin, int thinning, int ckrep,
		   double *convg, orc_result *res);
void orc_result_free(orc_result *res);

double orc_gelman_rubin(const double *vec, int numchains, int totrep); /* check_converg.c:100-153 */

/* keyed-schedule geometry (must equal isg_keyed_layout() of the product) */
typedef struct {
	uint64_t SP, SZ, ZI0, B0, offS, offG, offZ, offA, BLK;
} orc_keyed_layout;
void orc_keyed_get_layout(const orc_chain *c, orc_keyed_layout *out);

/* unit-level entry points for known-answer tests */
double orc_rgamma(orc_chain *c, double a);
int orc_rgeom(orc_chain *c, double p);
double orc_rnormal(orc_chain *c, double mean, double sd);
void orc_rdirich(orc_chain *c, const double *alpha, int n, double *out, double add);
int orc_disc_unif(orc_chain *c, double *vec, int n);
double orc_genofreq(orc_chain *c, int a0, int a1, double f0, double f1, int gen);

/* text reader for the small parity cases: -af 0 -lb 0 -a 0 (data_interface.c:91-245, 489-569, 812-846) */
int orc_read_text_diploid(const char *path, int *N, int *L, int **allelenum, int **geno, int **miss);
#endif
