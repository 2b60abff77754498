/*
 * instruct_types.h -- the reference's public struct surface, re-typed for the drop-in sampler.
 *
 * mcmc_updating() receives SEQDATA and INIT BY VALUE and returns CHAIN by value
 * (reference mcmc.h:56), so these definitions must be byte-identical to
 *   SEQDATA  data_interface.h:10-56      INIT   initial.h:9-19      CONVG  check_converg.h:10-18
 *   UPMCMC   mcmc.h:12-27                CHAIN  mcmc.h:29-53
 * The _Static_asserts below are the sizeof/offsetof values printed by a probe compiled against
 * the reference headers (x86-64 SysV, gcc 11).
 */
#ifndef INSTRUCT_TYPES_H
#define INSTRUCT_TYPES_H
#include <stddef.h>

typedef struct seqd {
	int ploid;
	int popnum;
	int locinum;
	int totalsize;

	int datafmt;
	int ***seqdata;      /* [N][L][P] allele codes, -9 = missing */
	int *allelenum;      /* [L] */
	int allelenum_max;
	char ***alleletype;
	int **alleleid;      /* ploidy 4: number of distinct alleles per (i, j) */

	char **indvname;
	char **poptype;
	int *popindx;
	int pop_count;
	int label;
	int popdata;
	int markername_flag;
	char **marker_names;
	int n_extra_col;
	char ***extra_col;

	char *missingdata;
	int missingnum;
	int *missvec;
	int **missindx;      /* [N][L] 1 = missing */

	double siglevel;
	int back_refl;
	int type_freq;
	int nstep_check_empty_cluster;
	int prior_flag;
	int mode;

	double alpha_dpm;
	int print_iter;
	int print_freq;
	int inf_K;
	int distr_fmt;
	int autopoly;
	double max_mem;
} SEQDATA;

typedef struct initialdata {
	float **initd;       /* [chain][K] initial selfing rates */
	char **chn_name;
	int *name_len;
	int chainnum;
	long update;
	long burnin;
	int thinning;
	int popnum;
} INIT;

typedef struct convg {
	double *convg_ld;    /* [n_chain * ckrep] */
	int n_chain;
	int ckrep;
	char *convgfilename;
} CONVG;

typedef struct UPMC {
	int *generation;
	double ***freq;
	double ***freq2;
	int ***z;
	int *zz;
	double **qq;
	double alpha;
	double *inbreed;
	int *state;
	double *self_rates;
	double totallkh;
	double *indvlkh;
	int ***geno;
} UPMCMC;

typedef struct MC {
	long steps;
	long step;
	int name_len;
	char *chn_name;
	int flag_empty_cluster;

	double totallkh;
	double *indvlkh;
	double *self_rates;
	double **qq;
	double *inbreed;
	double *gen;
	long **z;
	double ***freq;

	double totallkh2;
	double *self_rates2;
	double **qq2;
	double *inbreed2;
	double *gen2;
	double ***freq2;
} CHAIN;

_Static_assert(sizeof(SEQDATA) == 232, "SEQDATA");
_Static_assert(offsetof(SEQDATA, seqdata) == 24, "SEQDATA.seqdata");
_Static_assert(offsetof(SEQDATA, allelenum) == 32, "SEQDATA.allelenum");
_Static_assert(offsetof(SEQDATA, allelenum_max) == 40, "SEQDATA.allelenum_max");
_Static_assert(offsetof(SEQDATA, alleleid) == 56, "SEQDATA.alleleid");
_Static_assert(offsetof(SEQDATA, missindx) == 152, "SEQDATA.missindx");
_Static_assert(offsetof(SEQDATA, back_refl) == 168, "SEQDATA.back_refl");
_Static_assert(offsetof(SEQDATA, type_freq) == 172, "SEQDATA.type_freq");
_Static_assert(offsetof(SEQDATA, nstep_check_empty_cluster) == 176, "SEQDATA.nstep_check_empty_cluster");
_Static_assert(offsetof(SEQDATA, mode) == 184, "SEQDATA.mode");
_Static_assert(offsetof(SEQDATA, print_iter) == 200, "SEQDATA.print_iter");
_Static_assert(offsetof(SEQDATA, print_freq) == 204, "SEQDATA.print_freq");
_Static_assert(offsetof(SEQDATA, autopoly) == 216, "SEQDATA.autopoly");
_Static_assert(offsetof(SEQDATA, max_mem) == 224, "SEQDATA.max_mem");
_Static_assert(sizeof(INIT) == 56, "INIT");
_Static_assert(offsetof(INIT, name_len) == 16, "INIT.name_len");
_Static_assert(offsetof(INIT, update) == 32, "INIT.update");
_Static_assert(offsetof(INIT, burnin) == 40, "INIT.burnin");
_Static_assert(offsetof(INIT, thinning) == 48, "INIT.thinning");
_Static_assert(sizeof(CONVG) == 24, "CONVG");
_Static_assert(offsetof(CONVG, ckrep) == 12, "CONVG.ckrep");
_Static_assert(sizeof(UPMCMC) == 104, "UPMCMC");
_Static_assert(offsetof(UPMCMC, z) == 24, "UPMCMC.z");
_Static_assert(offsetof(UPMCMC, qq) == 40, "UPMCMC.qq");
_Static_assert(offsetof(UPMCMC, alpha) == 48, "UPMCMC.alpha");
_Static_assert(offsetof(UPMCMC, state) == 64, "UPMCMC.state");
_Static_assert(offsetof(UPMCMC, self_rates) == 72, "UPMCMC.self_rates");
_Static_assert(offsetof(UPMCMC, totallkh) == 80, "UPMCMC.totallkh");
_Static_assert(offsetof(UPMCMC, indvlkh) == 88, "UPMCMC.indvlkh");
_Static_assert(offsetof(UPMCMC, geno) == 96, "UPMCMC.geno");
_Static_assert(sizeof(CHAIN) == 152, "CHAIN");
_Static_assert(offsetof(CHAIN, chn_name) == 24, "CHAIN.chn_name");
_Static_assert(offsetof(CHAIN, flag_empty_cluster) == 32, "CHAIN.flag_empty_cluster");
_Static_assert(offsetof(CHAIN, totallkh) == 40, "CHAIN.totallkh");
_Static_assert(offsetof(CHAIN, qq) == 64, "CHAIN.qq");
_Static_assert(offsetof(CHAIN, gen) == 80, "CHAIN.gen");
_Static_assert(offsetof(CHAIN, freq) == 96, "CHAIN.freq");
_Static_assert(offsetof(CHAIN, totallkh2) == 104, "CHAIN.totallkh2");
_Static_assert(offsetof(CHAIN, qq2) == 120, "CHAIN.qq2");
_Static_assert(offsetof(CHAIN, gen2) == 136, "CHAIN.gen2");
_Static_assert(offsetof(CHAIN, freq2) == 144, "CHAIN.freq2");

/* symbols the host program provides (reference nrutil.h:46-80, random.h:14-16) */
void nrerror(char error_text[]);
int *ivector(long nl, long nh);
char *cvector(long nl, long nh);
double *dvector(long nl, long nh);
double **dmatrix(long nrl, long nrh, long ncl, long nch);
long **lmatrix(long nrl, long nrh, long ncl, long nch);
double ***d3tensor(long nrl, long nrh, long ncl, long nch, long ndl, long ndh);
int ***i3tensor(long nrl, long nrh, long ncl, long nch, long ndl, long ndh);
void free_ivector(int *v, long nl, long nh);
void free_cvector(char *v, long nl, long nh);
void free_dvector(double *v, long nl, long nh);
void free_dmatrix(double **m, long nrl, long nrh, long ncl, long nch);
void free_lmatrix(long **m, long nrl, long nrh, long ncl, long nch);
void free_d3tensor(double ***t, long nrl, long nrh, long ncl, long nch, long ndl, long ndh);
void free_i3tensor(int ***t, long nrl, long nrh, long ncl, long nch, long ndl, long ndh);
void setseeds(int sd1, int sd2, int sd3);
/* printseeds(FILE *) is declared where <stdio.h> is included */

/* the drop-in surface (reference mcmc.h:56-69) */
CHAIN mcmc_updating(SEQDATA data, INIT initial, int chn, CONVG *cvg);
void free_chain(CHAIN *chain, SEQDATA data);
int chcksame(int *pop, int num);
double genofreq_inbreedcoff(int *seqdata, double *freq, double inbreed, int ploid);
double dgeom(double self, int gen);
void print_info(UPMCMC *ptr, SEQDATA data, int step, int maxstep);
double adpt_indp(int *stat_tmp, int stat);
double hastings_stat(int *tmp, int *prev, int num);
int dt_stat(double num);
void allocate_node(UPMCMC **ptr, SEQDATA data);
void free_node(UPMCMC *ptr, SEQDATA data);
void allocate_chn(CHAIN *chain, SEQDATA data);
void store_chn(CHAIN *mchain, UPMCMC *ptr, SEQDATA data);
int check_empty_cluster(UPMCMC *ptr, SEQDATA data);
#endif
