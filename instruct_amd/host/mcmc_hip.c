/*
 * mcmc_hip.c -- drop-in replacement for the reference's mcmc.c: same exported symbols
 * (mcmc.h:56-69), same struct surface (instruct_types.h), with the per-iteration sweeps of the
 * diploid admixture / population-selfing samplers (mcmc.c:135-239) running on an MI355X through
 * the C ABI of include/instruct_hip.h.  Plain C; links against the host program's own
 * nrutil / random objects exactly like the reference's mcmc.o does.
 *
 * What stays on the host, as in the reference driver: chain bookkeeping (names, burn-in /
 * thinning schedule, CHAIN running means via store_chn, CONVG samples, the empty-cluster check,
 * progress printing).  The stream of random numbers is shared with the host program's random.c:
 * its seed triple is read at entry (printseeds) and written back at exit (setseeds), so draws
 * made by the driver before and after a chain (read_init, the next chain) line up with the
 * reference run.
 *
 * Environment: INSTRUCT_GPU_RNG=replay|keyed (default replay), INSTRUCT_DEVICE=<ordinal>.
 * Set by the multi-GPU launcher (instruct_mgpu.c), one process per chain: INSTRUCT_MGPU_RANK, INSTRUCT_MGPU_WORLD,
 * INSTRUCT_MGPU_DIR (rendezvous directory), INSTRUCT_MGPU_GATHER=rccl|file -- see mgpu_exchange() below.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "instruct_types.h"
#include "instruct_hip.h"

void printseeds(FILE *fp); /* random.c:60-63 */
double ran1(void);         /* random.c:34-47 */

#define MIN2(X, Y) (((X) > (Y)) ? (Y) : (X))

/* ------------------------------------------------------------------ small exported helpers */
int chcksame(int *pop, int num) /* mcmc.c:1658-1667: 1 if any element differs from the first */
{
	int i, flag = 0;
	for (i = 1; i < num; i++)
		if (pop[i] != pop[0]) flag = 1;
	return flag;
}

double genofreq_inbreedcoff(int *seqdata, double *freq, double inbreed, int ploid) /* mcmc.c:1707-1723 */
{
	if (chcksame(seqdata, ploid) == 0) return pow(freq[0], (double)ploid) * (1 - inbreed) + freq[0] * inbreed;
	return 2 * freq[0] * freq[1] * (1 - inbreed);
}

double dgeom(double self, int gen) { return pow(self, (double)(gen - 1)) * (1 - self); } /* mcmc.c:1596-1604 */

int dt_stat(double num) /* mcmc.c:1524-1546 */
{
	const double eps = 0.001;
	if (num <= 0.000 + eps && num >= 0.000 - eps) return 0;
	if (num >= 1.000 - eps && num <= 1.000 + eps) return 2;
	if (num >= 0.0 + eps && num < 1.000 - eps) return 1;
	fprintf(stdout, "ERROR: The value of selfing rate or inbreeding coefficient %f is beyond [0,1]!\n", num);
	exit(1);
}

double adpt_indp(int *stat_tmp, int stat) /* mcmc.c:1461-1520 (host stream: used by poly_geno.c, DPMM.c) */
{
	double tmp = 0, tt;
	if (stat == 0) {
		if (ran1() < 0.50) { tmp = 0.000; *stat_tmp = 0; }
		else { tmp = ran1(); *stat_tmp = 1; }
	} else if (stat == 2) {
		if (ran1() < 0.5) { tmp = 1.000; *stat_tmp = 2; }
		else { tmp = ran1(); *stat_tmp = 1; }
	} else if (stat == 1) {
		tt = ran1();
		if (tt <= 0.05) { tmp = 0.0000; *stat_tmp = 0; }
		else if (tt >= 0.95) { tmp = 1.000; *stat_tmp = 2; }
		else { tmp = ran1(); *stat_tmp = 1; }
	} else {
		fprintf(stdout, "ERROR: State %d is beyond 0, 1 and 2!\n", stat);
		exit(1);
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
double hastings_stat(int *tmp, int *prev, int num) /* mcmc.c:1550-1563 */
{
	int i;
	double t = 1.0;
	for (i = 0; i < num; i++) t *= q_trans(prev[i], tmp[i]) / q_trans(tmp[i], prev[i]);
	return t;
}

/* which UPMCMC / CHAIN members exist for a (ploidy, mode): the case analysis of mcmc.c:506-642 */
static int has_zq(SEQDATA d) { return (d.ploid == 2 && d.mode != 0) || d.ploid == 4; }
static int has_gen(SEQDATA d) { return d.ploid == 2 && (d.mode == 2 || d.mode == 3); }
static int has_state(SEQDATA d) { return d.back_refl == 0 && ((d.ploid == 2 && (d.mode == 2 || d.mode == 4)) || d.ploid == 4); }
/* length of the selfing-rate / inbreeding vector, 0 if the mode has none; *inbreed = 1 for modes 4, 5 */
static long rate_len(SEQDATA d, int *inbreed)
{
	*inbreed = 0;
	if (d.ploid == 4) return d.popnum;
	if (d.ploid != 2) return 0;
	switch (d.mode) {
	case 2: return d.popnum;
	case 3: return d.totalsize;
	case 4: *inbreed = 1; return d.popnum;
	case 5: *inbreed = 1; return d.totalsize;
	}
	return 0;
}

void allocate_node(UPMCMC **ptr, SEQDATA data) /* mcmc.c:506-546 */
{
	UPMCMC *p = (UPMCMC *)malloc(sizeof(UPMCMC));
	int inb;
	long n = rate_len(data, &inb);
	if (p == NULL) nrerror("Allocation failure in ptr");
	*ptr = p;
	if (n > 0) {
		if (inb) p->inbreed = dvector(0, n - 1);
		else p->self_rates = dvector(0, n - 1);
	}
	if (has_state(data)) p->state = ivector(0, data.popnum - 1);
	if (has_gen(data)) p->generation = ivector(0, data.totalsize - 1);
	p->indvlkh = dvector(0, data.totalsize - 1);
	p->freq = d3tensor(0, data.popnum - 1, 0, data.locinum - 1, 0, data.allelenum_max - 1);
	if (has_zq(data)) {
		p->z = i3tensor(0, data.totalsize - 1, 0, data.locinum - 1, 0, data.ploid - 1);
		p->qq = dmatrix(0, data.totalsize - 1, 0, data.popnum - 1);
	} else if (data.ploid == 2 && data.mode == 0) {
		p->zz = ivector(0, data.totalsize - 1);
	}
	if (data.ploid == 4) {
		if (data.autopoly == 0) p->freq2 = d3tensor(0, data.popnum - 1, 0, data.locinum - 1, 0, data.allelenum_max - 1);
		p->geno = i3tensor(0, data.totalsize - 1, 0, data.locinum - 1, 0, data.ploid - 1);
	}
}

void free_node(UPMCMC *p, SEQDATA data) /* mcmc.c:549-585 */
{
	int inb;
	long n = rate_len(data, &inb);
	if (n > 0) free_dvector(inb ? p->inbreed : p->self_rates, 0, n - 1);
	if (has_state(data)) free_ivector(p->state, 0, data.popnum - 1);
	if (has_gen(data)) free_ivector(p->generation, 0, data.totalsize - 1);
	free_dvector(p->indvlkh, 0, data.totalsize - 1);
	free_d3tensor(p->freq, 0, data.popnum - 1, 0, data.locinum - 1, 0, data.allelenum_max - 1);
	if (has_zq(data)) {
		free_i3tensor(p->z, 0, data.totalsize - 1, 0, data.locinum - 1, 0, data.ploid - 1);
		free_dmatrix(p->qq, 0, data.totalsize - 1, 0, data.popnum - 1);
	} else if (data.ploid == 2 && data.mode == 0) {
		free_ivector(p->zz, 0, data.totalsize - 1);
	}
	if (data.ploid == 4) {
		if (data.autopoly == 0) free_d3tensor(p->freq2, 0, data.popnum - 1, 0, data.locinum - 1, 0, data.allelenum_max - 1);
		free_i3tensor(p->geno, 0, data.totalsize - 1, 0, data.locinum - 1, 0, data.ploid - 1);
	}
	free(p);
}

static void fill(double *v, long n, double x)
{
	long i;
	for (i = 0; i < n; i++) v[i] = x;
}

void allocate_chn(CHAIN *c, SEQDATA data) /* mcmc.c:588-642 + initialize_chn :644-738: every running mean starts at 1 */
{
	int inb, i, j, k;
	long n = rate_len(data, &inb);
	c->indvlkh = dvector(0, data.totalsize - 1);
	fill(c->indvlkh, data.totalsize, 1);
	c->step = 0;
	c->totallkh = 1;
	c->totallkh2 = 1;
	if (has_zq(data)) {
		c->qq = dmatrix(0, data.totalsize - 1, 0, data.popnum - 1);
		c->qq2 = dmatrix(0, data.totalsize - 1, 0, data.popnum - 1);
		for (i = 0; i < data.totalsize; i++) {
			fill(c->qq[i], data.popnum, 1);
			fill(c->qq2[i], data.popnum, 1);
		}
	} else if (data.ploid == 2 && data.mode == 0) {
		c->z = lmatrix(0, data.totalsize - 1, 0, data.popnum - 1);
		for (i = 0; i < data.totalsize; i++)
			for (j = 0; j < data.popnum; j++) c->z[i][j] = 0;
	}
	if (n > 0) {
		double *a = dvector(0, n - 1), *b = dvector(0, n - 1);
		fill(a, n, 1);
		fill(b, n, 1);
		if (inb) { c->inbreed = a; c->inbreed2 = b; }
		else { c->self_rates = a; c->self_rates2 = b; }
	}
	if (has_gen(data)) {
		c->gen = dvector(0, data.totalsize - 1);
		c->gen2 = dvector(0, data.totalsize - 1);
		fill(c->gen, data.totalsize, 1);
		fill(c->gen2, data.totalsize, 1);
	}
	if (data.print_freq == 1 && data.ploid == 2) {
		c->freq = d3tensor(0, data.popnum - 1, 0, data.locinum - 1, 0, data.allelenum_max - 1);
		c->freq2 = d3tensor(0, data.popnum - 1, 0, data.locinum - 1, 0, data.allelenum_max - 1);
		for (j = 0; j < data.popnum; j++)
			for (i = 0; i < data.locinum; i++)
				for (k = 0; k < data.allelenum[i]; k++) {
					c->freq[j][i][k] = 1;
					c->freq2[j][i][k] = 1;
				}
	}
}

/* the multiplicative running mean of mcmc.c:1327-1332 (and every block below it) */
static void runmean(double *m, double x, long step)
{
	if (*m != 0) *m = *m * ((step + x / *m) / (1 + step));
	else *m = x / (1 + step);
}

/* the part of store_chn that stays on the host when the O(N K) / O(K L A) running means are kept on the device
 * (isg_store_step): the two scalars and the rate vector */
static void store_chn_small(CHAIN *c, UPMCMC *p, SEQDATA data)
{
	int inb, j;
	long n = rate_len(data, &inb), s = c->step;
	runmean(&c->totallkh, p->totallkh, s);
	runmean(&c->totallkh2, p->totallkh * p->totallkh, s);
	for (j = 0; j < n; j++) {
		double x = inb ? p->inbreed[j] : p->self_rates[j];
		runmean(inb ? &c->inbreed[j] : &c->self_rates[j], x, s);
		runmean(inb ? &c->inbreed2[j] : &c->self_rates2[j], x * x, s);
	}
	c->step++;
}

void store_chn(CHAIN *c, UPMCMC *p, SEQDATA data) /* mcmc.c:1320-1456 */
{
	int inb, i, j, k;
	long n = rate_len(data, &inb), s = c->step;
	runmean(&c->totallkh, p->totallkh, s);
	runmean(&c->totallkh2, p->totallkh * p->totallkh, s);
	for (i = 0; i < data.totalsize; i++) runmean(&c->indvlkh[i], p->indvlkh[i], s);
	if (has_zq(data)) {
		for (i = 0; i < data.totalsize; i++)
			for (j = 0; j < data.popnum; j++) {
				runmean(&c->qq[i][j], p->qq[i][j], s);
				runmean(&c->qq2[i][j], p->qq[i][j] * p->qq[i][j], s);
			}
	} else if (data.ploid == 2 && data.mode == 0) {
		for (i = 0; i < data.totalsize; i++) c->z[i][p->zz[i]] += 1;
	}
	for (j = 0; j < n; j++) {
		double x = inb ? p->inbreed[j] : p->self_rates[j];
		runmean(inb ? &c->inbreed[j] : &c->self_rates[j], x, s);
		runmean(inb ? &c->inbreed2[j] : &c->self_rates2[j], x * x, s);
	}
	if (has_gen(data))
		for (i = 0; i < data.totalsize; i++) {
			runmean(&c->gen[i], p->generation[i], s);
			runmean(&c->gen2[i], p->generation[i] * p->generation[i], s);
		}
	if (data.print_freq == 1 && data.ploid == 2)
		for (j = 0; j < data.popnum; j++)
			for (i = 0; i < data.locinum; i++)
				for (k = 0; k < data.allelenum[i]; k++) {
					runmean(&c->freq[j][i][k], p->freq[j][i][k], s);
					runmean(&c->freq2[j][i][k], p->freq[j][i][k] * p->freq[j][i][k], s);
				}
	c->step++;
}

void free_chain(CHAIN *c, SEQDATA data) /* mcmc.c:740-796 */
{
	int inb;
	long n = rate_len(data, &inb);
	free_cvector(c->chn_name, 0, c->name_len - 1);
	free_dvector(c->indvlkh, 0, data.totalsize - 1);
	if (has_zq(data)) {
		free_dmatrix(c->qq, 0, data.totalsize - 1, 0, data.popnum - 1);
		free_dmatrix(c->qq2, 0, data.totalsize - 1, 0, data.popnum - 1);
	} else if (data.ploid == 2 && data.mode == 0) {
		free_lmatrix(c->z, 0, data.totalsize - 1, 0, data.popnum - 1);
	}
	if (n > 0) {
		free_dvector(inb ? c->inbreed : c->self_rates, 0, n - 1);
		free_dvector(inb ? c->inbreed2 : c->self_rates2, 0, n - 1);
	}
	if (has_gen(data)) {
		free_dvector(c->gen, 0, data.totalsize - 1);
		free_dvector(c->gen2, 0, data.totalsize - 1);
	}
	if (data.print_freq == 1 && data.ploid == 2) {
		free_d3tensor(c->freq, 0, data.popnum - 1, 0, data.locinum - 1, 0, data.allelenum_max - 1);
		free_d3tensor(c->freq2, 0, data.popnum - 1, 0, data.locinum - 1, 0, data.allelenum_max - 1);
	}
}

int check_empty_cluster(UPMCMC *p, SEQDATA data) /* mcmc.c:1944-1974 */
{
	int j, k;
	for (k = 0; k < data.popnum; k++) {
		double sum = 0;
		for (j = 0; j < data.totalsize; j++) sum += p->qq[j][k];
		if (sum < 0.01) return 1;
	}
	return 0;
}

void print_info(UPMCMC *p, SEQDATA data, int step, int maxstep) /* mcmc.c:1267-1316 */
{
	int i, s = maxstep / 100;
	if (step % s != 0) return;
	fprintf(stdout, "\nStep=%d\tlog_likelihood=%f\n", step + 1, p->totallkh);
	if (data.mode == 2 || data.ploid == 4 || data.mode == 4) {
		double *v = (data.mode == 4 && data.ploid != 4) ? p->inbreed : p->self_rates;
		const char *fmt = (data.mode == 4 && data.ploid != 4) ? "f_%d=%f" : "s_%d=%f";
		for (i = 0; i < data.popnum; i++) {
			fprintf(stdout, fmt, i, v[i]);
			if (data.back_refl == 0) fprintf(stdout, " st_%d=%d", i, p->state[i]);
			if (i < data.popnum - 1) fprintf(stdout, " ");
		}
		fprintf(stdout, "\n");
	}
	if (data.mode == 3 || data.mode == 5) {
		for (i = 0; i < data.totalsize; i++)
			fprintf(stdout, data.mode == 3 ? "s_%d=%f " : "f_%d=%f ", i, data.mode == 3 ? p->self_rates[i] : p->inbreed[i]);
		fprintf(stdout, "\n");
	}
}

/* ------------------------------------------------------------------ the MI355X chain driver */
/* A NaN of the reference run is always x86's default NaN -- sign bit set, printed "-nan" -- because it can only come from an
 * invalid operation on finite inputs (inf - inf when a cluster's rate sits at exactly 0 or 1, ploidy 4 with -e 0) and is
 * propagated unchanged from there.  The device's default NaN has the sign bit clear: give it the reference's sign before the
 * host code prints it or folds it into the CHAIN means (print_info mcmc.c:1275, result_analysis.c:399-412). */
static double ref_nan(double x) { return isnan(x) ? copysign(x, -1.0) : x; }

static void hip_fail(const char *where)
{
	static char msg[512];
	snprintf(msg, sizeof(msg), "%s: %s", where, isg_last_error());
	nrerror(msg);
}

static void read_host_seeds(long s[3]) /* the host program's random.c keeps its state file-static */
{
	char *buf = NULL;
	size_t len = 0;
	FILE *f = open_memstream(&buf, &len);
	if (!f) nrerror("open_memstream failed");
	printseeds(f);
	fclose(f);
	if (sscanf(buf, "%ld %ld %ld", &s[0], &s[1], &s[2]) != 3) nrerror("cannot read the RNG seeds of the host program");
	free(buf);
}

/*
 * One chain per process and GPU (instruct_mgpu.c): the chain's stored log-likelihood samples (CONVG.convg_ld of chain 0,
 * mcmc.c:223-224) are what the Gelman-Rubin check needs from every chain (check_converg.c:44-91).  Each rank leaves its
 * samples as raw doubles in <dir>/convg.<rank>.bin; with INSTRUCT_MGPU_GATHER=rccl the ranks also exchange them with one
 * ncclAllGather over RCCL / xGMI (isg_gather_convg) and rank 0 leaves the gathered vector in <dir>/convg_all.bin.  The
 * launcher evaluates the statistic and assembles the result file.  A chain discarded for an empty cluster
 * (InStruct.c:185-190) does not take part: its re-run does.
 */
static void mgpu_exchange(isg_ctx *ctx, const CONVG *cvg, int chn)
{
	const char *er = getenv("INSTRUCT_MGPU_RANK"), *ew = getenv("INSTRUCT_MGPU_WORLD"), *dir = getenv("INSTRUCT_MGPU_DIR"), *eg = getenv("INSTRUCT_MGPU_GATHER");
	char path[4096];
	const double *mine;
	int rank, world, n;
	FILE *f;
	if (!er || !ew || !dir || cvg == NULL) return;
	rank = atoi(er);
	world = atoi(ew);
	n = cvg->ckrep;
	mine = cvg->convg_ld + (size_t)chn * n;
	snprintf(path, sizeof(path), "%s/convg.%d.bin", dir, rank);
	if ((f = fopen(path, "wb")) == NULL || fwrite(mine, sizeof(double), (size_t)n, f) != (size_t)n) nrerror("cannot write the log-likelihood samples for the multi-GPU launcher");
	fclose(f);
	if (eg && strcmp(eg, "rccl") == 0) {
		double *all = (double *)malloc(sizeof(double) * (size_t)n * world);
		if (!all) nrerror("allocation failure in mgpu_exchange");
		snprintf(path, sizeof(path), "%s/nccl_id", dir);
		if (isg_gather_convg(ctx, rank, world, path, mine, n, all)) hip_fail("isg_gather_convg");
		if (rank == 0) {
			char tmp[4200];
			snprintf(path, sizeof(path), "%s/convg_all.bin", dir);
			snprintf(tmp, sizeof(tmp), "%s.tmp", path);
			if ((f = fopen(tmp, "wb")) == NULL || fwrite(all, sizeof(double), (size_t)n * world, f) != (size_t)n * world) nrerror("cannot write the gathered log-likelihood samples");
			fclose(f);
			if (rename(tmp, path) != 0) nrerror("cannot publish the gathered log-likelihood samples");
		}
		free(all);
	}
}

/* one cached device context per process: the driver runs chains back to back on the same data */
static isg_ctx *g_ctx;
static int ***g_ctx_key;
static int g_ctx_K, g_ctx_mode;

static isg_ctx *get_ctx(SEQDATA d)
{
	isg_config cfg;
	int32_t *geno, *miss;
	const char *e;
	long i, j, k, N = d.totalsize, L = d.locinum;
	if (g_ctx && g_ctx_key == d.seqdata && g_ctx_K == d.popnum && g_ctx_mode == d.mode) return g_ctx;
	if (g_ctx) isg_ctx_destroy(g_ctx);
	g_ctx = NULL;
	memset(&cfg, 0, sizeof(cfg));
	cfg.N = d.totalsize; cfg.L = d.locinum; cfg.P = d.ploid; cfg.K = d.popnum;
	cfg.mode = d.mode; cfg.type_freq = d.type_freq; cfg.back_refl = d.back_refl;
	e = getenv("INSTRUCT_GPU_RNG");
	cfg.rng_sched = (e && strcmp(e, "keyed") == 0) ? ISG_SCHED_KEYED : ISG_SCHED_REPLAY;
	e = getenv("INSTRUCT_DEVICE");
	cfg.device = e ? atoi(e) : 0;
	/* flatten through the pointer tables: after monomorphic loci were dropped the rows of the
	 * i3tensor are no longer densely packed (data_interface.c:493, 551) */
	geno = (int32_t *)malloc(sizeof(int32_t) * N * L * d.ploid);
	miss = (int32_t *)malloc(sizeof(int32_t) * N * L);
	if (!geno || !miss) nrerror("allocation failure while packing genotypes");
	if (d.ploid == 4) { /* distinct observed alleles + their number (transform_data2, data_interface.c:617-640) */
		for (i = 0; i < N; i++)
			for (j = 0; j < L; j++) {
				miss[i * L + j] = d.alleleid[i][j];
				for (k = 0; k < 4; k++) geno[(i * L + j) * 4 + k] = k < d.alleleid[i][j] ? d.seqdata[i][j][k] : -1;
			}
		cfg.rng_sched = ISG_SCHED_REPLAY;
		cfg.reserved[0] = (d.autopoly == 0); /* allotetraploid: two subgenomes (freq, freq2) */
		if (isg_ctx_create_poly(&cfg, d.allelenum, geno, miss, &g_ctx)) hip_fail("isg_ctx_create_poly");
	} else {
		for (i = 0; i < N; i++)
			for (j = 0; j < L; j++) {
				miss[i * L + j] = d.missindx[i][j];
				for (k = 0; k < d.ploid; k++) geno[(i * L + j) * d.ploid + k] = d.seqdata[i][j][k];
			}
		if (isg_ctx_create(&cfg, d.allelenum, geno, miss, &g_ctx)) hip_fail("isg_ctx_create");
	}
	free(geno);
	free(miss);
	g_ctx_key = d.seqdata;
	g_ctx_K = d.popnum;
	g_ctx_mode = d.mode;
	return g_ctx;
}

/* mcmc_POP_admixture (mcmc.c:135-179), mcmc_POP_selfing (mcmc.c:182-239) and, for ploidy 4 with -ap 1,
 * mcmc_POP_tetra_selfing (poly_geno.c:75-140; that driver has no free_space() epilogue) */
static CHAIN mcmc_hip_chain(SEQDATA data, INIT initial, int chn, CONVG *cvg)
{
	isg_ctx *ctx = get_ctx(data);
	CHAIN mchain;
	UPMCMC node; /* host view of the sampler state for store_chn / print_info / check_empty_cluster */
	long seeds[3], cnt_step = 0, step;
	const int N = data.totalsize, K = data.popnum, L = data.locinum, A = data.allelenum_max;
	double *qqflat = (double *)malloc(sizeof(double) * (size_t)N * K);
	double *freqflat = data.print_freq == 1 ? (double *)malloc(sizeof(double) * (size_t)K * L * A) : NULL;
	const int tetra = (data.ploid == 4), inbr = (data.ploid == 2 && data.mode == 4), indv = (data.ploid == 2 && data.mode == 3),
		  finb = (data.ploid == 2 && data.mode == 5);
	int i, j, k, stored_on_device = 0;

	memset(&mchain, 0, sizeof(mchain));
	memset(&node, 0, sizeof(node));
	/* initial_chn (mcmc.c:471-487) */
	mchain.name_len = initial.name_len[chn];
	mchain.chn_name = cvector(0, mchain.name_len - 1);
	for (j = 0; j < mchain.name_len; j++) mchain.chn_name[j] = initial.chn_name[chn][j];
	mchain.steps = (int)((initial.update - initial.burnin) / initial.thinning);
	fprintf(stdout, "\n\n%s Starts:\n", mchain.chn_name);

	node.qq = dmatrix(0, N - 1, 0, K - 1);
	node.indvlkh = dvector(0, N - 1);
	if (data.mode == 2 || tetra) {
		node.self_rates = dvector(0, K - 1);
		node.generation = ivector(0, N - 1);
		node.state = ivector(0, K - 1);
	}
	if (indv) { /* mode 3: one selfing rate and one generation per individual (allocate_node, mcmc.c:515-520) */
		node.self_rates = dvector(0, N - 1);
		node.generation = ivector(0, N - 1);
	}
	if (finb) node.inbreed = dvector(0, N - 1); /* mode 5: one coefficient per individual */
	if (inbr) { /* mode 4: UPMCMC.inbreed (allocate_node, mcmc.c:524-530) */
		node.inbreed = dvector(0, K - 1);
		node.state = ivector(0, K - 1);
	}
	if (data.print_freq == 1) node.freq = d3tensor(0, K - 1, 0, L - 1, 0, A - 1);

	read_host_seeds(seeds);
	if (isg_set_seeds(ctx, seeds[0], seeds[1], seeds[2])) hip_fail("isg_set_seeds");
	if (isg_chain_init(ctx, initial.initd[chn])) hip_fail("isg_chain_init");

	for (step = 0; step < initial.update; step++) {
		int stored = (step >= initial.burnin && (step + 1 - initial.burnin) % initial.thinning == 0);
		int want_state;
		if (isg_iteration(ctx)) hip_fail("isg_iteration");
		want_state = stored || data.print_iter == 1 || cnt_step == data.nstep_check_empty_cluster;
		if (want_state) {
			if (isg_get_totallkh(ctx, &node.totallkh)) hip_fail("isg_get_totallkh");
			node.totallkh = ref_nan(node.totallkh);
			/* qq itself is only looked at by print_info and check_empty_cluster; its running means are kept on the device */
			if (data.print_iter == 1 || cnt_step == data.nstep_check_empty_cluster || (stored && cnt_step + 1 == data.nstep_check_empty_cluster)) {
				isg_get_qq(ctx, qqflat);
				for (i = 0; i < N; i++)
					for (k = 0; k < K; k++) node.qq[i][k] = qqflat[(size_t)i * K + k];
			}
			if (data.mode == 2 || tetra) {
				isg_get_self_rates(ctx, node.self_rates);
				isg_get_state(ctx, node.state);
			}
			if (inbr) {
				isg_get_self_rates(ctx, node.inbreed);
				isg_get_state(ctx, node.state);
			}
			if (indv) isg_get_self_rates(ctx, node.self_rates);
			if (finb) isg_get_self_rates(ctx, node.inbreed);
		}
		if (data.print_iter == 1) print_info(&node, data, step, initial.update);
		if (step == initial.burnin - 1) {
			allocate_chn(&mchain, data);
			if (isg_store_begin(ctx, data.print_freq == 1 && !tetra)) hip_fail("isg_store_begin");
			stored_on_device = 1;
		}
		if (stored) {
			/* store_chn (mcmc.c:1320-1456): qq, qq2, indvlkh, gen, gen2, freq, freq2 on the device, the rest here */
			if (isg_store_step(ctx)) hip_fail("isg_store_step");
			store_chn_small(&mchain, &node, data);
			if (cnt_step < cvg->ckrep) cvg->convg_ld[chn * cvg->ckrep + cnt_step] = node.totallkh;
			cnt_step++;
		}
		if (cnt_step == data.nstep_check_empty_cluster) {
			if ((mchain.flag_empty_cluster = check_empty_cluster(&node, data)) == 1) {
				fprintf(stdout, "Chain %d has an empty cluster, thus discarded!\n", chn + 1);
				break;
			}
		}
	}
	/* free_space (mcmc.c:490-503) */
	if (mchain.flag_empty_cluster == 0 && !tetra) {
		if (cnt_step != mchain.steps) nrerror("The number of iterations attained is not the same as counted");
		fprintf(stdout, "\n\nChain %d is finished running.\n", chn + 1);
	}
	isg_get_seeds(ctx, seeds);
	setseeds((int)seeds[0], (int)seeds[1], (int)seeds[2]);
	if (mchain.flag_empty_cluster == 0) mgpu_exchange(ctx, cvg, chn);
	if (stored_on_device) { /* the running means come back once, into the CHAIN the caller will read */
		const int with_gen = has_gen(data), with_freq = (data.print_freq == 1 && !tetra);
		double *q1 = (double *)malloc(sizeof(double) * (size_t)N * K), *q2 = (double *)malloc(sizeof(double) * (size_t)N * K);
		double *f2 = with_freq ? (double *)malloc(sizeof(double) * (size_t)K * L * A) : NULL;
		long nst = 0;
		if (isg_store_fetch(ctx, q1, q2, mchain.indvlkh, with_gen ? mchain.gen : NULL, with_gen ? mchain.gen2 : NULL, with_freq ? freqflat : NULL, f2, &nst))
			hip_fail("isg_store_fetch");
		if (nst != mchain.step) nrerror("The number of steps stored on the device is not the same as counted");
		for (i = 0; i < N; i++) mchain.indvlkh[i] = ref_nan(mchain.indvlkh[i]);
		for (i = 0; i < N; i++)
			for (k = 0; k < K; k++) {
				mchain.qq[i][k] = q1[(size_t)i * K + k];
				mchain.qq2[i][k] = q2[(size_t)i * K + k];
			}
		if (with_freq)
			for (k = 0; k < K; k++)
				for (j = 0; j < L; j++)
					for (i = 0; i < data.allelenum[j]; i++) {
						mchain.freq[k][j][i] = freqflat[((size_t)k * L + j) * A + i];
						mchain.freq2[k][j][i] = f2[((size_t)k * L + j) * A + i];
					}
		free(q1);
		free(q2);
		free(f2);
	}

	free_dmatrix(node.qq, 0, N - 1, 0, K - 1);
	free_dvector(node.indvlkh, 0, N - 1);
	if (finb) free_dvector(node.inbreed, 0, N - 1);
	if (indv) {
		free_dvector(node.self_rates, 0, N - 1);
		free_ivector(node.generation, 0, N - 1);
	}
	if (inbr) {
		free_dvector(node.inbreed, 0, K - 1);
		free_ivector(node.state, 0, K - 1);
	}
	if (data.mode == 2 || tetra) {
		free_dvector(node.self_rates, 0, K - 1);
		free_ivector(node.generation, 0, N - 1);
		free_ivector(node.state, 0, K - 1);
	}
	if (data.print_freq == 1) free_d3tensor(node.freq, 0, K - 1, 0, L - 1, 0, A - 1);
	free(qqflat);
	free(freqflat);
	return mchain;
}

/* mcmc_POP_no_admixture (mcmc.c:90-132): whole individuals are assigned; no initial_chn / free_space around the loop */
static CHAIN mcmc_hip_chain0(SEQDATA data, INIT initial, int chn, CONVG *cvg)
{
	isg_ctx *ctx = get_ctx(data);
	CHAIN mchain;
	UPMCMC node;
	long seeds[3], cnt_step = 0, step;
	const int N = data.totalsize, K = data.popnum, L = data.locinum, A = data.allelenum_max;
	double *freqflat = data.print_freq == 1 ? (double *)malloc(sizeof(double) * (size_t)K * L * A) : NULL;
	int i, j, k;
	memset(&mchain, 0, sizeof(mchain));
	memset(&node, 0, sizeof(node));
	if (data.print_freq == 1) node.freq = d3tensor(0, K - 1, 0, L - 1, 0, A - 1);
	mchain.name_len = initial.name_len[chn];
	mchain.chn_name = cvector(0, mchain.name_len - 1);
	for (j = 0; j < mchain.name_len; j++) mchain.chn_name[j] = initial.chn_name[chn][j];
	mchain.steps = (int)((initial.update - initial.burnin) / initial.thinning);
	fprintf(stdout, "\n\n%s Starts:\n", mchain.chn_name);
	node.zz = ivector(0, N - 1);
	node.indvlkh = dvector(0, N - 1);
	read_host_seeds(seeds);
	if (isg_set_seeds(ctx, seeds[0], seeds[1], seeds[2])) hip_fail("isg_set_seeds");
	if (isg_chain_init(ctx, initial.initd[chn])) hip_fail("isg_chain_init");
	for (step = 0; step < initial.update; step++) {
		const int stored = (step >= initial.burnin && (step + 1 - initial.burnin) % initial.thinning == 0);
		if (isg_iteration(ctx)) hip_fail("isg_iteration");
		if (stored || data.print_iter == 1) {
			if (isg_get_totallkh(ctx, &node.totallkh)) hip_fail("isg_get_totallkh");
			node.totallkh = ref_nan(node.totallkh);
		}
		if (data.print_iter == 1) print_info(&node, data, step, initial.update);
		if (step == initial.burnin - 1) allocate_chn(&mchain, data);
		if (stored) {
			isg_get_indvlkh(ctx, node.indvlkh);
			isg_get_generation(ctx, node.zz); /* the ABI returns the cluster of each individual in the generation slots */
			if (data.print_freq == 1) {
				if (isg_get_freq(ctx, freqflat)) hip_fail("isg_get_freq");
				for (k = 0; k < K; k++)
					for (j = 0; j < L; j++)
						for (i = 0; i < data.allelenum[j]; i++) node.freq[k][j][i] = freqflat[((size_t)k * L + j) * A + i];
			}
			store_chn(&mchain, &node, data);
			if (cnt_step < cvg->ckrep) cvg->convg_ld[chn * cvg->ckrep + cnt_step] = node.totallkh;
			cnt_step++;
		}
	}
	isg_get_seeds(ctx, seeds);
	setseeds((int)seeds[0], (int)seeds[1], (int)seeds[2]);
	mgpu_exchange(ctx, cvg, chn);
	free_ivector(node.zz, 0, N - 1);
	free_dvector(node.indvlkh, 0, N - 1);
	if (data.print_freq == 1) free_d3tensor(node.freq, 0, K - 1, 0, L - 1, 0, A - 1);
	free(freqflat);
	return mchain;
}

CHAIN mcmc_updating(SEQDATA data, INIT initial, int chn, CONVG *cvg) /* mcmc.c:63-87 */
{
	CHAIN chain;
	memset(&chain, 0, sizeof(chain));
	if (data.ploid == 2 && data.mode == 0) return mcmc_hip_chain0(data, initial, chn, cvg);
	if (data.ploid == 2 && (data.mode == 1 || data.mode == 2 || data.mode == 4 || ((data.mode == 3 || data.mode == 5) && data.prior_flag == 0)))
		return mcmc_hip_chain(data, initial, chn, cvg);
	if (data.ploid == 4) return mcmc_hip_chain(data, initial, chn, cvg); /* -ap 1 autotetraploid, -ap 0 allotetraploid */
	nrerror("this build of the sampler accelerates diploid modes 0, 1, 2, 4 and, with the uniform prior, 3 and 5 (-v 0 .. -v 5, -f 0) and tetraploids (-p 4, -ap 0 / 1); other modes need the reference mcmc.c");
	return chain;
}
