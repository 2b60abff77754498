/*
 * ref_dump.c -- TEST INFRASTRUCTURE (oracle/): golden-trajectory generator driven by the REAL
 * reference code.  Built ONLY in the development container (where /root/reference exists) by
 * oracle/Makefile into oracle/_ref/ref_dump; nothing of the reference is copied into this
 * repository: the reference translation unit is pulled in by absolute path at compile time so
 * that its `static` sweeps (update_P, update_S_POP, update_G, update_ZQ, update_alpha, cal_lkh)
 * can be called one by one and the sampler state dumped in between.
 *
 * The driver below re-states only the ~30-line chain loop of mcmc_POP_selfing
 * (reference mcmc.c:182-239) / mcmc_POP_admixture (mcmc.c:135-179), calling the reference's
 * own functions for every step, and main()'s set-up order (InStruct.c:152-193:
 * param_decomp -> read_data -> read_init -> per chain mcmc_updating).
 *
 * usage: ref_dump data.txt out.golden K N L u b t c e y r j s1 s2 s3 mode pf detail_every
 */
#include "/root/reference/mcmc.c"
#include "dump_fmt.h"

double ref_GelmanRubin(double *vec, int numchains, int totrep); /* ref_gr.c */

static FILE *G;
static dump_dims D;
static int *vflat, *zflat, *gflat, *cflat;
static double *fflat, *qflat, *nflat;

static void seeds_line(void) { fprintf(G, " seeds="); printseeds(G); }

static void flatten(UPMCMC *p, SEQDATA d, double **qqnum)
{
	long i, j, k;
	for (i = 0; i < D.N; i++)
		for (j = 0; j < D.L; j++)
			for (k = 0; k < D.P; k++)
				zflat[(i * D.L + j) * D.P + k] = p->z[i][j][k];
	for (k = 0; k < D.K; k++)
		for (j = 0; j < D.L; j++)
			for (i = 0; i < D.Amax; i++)
				fflat[(k * D.L + j) * D.Amax + i] = (i < d.allelenum[j]) ? p->freq[k][j][i] : 0.0;
	for (i = 0; i < D.N; i++)
		for (k = 0; k < D.K; k++) {
			qflat[i * D.K + k] = p->qq[i][k];
			nflat[i * D.K + k] = qqnum[i][k];
		}
}

static void detail(UPMCMC *p, SEQDATA d, long step)
{
	int i, j, k;
	for (i = 0; i < D.N && i < 4; i++) {
		fprintf(G, "it %ld qq %d", step, i);
		for (k = 0; k < D.K; k++) fprintf(G, " %a", p->qq[i][k]);
		fprintf(G, "\n");
	}
	for (k = 0; k < D.K; k++)
		for (j = 0; j < D.L && j < 4; j++) {
			fprintf(G, "it %ld freq %d %d", step, k, j);
			for (i = 0; i < d.allelenum[j]; i++) fprintf(G, " %a", p->freq[k][j][i]);
			fprintf(G, "\n");
		}
}

static void dump_chain(CHAIN *c, SEQDATA d)
{
	int i, k, j;
	fprintf(G, "chain steps=%ld step=%ld flag_empty=%d totallkh=%a totallkh2=%a\n", c->steps, c->step,
		c->flag_empty_cluster, c->totallkh, c->totallkh2);
	dump_vec(G, "chain indvlkh", c->indvlkh, D.N);
	if (d.mode == 3) {
		dump_vec(G, "chain self_rates", c->self_rates, D.N);
		dump_vec(G, "chain self_rates2", c->self_rates2, D.N);
		dump_vec(G, "chain gen", c->gen, D.N);
		dump_vec(G, "chain gen2", c->gen2, D.N);
	}
	if (d.mode == 5) {
		dump_vec(G, "chain self_rates", c->inbreed, D.N);
		dump_vec(G, "chain self_rates2", c->inbreed2, D.N);
	}
	if (d.mode == 4) {
		dump_vec(G, "chain self_rates", c->inbreed, D.K);
		dump_vec(G, "chain self_rates2", c->inbreed2, D.K);
	}
	if (d.mode == 2) {
		dump_vec(G, "chain self_rates", c->self_rates, D.K);
		dump_vec(G, "chain self_rates2", c->self_rates2, D.K);
		dump_vec(G, "chain gen", c->gen, D.N);
		dump_vec(G, "chain gen2", c->gen2, D.N);
	}
	for (i = 0; i < D.N; i++) {
		fprintf(G, "chain qq %d", i);
		for (k = 0; k < D.K; k++) fprintf(G, " %a", c->qq[i][k]);
		for (k = 0; k < D.K; k++) fprintf(G, " %a", c->qq2[i][k]);
		fprintf(G, "\n");
	}
	if (d.print_freq == 1) {
		uint64_t h = fnv_init(), h2 = fnv_init();
		for (k = 0; k < D.K; k++)
			for (j = 0; j < D.L; j++)
				for (i = 0; i < d.allelenum[j]; i++) {
					h = fnv_f64(h, c->freq[k][j][i]);
					h2 = fnv_f64(h2, c->freq2[k][j][i]);
				}
		fprintf(G, "chain hfreq=%016llx hfreq2=%016llx\n", (unsigned long long)h, (unsigned long long)h2);
	}
}

/* the chain loop of mcmc.c:182-239 (mode 2) / mcmc.c:135-179 (mode 1), with dumps between sweeps */
static CHAIN run_chain(SEQDATA data, INIT initial, int chn, CONVG *cvg, int detail_every)
{
	int i;
	long cnt_step = 0, step;
	double **qqnum;
	CHAIN mchain;
	UPMCMC *ptr;

	mchain.flag_empty_cluster = 0; /* reference leaves this uninitialised until the first check */
	initial_chn(&qqnum, data, &ptr, &mchain, initial, chn);
	fprintf(G, "chain %d init alpha=%a", chn, ptr->alpha);
	seeds_line();
	if (data.mode == 2) {
		for (i = 0; i < data.totalsize; i++) {
			ptr->generation[i] = rgeom(ran1());
			if (ptr->generation[i] > 50) ptr->generation[i] = 50;
		}
		for (i = 0; i < data.popnum; i++) {
			ptr->self_rates[i] = initial.initd[chn][i];
			if (data.back_refl == 0) ptr->state[i] = dt_stat(ptr->self_rates[i]);
		}
		fprintf(G, "chain %d geninit hgen=%016llx", chn, (unsigned long long)hash_i32v(ptr->generation, D.N));
		seeds_line();
	}
	if (data.mode == 3) { /* mcmc_INDV_selfing, mcmc.c:324-331 (prior_flag 0); sic: no clamp of the generations here */
		for (i = 0; i < data.totalsize; i++) ptr->self_rates[i] = ran1();
		for (i = 0; i < data.totalsize; i++) ptr->generation[i] = rgeom(1 - ptr->self_rates[i]);
		fprintf(G, "chain %d geninit hgen=%016llx hS=%016llx", chn, (unsigned long long)hash_i32v(ptr->generation, D.N),
			(unsigned long long)hash_f64v(ptr->self_rates, D.N));
		seeds_line();
	}
	if (data.mode == 5) { /* mcmc_INDV_inbreedcoff, mcmc.c:412-415 (prior_flag 0) */
		for (i = 0; i < data.totalsize; i++) ptr->inbreed[i] = ran1();
		fprintf(G, "chain %d geninit hS=%016llx", chn, (unsigned long long)hash_f64v(ptr->inbreed, D.N));
		seeds_line();
	}
	if (data.mode == 4) { /* mcmc_POP_inbreedcoff, mcmc.c:255-259 */
		for (i = 0; i < data.popnum; i++) {
			ptr->inbreed[i] = initial.initd[chn][i];
			if (data.back_refl == 0) ptr->state[i] = dt_stat(ptr->inbreed[i]);
		}
	}
	update_ZQ(&ptr, data, 1, &qqnum);
	flatten(ptr, data, qqnum);
	fprintf(G, "chain %d zqinit hz=%016llx hqq=%016llx", chn, (unsigned long long)hash_z(&D, zflat),
		(unsigned long long)hash_f64v(qflat, (long)D.N * D.K));
	seeds_line();

	for (step = 0; step < initial.update; step++) {
		update_P(&ptr, data);
		flatten(ptr, data, qqnum);
		count_alleles_plain(&D, gflat, zflat, cflat);
		fprintf(G, "it %ld P hcnt=%016llx hfreq=%016llx", step, (unsigned long long)hash_counts(&D, cflat),
			(unsigned long long)hash_freq(&D, fflat));
		seeds_line();
		if (data.mode == 2) {
			update_S_POP(data, &ptr);
			fprintf(G, "it %ld S", step);
			for (i = 0; i < D.K; i++) fprintf(G, " %a", ptr->self_rates[i]);
			if (data.back_refl == 0)
				for (i = 0; i < D.K; i++) fprintf(G, " st%d", ptr->state[i]);
			seeds_line();
			update_G(data, &ptr);
			fprintf(G, "it %ld G hgen=%016llx", step, (unsigned long long)hash_i32v(ptr->generation, D.N));
			seeds_line();
		}
		if (data.mode == 3) { /* the loop body of mcmc.c:344-348 */
			update_S_IND(data.totalsize, &ptr);
			fprintf(G, "it %ld SI", step);
			for (i = 0; i < 4 && i < D.N; i++) fprintf(G, " %a", ptr->self_rates[i]);
			fprintf(G, " hS=%016llx", (unsigned long long)hash_f64v(ptr->self_rates, D.N));
			seeds_line();
			update_G(data, &ptr);
			fprintf(G, "it %ld G hgen=%016llx", step, (unsigned long long)hash_i32v(ptr->generation, D.N));
			seeds_line();
		}
		if (data.mode == 5) { /* the loop body of mcmc.c:420-433 */
			update_F_IND(data.totalsize, &ptr, data);
			fprintf(G, "it %ld SI", step);
			for (i = 0; i < 4 && i < D.N; i++) fprintf(G, " %a", ptr->inbreed[i]);
			fprintf(G, " hS=%016llx", (unsigned long long)hash_f64v(ptr->inbreed, D.N));
			seeds_line();
		}
		if (data.mode == 4) { /* the loop body of mcmc.c:262-268; the inbreeding coefficients go on the S line */
			update_inbreedcoff_POP(data, &ptr);
			fprintf(G, "it %ld S", step);
			for (i = 0; i < D.K; i++) fprintf(G, " %a", ptr->inbreed[i]);
			if (data.back_refl == 0)
				for (i = 0; i < D.K; i++) fprintf(G, " st%d", ptr->state[i]);
			seeds_line();
		}
		update_ZQ(&ptr, data, 0, &qqnum);
		flatten(ptr, data, qqnum);
		fprintf(G, "it %ld ZQ hz=%016llx hqq=%016llx hqqnum=%016llx", step, (unsigned long long)hash_z(&D, zflat),
			(unsigned long long)hash_f64v(qflat, (long)D.N * D.K), (unsigned long long)hash_f64v(nflat, (long)D.N * D.K));
		seeds_line();
		update_alpha(&ptr, data, qqnum);
		fprintf(G, "it %ld A alpha=%a", step, ptr->alpha);
		seeds_line();
		cal_lkh(&ptr, data);
		fprintf(G, "it %ld L totallkh=%a hindv=%016llx\n", step, ptr->totallkh,
			(unsigned long long)hash_f64v(ptr->indvlkh, D.N));
		if (detail_every > 0 && step % detail_every == 0) detail(ptr, data, step);

		if (step == initial.burnin - 1) allocate_chn(&mchain, data);
		if (step >= initial.burnin && (step + 1 - initial.burnin) % initial.thinning == 0) {
			store_chn(&mchain, ptr, data);
			if (cnt_step < cvg->ckrep) cvg->convg_ld[chn * cvg->ckrep + cnt_step] = ptr->totallkh;
			cnt_step++;
		}
		if (cnt_step == data.nstep_check_empty_cluster) {
			if ((mchain.flag_empty_cluster = check_empty_cluster(ptr, data)) == 1) {
				fprintf(G, "chain %d empty_cluster at step %ld\n", chn, step);
				break;
			}
		}
	}
	free_space(cnt_step, &mchain, chn, qqnum, ptr, data);
	return mchain;
}

/* mode 0 (mcmc_POP_no_admixture, mcmc.c:90-132): whole individuals are assigned, no alpha, no Q */
static void flatten0(UPMCMC *p, SEQDATA d)
{
	long i, j, k;
	for (i = 0; i < D.N; i++)
		for (j = 0; j < D.L; j++)
			for (k = 0; k < D.P; k++) zflat[(i * D.L + j) * D.P + k] = p->zz[i];
	for (k = 0; k < D.K; k++)
		for (j = 0; j < D.L; j++)
			for (i = 0; i < D.Amax; i++) fflat[(k * D.L + j) * D.Amax + i] = (i < d.allelenum[j]) ? p->freq[k][j][i] : 0.0;
}
static CHAIN run_chain0(SEQDATA data, INIT initial, int chn, CONVG *cvg)
{
	int j;
	long cnt_step = 0, step;
	CHAIN mchain;
	UPMCMC *ptr;
	mchain.flag_empty_cluster = 0;
	mchain.name_len = initial.name_len[chn];
	mchain.chn_name = cvector(0, mchain.name_len - 1);
	for (j = 0; j < mchain.name_len; j++) mchain.chn_name[j] = initial.chn_name[chn][j];
	mchain.steps = (int)((initial.update - initial.burnin) / initial.thinning);
	allocate_node(&ptr, data);
	update_Z(&ptr, data, 1);
	fprintf(G, "chain %d zinit hzz=%016llx", chn, (unsigned long long)hash_i32v(ptr->zz, D.N));
	seeds_line();
	for (step = 0; step < initial.update; step++) {
		update_P(&ptr, data);
		flatten0(ptr, data);
		count_alleles_plain(&D, gflat, zflat, cflat);
		fprintf(G, "it %ld P hcnt=%016llx hfreq=%016llx", step, (unsigned long long)hash_counts(&D, cflat), (unsigned long long)hash_freq(&D, fflat));
		seeds_line();
		update_Z(&ptr, data, 0);
		fprintf(G, "it %ld Z hzz=%016llx", step, (unsigned long long)hash_i32v(ptr->zz, D.N));
		seeds_line();
		cal_lkh(&ptr, data);
		fprintf(G, "it %ld L totallkh=%a hindv=%016llx\n", step, ptr->totallkh, (unsigned long long)hash_f64v(ptr->indvlkh, D.N));
		if (step == initial.burnin - 1) allocate_chn(&mchain, data);
		if (step >= initial.burnin && (step + 1 - initial.burnin) % initial.thinning == 0) {
			store_chn(&mchain, ptr, data);
			if (cnt_step < cvg->ckrep) cvg->convg_ld[chn * cvg->ckrep + cnt_step] = ptr->totallkh;
			cnt_step++;
		}
	}
	free_node(ptr, data);
	return mchain;
}
static void dump_chain0(CHAIN *c)
{
	int i, k;
	fprintf(G, "chain steps=%ld step=%ld flag_empty=%d totallkh=%a totallkh2=%a\n", c->steps, c->step, c->flag_empty_cluster, c->totallkh, c->totallkh2);
	dump_vec(G, "chain indvlkh", c->indvlkh, D.N);
	for (i = 0; i < D.N; i++) {
		fprintf(G, "chain z %d", i);
		for (k = 0; k < D.K; k++) fprintf(G, " %ld", c->z[i][k]);
		fprintf(G, "\n");
	}
}

int main(int argc, char **argv)
{
	SEQDATA data;
	INIT initial;
	CONVG cvg;
	CHAIN chain;
	int K, N, L, c, e, y, r, j, s1, s2, s3, mode, pf, detail_every, chn, i, jj, k;
	long u, b;
	int t;
	if (argc != 20) {
		fprintf(stderr, "usage: ref_dump data out K N L u b t c e y r j s1 s2 s3 mode pf detail_every\n");
		return 2;
	}
	K = atoi(argv[3]); N = atoi(argv[4]); L = atoi(argv[5]); u = atol(argv[6]); b = atol(argv[7]);
	t = atoi(argv[8]); c = atoi(argv[9]); e = atoi(argv[10]); y = atoi(argv[11]); r = atoi(argv[12]);
	j = atoi(argv[13]); s1 = atoi(argv[14]); s2 = atoi(argv[15]); s3 = atoi(argv[16]);
	mode = atoi(argv[17]); pf = atoi(argv[18]); detail_every = atoi(argv[19]);
	if ((G = fopen(argv[2], "w")) == NULL) return 2;

	setseeds(s1, s2, s3); /* InStruct.c:424-431 */
	data = read_data(argv[1], 2, N, K, L, "-9", 0, 0, 0.9, e, y, j, 0, mode, 0, 0, 10.0, 0, pf, 0, 1, 1, 0, 1.0e9);
	initial = read_init(NULL, c, K, u, b, t);

	D.N = data.totalsize; D.L = data.locinum; D.P = data.ploid; D.K = data.popnum; D.Amax = data.allelenum_max;
	D.allelenum = data.allelenum;
	vflat = malloc(sizeof(int) * D.N * D.L);
	zflat = malloc(sizeof(int) * D.N * D.L * D.P);
	gflat = malloc(sizeof(int) * D.N * D.L * D.P);
	cflat = malloc(sizeof(int) * D.K * D.L * D.Amax);
	fflat = malloc(sizeof(double) * D.K * D.L * D.Amax);
	qflat = malloc(sizeof(double) * D.N * D.K);
	nflat = malloc(sizeof(double) * D.N * D.K);
	for (i = 0; i < D.N; i++)
		for (jj = 0; jj < D.L; jj++) {
			vflat[i * D.L + jj] = (data.missindx[i][jj] != 1 && data.allelenum[jj] > 1);
			for (k = 0; k < D.P; k++) gflat[(i * D.L + jj) * D.P + k] = data.seqdata[i][jj][k];
		}
	D.valid = vflat;

	fprintf(G, "# instruct golden v1 (generated by oracle/ref_dump.c from the reference sweeps)\n");
	fprintf(G, "cfg N=%d L=%d K=%d P=%d Amax=%d mode=%d e=%d y=%d u=%ld b=%ld t=%d c=%d r=%d j=%d pf=%d s=%d,%d,%d\n",
		D.N, D.L, D.K, D.P, D.Amax, mode, e, y, u, b, t, c, r, j, pf, s1, s2, s3);
	fprintf(G, "data hgeno=%016llx hvalid=%016llx hallelenum=%016llx\n",
		(unsigned long long)hash_i32v(gflat, (long)D.N * D.L * D.P), (unsigned long long)hash_i32v(vflat, (long)D.N * D.L),
		(unsigned long long)hash_i32v(data.allelenum, D.L));
	for (chn = 0; chn < c; chn++) {
		fprintf(G, "initd %d", chn);
		for (i = 0; i < K; i++) fprintf(G, " %a", (double)initial.initd[chn][i]);
		fprintf(G, "\n");
	}
	fprintf(G, "init"); seeds_line();

	allocate_convg(data, &cvg, c, r, NULL);
	for (chn = 0; chn < c; chn++) {
		chain = (mode == 0) ? run_chain0(data, initial, chn, &cvg) : run_chain(data, initial, chn, &cvg, detail_every);
		if (chain.flag_empty_cluster == 1) { /* InStruct.c:185-190 */
			free_chain(&chain, data);
			chn--;
			continue;
		}
		fprintf(G, "chain %d done", chn); seeds_line();
		if (mode == 0) dump_chain0(&chain);
		else dump_chain(&chain, data);
		free_chain(&chain, data);
	}
	dump_vec(G, "convg", cvg.convg_ld, c * r);
	if (c > 1) fprintf(G, "GR %a\n", ref_GelmanRubin(cvg.convg_ld, cvg.n_chain, cvg.ckrep));
	fclose(G);
	return 0;
}
