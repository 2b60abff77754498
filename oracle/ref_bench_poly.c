/*
 * ref_bench_poly.c -- TEST INFRASTRUCTURE (oracle/): times the REAL reference ploidy-4 sweeps -- the bare loop of
 * mcmc_POP_tetra_selfing (reference poly_geno.c:98-116: update_P_auto / update_P_allo, calc_exfreq_*, update_S_POP,
 * update_ZQ, update_geno, cal_lkd) -- on an in-memory SEQDATA, for bench.py's ploidy-4 cpu_baseline
 * ("kind": "reference").  Nothing but the sweeps sits between the clock reads (no dumps, no hashing, no file reading).
 * Built only in the development container into oracle/_ref/ref_bench_poly (the reference translation unit is included
 * by absolute path at compile time; nothing is copied); the binary travels to the GPU box.
 *
 * Input: N*L*4 bytes, the sorted distinct allele codes observed per (individual, locus), 0xFF padded
 * (SEQDATA.seqdata / alleleid as transform_data2 leaves them, data_interface.c:571-669).
 *
 * usage: ref_bench_poly obs.u8 N L K iters autopoly s1 s2 s3   ->  one JSON line on stderr
 */
#include <time.h>
#include "/root/reference/poly_geno.c"

static double now(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char **argv)
{
	SEQDATA d;
	INIT ini;
	CHAIN ch;
	UPMCMC *ptr = NULL;
	POLY *tetra;
	double **qqnum, t[6] = {0, 0, 0, 0, 0, 0}, t0, t1, tinit, tot = 0;
	int N, L, K, iters, i, j, k, it, autopoly;
	unsigned char *buf;
	FILE *f, *devnull;
	if (argc != 10) { fprintf(stderr, "usage: ref_bench_poly obs.u8 N L K iters autopoly s1 s2 s3\n"); return 2; }
	N = atoi(argv[2]); L = atoi(argv[3]); K = atoi(argv[4]); iters = atoi(argv[5]); autopoly = atoi(argv[6]);
	memset(&d, 0, sizeof(d));
	d.ploid = 4; d.popnum = K; d.locinum = L; d.totalsize = N; d.mode = 2; d.type_freq = 1; d.back_refl = 1;
	d.nstep_check_empty_cluster = 20; d.print_iter = 0; d.print_freq = 0; d.autopoly = autopoly; d.missingnum = -9;
	buf = malloc((size_t)N * L * 4);
	if ((f = fopen(argv[1], "rb")) == NULL || fread(buf, 1, (size_t)N * L * 4, f) != (size_t)N * L * 4) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
	fclose(f);
	d.seqdata = i3tensor(0, N - 1, 0, L - 1, 0, 3);
	d.alleleid = imatrix(0, N - 1, 0, L - 1);
	d.missindx = imatrix(0, N - 1, 0, L - 1);
	d.missvec = ivector(0, N - 1);
	d.allelenum = ivector(0, L - 1);
	for (j = 0; j < L; j++) d.allelenum[j] = 0;
	for (i = 0; i < N; i++) {
		d.missvec[i] = 0;
		for (j = 0; j < L; j++) {
			int n = 0;
			for (k = 0; k < 4; k++) {
				int a = buf[((size_t)i * L + j) * 4 + k];
				d.seqdata[i][j][k] = -1;
				if (a != 0xff) { d.seqdata[i][j][n++] = a; if (a + 1 > d.allelenum[j]) d.allelenum[j] = a + 1; }
			}
			d.alleleid[i][j] = n;
			if (n == 0) d.seqdata[i][j][0] = -9;
			d.missindx[i][j] = (n == 0);
			d.missvec[i] += d.missindx[i][j];
		}
	}
	free(buf);
	d.allelenum_max = 0;
	for (j = 0; j < L; j++) if (d.allelenum[j] > d.allelenum_max) d.allelenum_max = d.allelenum[j];
	setseeds(atoi(argv[7]), atoi(argv[8]), atoi(argv[9]));
	ini = read_init(NULL, 1, K, 1000, 500, 10);
	devnull = freopen("/dev/null", "w", stdout); /* initial_chn prints the chain banner */
	(void)devnull;
	t0 = now();
	/* mcmc_POP_tetra_selfing, poly_geno.c:75-96 */
	tetra = (POLY *)malloc(sizeof(POLY));
	gen_polyinfo(tetra, d);
	ch.flag_empty_cluster = 0;
	initial_chn(&qqnum, d, &ptr, &ch, ini, 0);
	initial_geno(ptr, d, &tetra);
	for (i = 0; i < K; i++) {
		ptr->self_rates[i] = ini.initd[0][i];
		if (d.back_refl == 0) ptr->state[i] = dt_stat(ptr->self_rates[i]);
	}
	update_ZQ(&ptr, d, 1, &qqnum, tetra);
	tinit = now() - t0;
	for (it = 0; it < iters; it++) { /* poly_geno.c:98-116 */
		t0 = now();
		if (d.autopoly == 1) { update_P_auto(&ptr, d); calc_exfreq_auto(ptr, d, tetra); }
		else { update_P_allo(&ptr, d); calc_exfreq_allo(ptr, d, tetra); }
		t1 = now(); t[0] += t1 - t0;
		t0 = t1; update_S_POP(d, &ptr, tetra); t1 = now(); t[1] += t1 - t0;
		t0 = t1; update_ZQ(&ptr, d, 0, &qqnum, tetra); t1 = now(); t[2] += t1 - t0;
		t0 = t1; update_geno(ptr, d, tetra); t1 = now(); t[3] += t1 - t0;
		t0 = t1; ptr->totallkh = cal_lkd(ptr, d, tetra); t1 = now(); t[4] += t1 - t0;
	}
	for (i = 0; i < 5; i++) tot += t[i];
	fprintf(stderr, "{\"N\": %d, \"L\": %d, \"K\": %d, \"iters\": %d, \"autopoly\": %d, \"init_s\": %.4f, \"s_per_iter\": %.6f, "
		"\"update_P\": %.6f, \"update_S_POP\": %.6f, \"update_ZQ\": %.6f, \"update_geno\": %.6f, \"cal_lkd\": %.6f, \"totallkh\": %.6f}\n",
		N, L, K, iters, autopoly, tinit, tot / iters, t[0] / iters, t[1] / iters, t[2] / iters, t[3] / iters, t[4] / iters, ptr->totallkh);
	return 0;
}
