/*
 * ref_bench.c -- TEST INFRASTRUCTURE (oracle/): times the REAL reference sweeps
 * (update_P, update_S_POP, update_G, update_ZQ, update_alpha, cal_lkh -- reference mcmc.c:210-215)
 * on an in-memory SEQDATA, for bench.py's cpu_baseline ("kind": "reference").  Built only in the
 * development container into oracle/_ref/ref_bench (the reference translation unit is included by
 * absolute path at compile time; nothing is copied); the binary travels to the GPU box.
 *
 * The reference text reader needs ~100 bytes per token (data_interface.c:18,109-115), so the
 * benchmark input is handed over as packed bytes instead: N*L*2 allele codes, 0xFF = missing.
 *
 * usage: ref_bench geno.u8 N L K iters s1 s2 s3   ->  one JSON line on stdout
 */
#include <time.h>
#include "/root/reference/mcmc.c"

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
	UPMCMC *ptr;
	double **qqnum, t[7] = {0, 0, 0, 0, 0, 0, 0}, t0, t1, tinit, tit, per[64];
	int N, L, K, iters, i, j, k, it;
	unsigned char *buf;
	FILE *f, *devnull;
	if (argc != 9) { fprintf(stderr, "usage: ref_bench geno.u8 N L K iters s1 s2 s3\n"); return 2; }
	N = atoi(argv[2]); L = atoi(argv[3]); K = atoi(argv[4]); iters = atoi(argv[5]);
	memset(&d, 0, sizeof(d));
	d.ploid = 2; d.popnum = K; d.locinum = L; d.totalsize = N; d.mode = 2; d.type_freq = 1; d.back_refl = 1;
	d.nstep_check_empty_cluster = 20; d.print_iter = 0; d.print_freq = 0; d.autopoly = 1;
	buf = malloc((size_t)N * L * 2);
	if ((f = fopen(argv[1], "rb")) == NULL || fread(buf, 1, (size_t)N * L * 2, f) != (size_t)N * L * 2) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
	fclose(f);
	d.seqdata = i3tensor(0, N - 1, 0, L - 1, 0, 1);
	d.missindx = imatrix(0, N - 1, 0, L - 1);
	d.allelenum = ivector(0, L - 1);
	for (j = 0; j < L; j++) d.allelenum[j] = 0;
	for (i = 0; i < N; i++)
		for (j = 0; j < L; j++) {
			d.missindx[i][j] = 0;
			for (k = 0; k < 2; k++) {
				int a = buf[((size_t)i * L + j) * 2 + k];
				if (a == 0xff) { d.seqdata[i][j][k] = -9; d.missindx[i][j] = 1; }
				else { d.seqdata[i][j][k] = a; if (a + 1 > d.allelenum[j]) d.allelenum[j] = a + 1; }
			}
		}
	free(buf);
	d.allelenum_max = 0;
	for (j = 0; j < L; j++) if (d.allelenum[j] > d.allelenum_max) d.allelenum_max = d.allelenum[j];
	setseeds(atoi(argv[6]), atoi(argv[7]), atoi(argv[8]));
	ini = read_init(NULL, 1, K, 1000, 500, 10);
	devnull = freopen("/dev/null", "w", stdout); /* initial_chn prints the chain banner */
	(void)devnull;
	t0 = now();
	initial_chn(&qqnum, d, &ptr, &ch, ini, 0);
	for (i = 0; i < N; i++) {
		ptr->generation[i] = rgeom(ran1());
		if (ptr->generation[i] > 50) ptr->generation[i] = 50;
	}
	for (i = 0; i < K; i++) ptr->self_rates[i] = ini.initd[0][i];
	update_ZQ(&ptr, d, 1, &qqnum);
	tinit = now() - t0;
	for (it = 0; it < iters; it++) {
		tit = now();
		t0 = now(); update_P(&ptr, d); t1 = now(); t[0] += t1 - t0;
		t0 = t1; update_S_POP(d, &ptr); t1 = now(); t[1] += t1 - t0;
		t0 = t1; update_G(d, &ptr); t1 = now(); t[2] += t1 - t0;
		t0 = t1; update_ZQ(&ptr, d, 0, &qqnum); t1 = now(); t[3] += t1 - t0;
		t0 = t1; update_alpha(&ptr, d, qqnum); t1 = now(); t[4] += t1 - t0;
		t0 = t1; cal_lkh(&ptr, d); t1 = now(); t[5] += t1 - t0;
		if (it < 64) per[it] = t1 - tit;
	}
	for (i = 0; i < 6; i++) t[6] += t[i];
	fprintf(stderr, "{\"N\": %d, \"L\": %d, \"K\": %d, \"iters\": %d, \"init_s\": %.4f, \"s_per_iter\": %.6f, "
		"\"update_P\": %.6f, \"update_S_POP\": %.6f, \"update_G\": %.6f, \"update_ZQ\": %.6f, \"update_alpha\": %.6f, \"cal_lkh\": %.6f, "
		"\"totallkh\": %.6f, \"per_iter_s\": [", N, L, K, iters, tinit, t[6] / iters, t[0] / iters, t[1] / iters, t[2] / iters, t[3] / iters,
		t[4] / iters, t[5] / iters, ptr->totallkh);
	for (it = 0; it < iters && it < 64; it++) fprintf(stderr, "%s%.6f", it ? ", " : "", per[it]);
	fprintf(stderr, "]}\n");
	return 0;
}
