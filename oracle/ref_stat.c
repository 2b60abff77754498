/*
 * ref_stat.c -- TEST INFRASTRUCTURE (oracle/): posterior summaries of the REAL reference chain
 * (update_P, update_S_POP, update_G, update_ZQ, update_alpha, cal_lkh -- reference mcmc.c:210-215), for the
 * statistical-parity fixture of the keyed RNG schedule (tests/golden/make_keyed_stat.py).  The keyed schedule draws
 * the same distributions from other stream positions, so it cannot be compared sweep by sweep: its posterior means
 * must agree with the reference's within Monte-Carlo error instead.  Built only in the development container into
 * oracle/_ref/ref_stat (the reference translation unit is included by absolute path; nothing is copied).
 *
 * usage: ref_stat geno.u8 N L K iters burn s1 s2 s3 out.txt
 *   out.txt: "alpha <mean>", "totallkh <mean>", "self <K means>", then N lines of K posterior-mean qq values
 */
#include "/root/reference/mcmc.c"

int main(int argc, char **argv)
{
	SEQDATA d;
	INIT ini;
	CHAIN ch;
	UPMCMC *ptr;
	double **qqnum, *sq, *ss, sa = 0, sl = 0;
	int N, L, K, iters, burn, i, j, k, it, ns = 0;
	unsigned char *buf;
	FILE *f, *out;
	if (argc != 11) { fprintf(stderr, "usage: ref_stat geno.u8 N L K iters burn s1 s2 s3 out.txt\n"); return 2; }
	N = atoi(argv[2]); L = atoi(argv[3]); K = atoi(argv[4]); iters = atoi(argv[5]); burn = atoi(argv[6]);
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
	setseeds(atoi(argv[7]), atoi(argv[8]), atoi(argv[9]));
	ini = read_init(NULL, 1, K, 1000, 500, 10);
	out = fopen(argv[10], "w");
	if (!out) { fprintf(stderr, "cannot write %s\n", argv[10]); return 2; }
	if (!freopen("/dev/null", "w", stdout)) return 2; /* initial_chn prints the chain banner */
	initial_chn(&qqnum, d, &ptr, &ch, ini, 0);
	for (i = 0; i < N; i++) {
		ptr->generation[i] = rgeom(ran1());
		if (ptr->generation[i] > 50) ptr->generation[i] = 50;
	}
	for (i = 0; i < K; i++) ptr->self_rates[i] = ini.initd[0][i];
	update_ZQ(&ptr, d, 1, &qqnum);
	sq = calloc((size_t)N * K, sizeof(double));
	ss = calloc((size_t)K, sizeof(double));
	for (it = 0; it < iters; it++) {
		update_P(&ptr, d);
		update_S_POP(d, &ptr);
		update_G(d, &ptr);
		update_ZQ(&ptr, d, 0, &qqnum);
		update_alpha(&ptr, d, qqnum);
		cal_lkh(&ptr, d);
		if (it < burn) continue;
		ns++;
		for (i = 0; i < N; i++)
			for (k = 0; k < K; k++) sq[(size_t)i * K + k] += ptr->qq[i][k];
		for (k = 0; k < K; k++) ss[k] += ptr->self_rates[k];
		sa += ptr->alpha;
		sl += ptr->totallkh;
	}
	fprintf(out, "alpha %.17g\ntotallkh %.17g\nself", sa / ns, sl / ns);
	for (k = 0; k < K; k++) fprintf(out, " %.17g", ss[k] / ns);
	fprintf(out, "\n");
	for (i = 0; i < N; i++) {
		for (k = 0; k < K; k++) fprintf(out, "%s%.9g", k ? " " : "", sq[(size_t)i * K + k] / ns);
		fprintf(out, "\n");
	}
	fclose(out);
	return 0;
}
