/*
 * isg_oracle_poly.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference's autotetraploid sampler
 * (poly_geno.c:75-140 with -ap 1) in its reference configuration: glibc math, sequential sums, one
 * sequential Wichmann-Hill stream, and the reference's float/double promotion pattern for the genotype
 * frequency tables (poly_geno.h:19-20 declares them float).  Pinned byte-for-byte to
 * tests/golden/t*.golden, which oracle/ref_dump_poly.c generated from the real reference sweeps.
 *
 * NOT part of the product.  The MI355X kernels for this path are not written yet;
 * this file and its fixtures are the parity anchor they will be built against.
 *
 * Allotetraploid (-ap 0; fourth switch): update_P_allo (poly_geno.c:441-518), calc_exfreq_allo (:1592-1670), allo_genfreq
 * (:2122-2304), choose_two/tri/tetra_allo (:962-1215) and the two-subgenome terms of calc_genofq (:1262-1282); pinned to
 * tests/golden/ta*.golden the same way.
 *
 * usage (dump tool, main() below): orc_dump_poly data.txt out K N L u b t e r j s1 s2 s3 [math accum [keyed [allo]]]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "dump_fmt.h"
#include "isg_math.h"
#include "isg_wh.h"
/* genotype-class tables: the shared template (instruct_amd/csrc/isg_poly_tables.h) instantiated twice --
 * with glibc (the reference configuration pinned to the golden files) and with the canonical math */
#define PT_NAME(x) ptl_##x
#define PT_LOG(x) log(x)
#define PT_EXP(x) exp(x)
#include "isg_poly_tables.h"
#undef PT_NAME
#undef PT_LOG
#undef PT_EXP
#define PT_NAME(x) pti_##x
#define PT_LOG(x) isg_log(x)
#define PT_EXP(x) isg_exp(x)
#include "isg_poly_tables.h"
#undef PT_NAME
#undef PT_LOG
#undef PT_EXP

#define MIN2(X, Y) (((X) > (Y)) ? (Y) : (X))
#define P4 4

/* configuration switches (see isg_oracle.h): 0/0 = reference configuration */
static int g_math, g_accum;
static double m_log(double x) { return g_math ? isg_log(x) : log(x); }
static double m_exp(double x) { return g_math ? isg_exp(x) : exp(x); }
static double m_pow(double x, double y) { return g_math ? isg_pow(x, y) : pow(x, y); }
static double m_sqrt(double x) { return g_math ? isg_sqrt(x) : sqrt(x); }
typedef struct { double s; isg_acc a; } summer;
static void sum_init(summer *s) { s->s = 0; isg_acc_zero(&s->a); }
static void sum_add(summer *s, double v) { if (g_accum) isg_acc_add(&s->a, v); else s->s += v; }
static double sum_val(const summer *s) { return g_accum ? isg_acc_value(&s->a) : s->s; }

/* ------------------------------------------------------------------ RNG + samplers (random.c) */
static long sd1 = 13, sd2 = 4, sd3 = 1972;
/* keyed schedule (third switch; same samplers, same generator): every consumer first seeks to a stream position
 * that depends only on (iteration, phase, index), counted from the chain origin -- layout in
 * include/instruct_hip.h.  Off = the reference's sequential consumption. */
static int g_keyed;
static isg_wh g_origin;
static isg_wh_tables g_tab;
static uint64_t ky_SP, ky_SZ, ky_G0, ky_ZI0, ky_B0, ky_offS, ky_offZ, ky_offGE, ky_BLK, ky_iter;
static void rng_seek(uint64_t pos)
{
	isg_wh s;
	if (!g_keyed) return;
	s = isg_wh_jump(&g_tab, g_origin, pos);
	sd1 = s.s1; sd2 = s.s2; sd3 = s.s3;
}
static double ran1(void)
{
	sd1 = (171 * sd1) % 30269;
	sd2 = (172 * sd2) % 30307;
	sd3 = (170 * sd3) % 30323;
	{
		double x = sd1 / 30269.0 + sd2 / 30307.0 + sd3 / 30323.0;
		if (!g_math) return fmod(x, 1.0);
		if (x >= 2.0) x -= 2.0; else if (x >= 1.0) x -= 1.0;
		return x;
	}
}
#define E_CONST 2.71828182
static double rgamma1(double alpha)
{
	double u0 = ran1(), u1 = ran1(), r, x;
	if (u0 > E_CONST / (alpha + E_CONST)) {
		r = -m_log((alpha + E_CONST) * (1 - u0) / (alpha * E_CONST));
		if (u1 > m_pow(r, alpha - 1)) return -1;
		return r;
	}
	x = (alpha + E_CONST) * u0 / E_CONST;
	r = m_pow(x, 1 / alpha);
	if (u1 > m_exp(-r)) return -1;
	return r;
}
static double rgamma2(double alpha)
{
	double u1, u2, c1 = alpha - 1, c2 = (alpha - 1 / (6 * alpha)) / c1, c3 = 2 / c1, c4 = c3 + 2, c5 = 1 / m_sqrt(alpha), w;
	do {
		u1 = ran1();
		u2 = ran1();
		if (alpha > 2.5) u1 = u2 + c5 * (1 - 1.86 * u1);
	} while ((u1 >= 1) || (u1 <= 0));
	w = c2 * u2 / u1;
	if ((c3 * u1 + w + 1 / w) > c4)
		if ((c3 * m_log(u1) - m_log(w) + w) >= 1) return -1;
	return c1 * w;
}
static double rgamma(double alpha)
{
	double r = 0;
	if (alpha < 1) do { r = rgamma1(alpha); } while (r < 0);
	if (alpha == 1) r = -(1 / 1.0) * m_log(ran1());
	if (alpha > 1) do { r = rgamma2(alpha); } while (r < 0);
	return r;
}
static void rdirich(const double *alpha, int n, double *out, double add)
{
	double sum = 0;
	int k;
	for (k = 0; k < n; k++) { out[k] = rgamma(alpha[k] + add); sum += out[k]; }
	for (k = 0; k < n; k++) out[k] /= sum;
}
static int disc_unif(double *vec, int length) /* random.c:403-430 */
{
	int i, flag = 0;
	double x = ran1();
	for (i = 0; i < length; i++) vec[i] /= vec[length - 1];
	if (x <= vec[0] && x >= 0.00) flag = 0;
	else
		for (i = 1; i < length; i++)
			if (x > vec[i - 1] && x <= vec[i]) flag = i;
	return flag;
}

/* ------------------------------------------------------------------ problem + state */
static int N, L, K, Amax, back_refl;
static int *allelenum, *obs, *alleleid;    /* obs [N][L][4] sorted distinct codes, alleleid [N][L] */
static int *z, *geno, *state;              /* [N][L][4] */
static double *freq, *freq2, *qq, *qqnum, *S, *indvlkh, alpha, totallkh;
static int g_allo;                         /* 1: allotetraploid (-ap 0): copies 0, 1 / 2, 3 belong to two subgenomes (freq / freq2) */
/* POLY (poly_geno.h:10-21) */
static int num_allele, *allele_poly, (*genonum)[6], **genolist;
static float **exfreq, **genofreq;         /* [K*L] -> float[G] */
static int err_flag;

#define OBS(i, j, k) obs[((long)(i) * L + (j)) * P4 + (k)]
#define Z(i, j, k) z[((long)(i) * L + (j)) * P4 + (k)]
#define GENO(i, j, k) geno[((long)(i) * L + (j)) * P4 + (k)]
#define FREQ(k, j, a) freq[((long)(k) * L + (j)) * Amax + (a)]
#define FREQ2(k, j, a) freq2[((long)(k) * L + (j)) * Amax + (a)]
#define VALID(i, j) (alleleid[(long)(i) * L + (j)] != 0)

static int exists(int value, const int *vec, int leng) /* data_interface.c:865-877 */
{
	int i, flag = 0;
	for (i = 0; i < leng; i++) if (value == vec[i]) flag = 1;
	return flag;
}
static int find_id(int num, const int *array, int len) /* poly_geno.c:2367-2381 */
{
	int i;
	for (i = 0; i < len; i++) if (array[i] == num) return i;
	err_flag = 1;
	return 0;
}
static int chcksame(const int *p, int n) { int i, f = 0; for (i = 1; i < n; i++) if (p[i] != p[0]) f = 1; return f; }
static int nid_of(int j) { return find_id(allelenum[j], allele_poly, num_allele); }
static int gtot(int j) { return genonum[nid_of(j)][0]; }

static isg_polyclass *pclass; /* per distinct allele count */
static void gen_polyinfo(void) /* poly_geno.c:143-184, 1673-1800 */
{
	int i, j, k, cnt, tmp, *t = malloc(sizeof(int) * (L + 1));
	cnt = 0;
	for (i = 0; i < L; i++) if (!exists(allelenum[i], t, cnt)) t[cnt++] = allelenum[i];
	for (i = 0; i < cnt - 1; i++) for (j = i + 1; j < cnt; j++) if (t[i] > t[j]) { tmp = t[i]; t[i] = t[j]; t[j] = tmp; }
	num_allele = cnt;
	allele_poly = t;
	genonum = malloc(sizeof(*genonum) * cnt);
	genolist = malloc(sizeof(int *) * cnt);
	pclass = malloc(sizeof(isg_polyclass) * cnt);
	for (i = 0; i < cnt; i++) {
		genolist[i] = malloc(sizeof(int) * ((g_allo ? isg_allo_G(allele_poly[i]) : isg_poly_G(allele_poly[i])) + 1));
		if (g_allo) isg_allo_build(allele_poly[i], genonum[i], genolist[i]);
		else isg_poly_build(allele_poly[i], genonum[i], genolist[i]);
		pclass[i].n = allele_poly[i];
		pclass[i].G = genonum[i][0];
		memcpy(pclass[i].g, genonum[i], sizeof(int) * 6);
		pclass[i].list = genolist[i];
	}
	exfreq = malloc(sizeof(float *) * K * L);
	genofreq = malloc(sizeof(float *) * K * L);
	for (k = 0; k < K; k++)
		for (j = 0; j < L; j++) {
			exfreq[k * L + j] = calloc(gtot(j), sizeof(float));
			genofreq[k * L + j] = calloc(gtot(j), sizeof(float));
		}
}

static void set_geno(int i, int j, int a, int b, int c, int d) { GENO(i, j, 0) = a; GENO(i, j, 1) = b; GENO(i, j, 2) = c; GENO(i, j, 3) = d; }
static void two_allele_auto(int num, int i, int j) /* poly_geno.c:2440-2464 */
{
	int a = OBS(i, j, 0), b = OBS(i, j, 1);
	if (num == 1) set_geno(i, j, a, a, a, b);
	else if (num == 2) set_geno(i, j, b, b, b, a);
	else if (num == 3) set_geno(i, j, a, a, b, b);
}
static void tri_allele_auto(int num, int i, int j) /* poly_geno.c:2509-2530 */
{
	int a = OBS(i, j, 0), b = OBS(i, j, 1), c = OBS(i, j, 2);
	if (num == 1) set_geno(i, j, a, a, b, c);
	else if (num == 2) set_geno(i, j, b, b, a, c);
	else if (num == 3) set_geno(i, j, c, c, a, b);
}
/* two / tri / tetra_allele_allo (poly_geno.c:2466-2507, 2532-2612, 2614-2656): candidate `num` (1-based) as positions into
 * the observed allele list, first subgenome's pair then the second's */
static const signed char ALLO2[7][4] = {{0, 0, 0, 1}, {0, 1, 0, 0}, {0, 0, 1, 1}, {1, 1, 0, 0}, {0, 1, 1, 1}, {1, 1, 0, 1}, {0, 1, 0, 1}};
static const signed char ALLO3[12][4] = {{0, 0, 1, 2}, {1, 2, 0, 0}, {1, 1, 0, 2}, {0, 2, 1, 1}, {2, 2, 0, 1}, {0, 1, 2, 2},
					 {0, 1, 1, 2}, {1, 2, 0, 1}, {1, 2, 0, 2}, {0, 2, 1, 2}, {0, 2, 0, 1}, {0, 1, 0, 2}};
static const signed char ALLO4[6][4] = {{0, 1, 2, 3}, {2, 3, 0, 1}, {0, 2, 1, 3}, {1, 3, 0, 2}, {0, 3, 1, 2}, {1, 2, 0, 3}};
static const signed char *allo_pattern(int naid, int num) { return naid == 2 ? ALLO2[num - 1] : naid == 3 ? ALLO3[num - 1] : ALLO4[num - 1]; }
static int allo_ncand(int naid) { return naid == 2 ? 7 : naid == 3 ? 12 : 6; } /* POLY.num_allogeno, poly_geno.c:146-149 */
static void set_allo(int num, int naid, int i, int j)
{
	const signed char *p = allo_pattern(naid, num);
	set_geno(i, j, OBS(i, j, p[0]), OBS(i, j, p[1]), OBS(i, j, p[2]), OBS(i, j, p[3]));
}
static int choose_unif(int temp) /* poly_geno.c:840-852 */
{
	double tmp[16];
	int j;
	for (j = 0; j < temp; j++) tmp[j] = (double)(j + 1) / (double)temp;
	return disc_unif(tmp, temp) + 1;
}
static void initial_geno(void) /* poly_geno.c:316-369 (autopoly) */
{
	int i, j, k;
	rng_seek(ky_G0);
	for (i = 0; i < N; i++)
		for (j = 0; j < L; j++) {
			if (!VALID(i, j)) continue;
			if (g_allo && alleleid[(long)i * L + j] > 1) { set_allo(choose_unif(allo_ncand(alleleid[(long)i * L + j])), alleleid[(long)i * L + j], i, j); continue; }
			switch (alleleid[(long)i * L + j]) {
			case 1: for (k = 0; k < P4; k++) GENO(i, j, k) = OBS(i, j, 0); break;
			case 2: two_allele_auto(choose_unif(3), i, j); break;
			case 3: tri_allele_auto(choose_unif(3), i, j); break;
			case 4: for (k = 0; k < P4; k++) GENO(i, j, k) = OBS(i, j, k); break;
			}
		}
}

static void update_P_auto(void) /* poly_geno.c:390-438 */
{
	int *cnt = calloc((size_t)K * L * Amax, sizeof(int));
	double *tmp = malloc(sizeof(double) * Amax);
	int i, j, k;
	for (i = 0; i < N; i++)
		for (j = 0; j < L; j++)
			if (VALID(i, j))
				for (k = 0; k < P4; k++) cnt[((long)Z(i, j, k) * L + j) * Amax + GENO(i, j, k)]++;
	for (i = 0; i < K; i++)
		for (j = 0; j < L; j++) {
			for (k = 0; k < allelenum[j]; k++) tmp[k] = (double)cnt[((long)i * L + j) * Amax + k];
			rng_seek(ky_B0 + ky_iter * ky_BLK + ((uint64_t)i * L + j) * ky_SP);
			rdirich(tmp, allelenum[j], &FREQ(i, j, 0), 1.0);
		}
	free(cnt);
	free(tmp);
}

static void update_P_allo(void) /* poly_geno.c:441-518: per (cluster, locus) the first subgenome's Dirichlet, then the second's */
{
	int *cnt = calloc((size_t)2 * K * L * Amax, sizeof(int)), *cnt2 = cnt + (size_t)K * L * Amax;
	double *tmp = malloc(sizeof(double) * Amax);
	int i, j, k;
	for (i = 0; i < N; i++)
		for (j = 0; j < L; j++)
			if (VALID(i, j))
				for (k = 0; k < P4; k++) (k < P4 / 2 ? cnt : cnt2)[((long)Z(i, j, k) * L + j) * Amax + GENO(i, j, k)]++;
	for (i = 0; i < K; i++)
		for (j = 0; j < L; j++) {
			for (k = 0; k < allelenum[j]; k++) tmp[k] = (double)cnt[((long)i * L + j) * Amax + k];
			rng_seek(ky_B0 + ky_iter * ky_BLK + ((uint64_t)i * L + j) * ky_SP); /* (keyed: both Dirichlets from this position on) */
			rdirich(tmp, allelenum[j], &FREQ(i, j, 0), 1.0);
			for (k = 0; k < allelenum[j]; k++) tmp[k] = (double)cnt2[((long)i * L + j) * Amax + k];
			rdirich(tmp, allelenum[j], &FREQ2(i, j, 0), 1.0);
		}
	free(cnt);
	free(tmp);
}

static void calc_exfreq_auto(void) /* poly_geno.c:1515-1590; -ap 0: calc_exfreq_allo :1592-1670 */
{
	int i, k;
	for (k = 0; k < K; k++)
		for (i = 0; i < L; i++) {
			if (g_allo) {
				if (g_math) pti_exfreq_row_allo(&pclass[nid_of(i)], &FREQ(k, i, 0), &FREQ2(k, i, 0), exfreq[k * L + i]);
				else ptl_exfreq_row_allo(&pclass[nid_of(i)], &FREQ(k, i, 0), &FREQ2(k, i, 0), exfreq[k * L + i]);
				continue;
			}
			if (g_math) pti_exfreq_row(&pclass[nid_of(i)], &FREQ(k, i, 0), exfreq[k * L + i]);
			else ptl_exfreq_row(&pclass[nid_of(i)], &FREQ(k, i, 0), exfreq[k * L + i]);
		}
}
static void calc_self_genofreq(double self_rate, float **tab, int k, int own) /* poly_geno.c:1219-1233, 1803-2028 */
{
	int i, e = 0;
	for (i = 0; i < L; i++) {
		float *out = own ? tab[k * L + i] : tab[i];
		if (g_allo) {
			if (g_math) pti_genfreq_row_allo((float)self_rate, &pclass[nid_of(i)], exfreq[k * L + i], out, &e);
			else ptl_genfreq_row_allo((float)self_rate, &pclass[nid_of(i)], exfreq[k * L + i], out, &e);
			continue;
		}
		if (g_math) pti_genfreq_row((float)self_rate, &pclass[nid_of(i)], exfreq[k * L + i], out, &e);
		else ptl_genfreq_row((float)self_rate, &pclass[nid_of(i)], exfreq[k * L + i], out, &e);
	}
	if (e) err_flag = 3;
}

static int copy_num(int val, const int *vec, int leng) { int i, n = 0; for (i = 0; i < leng; i++) if (val == vec[i]) n++; return n; }
static int get_cat_auto(const int *g) /* poly_geno.c:1313-1339 */
{
	int i, cnt = 0, tmp[4];
	tmp[cnt++] = g[0];
	for (i = 1; i < P4; i++) if (!exists(g[i], tmp, cnt)) tmp[cnt++] = g[i];
	switch (cnt) {
	case 1: return 0;
	case 2: return copy_num(tmp[0], g, P4) == 2 ? 2 : 1;
	case 3: return 3;
	default: return 4;
	}
}
static int get_index_auto(int j, const int *g, int *cat) /* poly_geno.c:1289-1311 */
{
	int i, temp = g[0], id = nid_of(j);
	*cat = g_allo ? isg_allo_cat(g) : get_cat_auto(g); /* get_cat_allo, poly_geno.c:1341-1372 */
	for (i = 1; i < P4; i++) temp = temp * allelenum[j] + g[i];
	return find_id(temp, genolist[id], genonum[id][0]);
}
static double calc_genofq(int j, int i, const int *zz) /* poly_geno.c:1235-1286 */
{
	int cat, m, gid;
	double ld = 0;
	if (!VALID(i, j)) return 0;
	gid = get_index_auto(j, &GENO(i, j, 0), &cat);
	if (chcksame(zz, P4) == 0) return (double)genofreq[zz[0] * L + j][gid];
	if (g_accum) { /* canonical: the terms are accumulated exactly by the caller */
		err_flag |= 8;
		return 0;
	}
	if (g_allo) { /* poly_geno.c:1262-1282 */
		for (m = 0; m < P4 / 2; m++) ld += m_log(FREQ(zz[m], j, GENO(i, j, m)));
		for (m = P4 / 2; m < P4; m++) ld += m_log(FREQ2(zz[m], j, GENO(i, j, m)));
		switch (cat) {
		case 1: ld += m_log(2); break;
		case 2: ld += m_log(2); break;
		case 3: ld += m_log(4); break;
		}
		return ld;
	}
	for (m = 0; m < P4; m++) ld += m_log(FREQ(zz[m], j, GENO(i, j, m)));
	switch (cat) {
	case 1: ld += m_log(4); break;
	case 2: ld += m_log(6); break;
	case 3: ld += m_log(12); break;
	case 4: ld += m_log(24); break;
	}
	return ld;
}
/* the terms of one (individual, locus) added one by one to a sum (same terms, same order as calc_genofq /
 * cal_lkd_props; with exact accumulation the order is irrelevant).  `id`/`tab`: cluster whose table is replaced */
static void add_terms(summer *t, int i, int j, int id, float **tab)
{
	int cat, m, gid = get_index_auto(j, &GENO(i, j, 0), &cat);
	const int *zz = &Z(i, j, 0);
	if (chcksame(zz, P4) == 0) {
		if (tab && id == zz[0]) sum_add(t, (double)tab[j][gid]);
		else sum_add(t, (double)genofreq[zz[0] * L + j][gid]);
		return;
	}
	if (g_allo) {
		for (m = 0; m < P4 / 2; m++) sum_add(t, m_log(FREQ(zz[m], j, GENO(i, j, m))));
		for (m = P4 / 2; m < P4; m++) sum_add(t, m_log(FREQ2(zz[m], j, GENO(i, j, m))));
		if (cat == 1 || cat == 2) sum_add(t, m_log(2));
		if (cat == 3) sum_add(t, m_log(4));
		return;
	}
	for (m = 0; m < P4; m++) sum_add(t, m_log(FREQ(zz[m], j, GENO(i, j, m))));
	switch (cat) {
	case 1: sum_add(t, m_log(4)); break;
	case 2: sum_add(t, m_log(6)); break;
	case 3: sum_add(t, m_log(12)); break;
	case 4: sum_add(t, m_log(24)); break;
	}
}
static double cal_lkd(void) /* poly_geno.c:715-735 */
{
	int i, j;
	double ld, sum = 0;
	if (g_accum) { /* canonical: per-individual and total sums order-independent */
		summer tot;
		sum_init(&tot);
		for (i = 0; i < N; i++) {
			summer t;
			sum_init(&t);
			for (j = 0; j < L; j++) if (VALID(i, j)) add_terms(&t, i, j, -1, NULL);
			indvlkh[i] = sum_val(&t);
			sum_add(&tot, indvlkh[i]);
		}
		return sum_val(&tot);
	}
	for (i = 0; i < N; i++) {
		ld = 0;
		for (j = 0; j < L; j++) if (VALID(i, j)) ld += calc_genofq(j, i, &Z(i, j, 0));
		indvlkh[i] = ld;
		sum += ld;
	}
	return sum;
}
/* canonical mode: the total log-likelihood as ONE exact sum over all terms (what the MH ratio of
 * update_S_POP compares; the reference's cal_lkd sums per-individual subtotals) */
static double cal_lkd_flat(int id, float **tab)
{
	int i, j;
	summer t;
	sum_init(&t);
	for (i = 0; i < N; i++)
		for (j = 0; j < L; j++) if (VALID(i, j)) add_terms(&t, i, j, id, tab);
	return sum_val(&t);
}
static double cal_lkd_props(int id, float **tab) /* poly_geno.c:645-711 */
{
	int i, j, m, cat, gid;
	double ld = 0;
	if (g_accum) return cal_lkd_flat(id, tab);
	for (i = 0; i < N; i++)
		for (j = 0; j < L; j++) {
			if (!VALID(i, j)) continue;
			gid = get_index_auto(j, &GENO(i, j, 0), &cat);
			if (chcksame(&Z(i, j, 0), P4) == 0) {
				if (id == Z(i, j, 0)) ld += (double)tab[j][gid];
				else ld += (double)genofreq[Z(i, j, 0) * L + j][gid];
			} else if (g_allo) { /* poly_geno.c:691-706 */
				for (m = 0; m < P4 / 2; m++) ld += log(FREQ(Z(i, j, m), j, GENO(i, j, m)));
				for (m = P4 / 2; m < P4; m++) ld += log(FREQ2(Z(i, j, m), j, GENO(i, j, m)));
				switch (cat) {
				case 1: ld += log(2); break;
				case 2: ld += log(2); break;
				case 3: ld += log(4); break;
				}
			} else {
				for (m = 0; m < P4; m++) ld += log(FREQ(Z(i, j, m), j, GENO(i, j, m)));
				switch (cat) {
				case 1: ld += log(4); break;
				case 2: ld += log(6); break;
				case 3: ld += log(12); break;
				case 4: ld += log(24); break;
				}
			}
		}
	return ld;
}

static int dt_stat(double num)
{
	double eps = 0.001;
	if (num <= 0.000 + eps && num >= 0.000 - eps) return 0;
	if (num >= 1.000 - eps && num <= 1.000 + eps) return 2;
	if (num >= 0.0 + eps && num < 1.000 - eps) return 1;
	err_flag = 4;
	return 1;
}
static double adpt_indp(int *stat_tmp, int stat)
{
	double tmp = 0, tt;
	if (stat == 0) { if (ran1() < 0.50) { tmp = 0; *stat_tmp = 0; } else { tmp = ran1(); *stat_tmp = 1; } }
	else if (stat == 2) { if (ran1() < 0.5) { tmp = 1; *stat_tmp = 2; } else { tmp = ran1(); *stat_tmp = 1; } }
	else { tt = ran1(); if (tt <= 0.05) { tmp = 0; *stat_tmp = 0; } else if (tt >= 0.95) { tmp = 1; *stat_tmp = 2; } else { tmp = ran1(); *stat_tmp = 1; } }
	return tmp;
}
static double q_trans(int a, int b)
{
	if (a == 0) return (b == 0 || b == 1) ? 0.5 : 0.0;
	if (a == 2) return (b == 2 || b == 1) ? 0.5 : 0.0;
	if (a == 1) return (b == 0 || b == 2) ? 0.05 : (b == 1 ? 0.90 : 0.0);
	return 0.0;
}

static void update_S_POP(void) /* poly_geno.c:584-643 */
{
	int i, j, *tem_stat = malloc(sizeof(int) * K);
	double delta0 = 0.05, mhratio, tmp = 0;
	float **tab = malloc(sizeof(float *) * L);
	for (i = 0; i < L; i++) tab[i] = malloc(sizeof(float) * gtot(i));
	for (j = 0; j < K; j++) calc_self_genofreq(S[j], genofreq, j, 1);
	rng_seek(ky_B0 + ky_iter * ky_BLK + ky_offS);
	for (j = 0; j < K; j++) {
		if (back_refl == 1) {
			tmp = ran1() * 2 * delta0 - delta0;
			tmp += S[j];
			if (tmp <= 0.000) tmp = 0.000 - tmp;
			else if (tmp >= 1.000) tmp = 1.000 - (tmp - 1.000);
		} else {
			for (i = 0; i < K; i++) tem_stat[i] = state[i];
			tmp = adpt_indp(tem_stat + j, state[j]);
		}
		calc_self_genofreq(tmp, tab, j, 0);
		if (g_accum) { cal_lkd(); mhratio = cal_lkd_flat(j, tab) - cal_lkd_flat(-1, NULL); } /* both totals as flat exact sums */
		else mhratio = cal_lkd_props(j, tab) - cal_lkd();
		if (back_refl == 0) {
			double h = 1.0;
			for (i = 0; i < K; i++) h *= q_trans(state[i], tem_stat[i]) / q_trans(tem_stat[i], state[i]);
			mhratio *= h; /* sic: the Hastings factor multiplies the LOG ratio (poly_geno.c:622-623) */
		}
		if (ran1() < m_exp(MIN2(0, mhratio))) {
			S[j] = tmp;
			if (back_refl == 0) state[j] = tem_stat[j];
			for (i = 0; i < L; i++) memcpy(genofreq[j * L + i], tab[i], sizeof(float) * gtot(i));
		}
	}
	for (i = 0; i < L; i++) free(tab[i]);
	free(tab);
	free(tem_stat);
}

static void update_ZQ(int init_flag) /* poly_geno.c:750-836 */
{
	int i, j, k, m;
	double *tmp = malloc(sizeof(double) * K);
	for (i = 0; i < N; i++) {
		rng_seek(init_flag == 1 ? ky_ZI0 + (uint64_t)i * ky_SZ : ky_B0 + ky_iter * ky_BLK + ky_offZ + (uint64_t)i * ky_SZ);
		for (j = 0; j < L; j++)
			if (VALID(i, j))
				for (k = 0; k < P4; k++) {
					for (m = 0; m < K; m++) {
						if (init_flag == 1) tmp[m] = (double)(m + 1) / K;
						else {
							tmp[m] = qq[(long)i * K + m] * FREQ(m, j, GENO(i, j, k));
							if (m >= 1) tmp[m] += tmp[m - 1];
						}
					}
					Z(i, j, k) = disc_unif(tmp, K);
				}
		for (m = 0; m < K; m++) qqnum[(long)i * K + m] = 0.0;
		for (j = 0; j < L; j++)
			if (VALID(i, j))
				for (k = 0; k < P4; k++) qqnum[(long)i * K + Z(i, j, k)] += 1.0;
		for (k = 0; k < K; k++) tmp[k] = qqnum[(long)i * K + k];
		rdirich(tmp, K, &qq[(long)i * K], alpha);
	}
	free(tmp);
}

static int choose_auto(int i, int j, int n_type) /* choose_two_auto / choose_tri_auto, poly_geno.c:854-960 */
{
	int a, b, num[3], id[3], n = allelenum[j], nid = nid_of(j);
	double tmp[3], fq[3], tm;
	for (a = 0; a < n_type; a++) id[a] = OBS(i, j, a);
	if (chcksame(&Z(i, j, 0), P4) == 0) {
		if (n_type == 2) {
			num[0] = id[0] * n * (n * n + n + 1) + id[1];
			num[1] = id[1] * n * (n * n + n + 1) + id[0];
			num[2] = (id[0] * n * n + id[1]) * (n + 1);
		} else {
			num[0] = id[0] * n * n * (n + 1) + id[1] * n + id[2];
			num[1] = id[1] * n * n * (n + 1) + id[0] * n + id[2];
			num[2] = id[2] * n * n * (n + 1) + id[0] * n + id[1];
		}
		for (a = 0; a < 3; a++) tmp[a] = (double)genofreq[Z(i, j, 0) * L + j][find_id(num[a], genolist[nid], genonum[nid][0])];
	} else {
		for (a = 0; a < n_type; a++) {
			fq[a] = 0;
			for (b = 0; b < K; b++) fq[a] += qq[(long)i * K + b] * FREQ(b, j, id[a]);
		}
		if (n_type == 2) {
			tmp[0] = m_log(4) + 3 * m_log(fq[0]) + m_log(fq[1]);
			tmp[1] = m_log(4) + 3 * m_log(fq[1]) + m_log(fq[0]);
			tmp[2] = m_log(6) + 2 * m_log(fq[0]) + 2 * m_log(fq[1]);
		} else {
			tmp[0] = 2 * m_log(fq[0]) + m_log(fq[1]) + m_log(fq[2]);
			tmp[1] = 2 * m_log(fq[1]) + m_log(fq[0]) + m_log(fq[2]);
			tmp[2] = 2 * m_log(fq[2]) + m_log(fq[1]) + m_log(fq[0]);
		}
	}
	tm = tmp[0];
	for (a = 0; a < 3; a++) tmp[a] = m_exp(tmp[a] - tm);
	for (a = 1; a < 3; a++) tmp[a] += tmp[a - 1];
	return disc_unif(tmp, 3) + 1;
}
/* choose_two / tri / tetra_allo (poly_geno.c:962-1215): the candidates' weights are the cluster's genotype frequencies when
 * all four copies sit in one cluster, else products of the individual's expected allele frequencies in the two subgenomes
 * (a factor 2 when both pairs are heterozygous: candidates 7 of two alleles and 7..12 of three) */
static int choose_allo(int i, int j, int naid)
{
	const int nc = allo_ncand(naid), n = allelenum[j];
	double tmp[12], fq[4], fq2[4], tm;
	int a, b;
	if (chcksame(&Z(i, j, 0), P4) == 0) {
		for (a = 0; a < nc; a++) {
			const signed char *p = allo_pattern(naid, a + 1);
			tmp[a] = (double)genofreq[Z(i, j, 0) * L + j][isg_allo_row(n, OBS(i, j, p[0]), OBS(i, j, p[1]), OBS(i, j, p[2]), OBS(i, j, p[3]))];
		}
	} else {
		for (a = 0; a < naid; a++) {
			fq[a] = 0;
			fq2[a] = 0;
			for (b = 0; b < K; b++) {
				fq[a] += qq[(long)i * K + b] * FREQ(b, j, OBS(i, j, a));
				fq2[a] += qq[(long)i * K + b] * FREQ2(b, j, OBS(i, j, a));
			}
		}
		for (a = 0; a < nc; a++) {
			const signed char *p = allo_pattern(naid, a + 1);
			const int hetA = p[0] != p[1], hetB = p[2] != p[3];
			/* the reference writes each sum out term by term; the order of its terms: [log 2 +] first pair, second pair, with a
			 * homozygous pair as 2 log f */
			double v = 0;
			int first = 1;
			if (hetA && hetB && naid < 4) { v = m_log(2); first = 0; }
			if (hetA) { v = first ? m_log(fq[p[0]]) : v + m_log(fq[p[0]]); first = 0; v += m_log(fq[p[1]]); }
			else { v = first ? 2 * m_log(fq[p[0]]) : v + 2 * m_log(fq[p[0]]); first = 0; }
			if (hetB) { v += m_log(fq2[p[2]]); v += m_log(fq2[p[3]]); }
			else v += 2 * m_log(fq2[p[2]]);
			tmp[a] = v;
		}
	}
	tm = tmp[0];
	for (a = 0; a < nc; a++) tmp[a] = m_exp(tmp[a] - tm);
	for (a = 1; a < nc; a++) tmp[a] += tmp[a - 1];
	return disc_unif(tmp, nc) + 1;
}
static void update_geno(void) /* poly_geno.c:520-580 (autopoly); the canonical-order fix-up never fires for -ap 1 */
{
	int i, j, k;
	rng_seek(ky_B0 + ky_iter * ky_BLK + ky_offGE);
	for (i = 0; i < N; i++)
		for (j = 0; j < L; j++) {
			if (!VALID(i, j)) continue;
			if (g_allo && alleleid[(long)i * L + j] > 1) { set_allo(choose_allo(i, j, alleleid[(long)i * L + j]), alleleid[(long)i * L + j], i, j); continue; }
			switch (alleleid[(long)i * L + j]) {
			case 1: for (k = 0; k < P4; k++) GENO(i, j, k) = OBS(i, j, 0); break;
			case 2: two_allele_auto(choose_auto(i, j, 2), i, j); break;
			case 3: tri_allele_auto(choose_auto(i, j, 3), i, j); break;
			case 4: for (k = 0; k < P4; k++) GENO(i, j, k) = OBS(i, j, k); break;
			}
		}
}

/* ------------------------------------------------------------------ reader: transform_data2 (data_interface.c:571-669) */
static int read_poly(const char *path)
{
	FILE *f = fopen(path, "r");
	char *line = NULL, **tok = NULL;
	size_t cap = 0;
	long nrows = 0, ntok = 0, T = -1, captok = 0, i, j, k;
	if (!f) return -1;
	while (getline(&line, &cap, f) > 0) {
		char *s = line;
		long cnt = 0;
		while (*s) {
			while (isspace((unsigned char)*s)) s++;
			if (!*s) break;
			{
				char *b = s;
				while (*s && !isspace((unsigned char)*s)) s++;
				if (ntok == captok) { captok = captok ? captok * 2 : 4096; tok = realloc(tok, sizeof(char *) * captok); }
				tok[ntok++] = strndup(b, (size_t)(s - b));
				cnt++;
			}
		}
		if (cnt == 0) continue;
		if (T < 0) T = cnt;
		nrows++;
	}
	fclose(f);
	N = (int)nrows;
	L = (int)(T / P4);
	allelenum = calloc(L, sizeof(int));
	obs = malloc(sizeof(int) * (size_t)N * L * P4);
	alleleid = calloc((size_t)N * L, sizeof(int));
	for (i = 0; i < (long)N * L * P4; i++) obs[i] = -1;
	for (j = 0; j < L; j++) {
		char **types = malloc(sizeof(char *) * (size_t)N * P4);
		int cnt = 0, m;
		for (i = 0; i < N; i++)
			for (k = 0; k < P4; k++) {
				const char *t = tok[i * T + j * P4 + k];
				if (strcmp(t, "-9") == 0) continue;
				for (m = 0; m < cnt; m++) if (strcmp(types[m], t) == 0) break;
				if (m == cnt) types[cnt++] = (char *)t;
			}
		allelenum[j] = cnt;
		for (i = 0; i < N; i++) {
			int flag = 0, a, b;
			for (k = 0; k < P4; k++) {
				const char *t = tok[i * T + j * P4 + k];
				if (strcmp(t, "-9") == 0) continue;
				for (m = 0; m < cnt; m++)
					if (strcmp(t, types[m]) == 0 && !exists(m, &OBS(i, j, 0), flag)) OBS(i, j, flag++) = m;
			}
			for (a = 0; a < flag - 1; a++)
				for (b = a + 1; b < flag; b++)
					if (OBS(i, j, a) > OBS(i, j, b)) { int t2 = OBS(i, j, a); OBS(i, j, a) = OBS(i, j, b); OBS(i, j, b) = t2; }
			alleleid[i * L + j] = flag;
		}
		free(types);
	}
	Amax = 0;
	for (j = 0; j < L; j++) if (allelenum[j] > Amax) Amax = allelenum[j];
	return 0;
}

/* ------------------------------------------------------------------ dump tool */
static uint64_t hash_tables(float **tab)
{
	uint64_t h = fnv_init();
	int k, j, g;
	for (k = 0; k < K; k++)
		for (j = 0; j < L; j++)
			for (g = 0; g < gtot(j); g++) h = fnv_bytes(h, &tab[k * L + j][g], 4);
	return h;
}
int main(int argc, char **argv)
{
	FILE *G;
	dump_dims D;
	int *vflat, *gflat, *cflat, e, r, jj, s1, s2, s3, t, i, j, k;
	long u, b, step, cnt_step = 0;
	float *initd;
	double *convg;
	/* CHAIN running means (store_chn, mcmc.c:1320-1456) */
	double c_tot = 1, c_tot2 = 1, *c_indv, *c_S, *c_qq;
	long c_step = 0, steps;
	int flag_empty = 0;
	if (argc != 15 && argc != 17 && argc != 18 && argc != 19) { fprintf(stderr, "usage: orc_dump_poly data out K N L u b t e r j s1 s2 s3 [math accum [keyed [allo]]]\n"); return 2; }
	if (argc >= 17) { g_math = atoi(argv[15]); g_accum = atoi(argv[16]); }
	if (argc >= 18) g_keyed = atoi(argv[17]);
	if (argc == 19) g_allo = atoi(argv[18]);
	K = atoi(argv[3]); u = atol(argv[6]); b = atol(argv[7]); t = atoi(argv[8]); e = atoi(argv[9]); r = atoi(argv[10]); jj = atoi(argv[11]);
	s1 = atoi(argv[12]); s2 = atoi(argv[13]); s3 = atoi(argv[14]);
	back_refl = e;
	if ((G = fopen(argv[2], "w")) == NULL || read_poly(argv[1])) return 2;
	sd1 = s1; sd2 = s2; sd3 = s3;
	initd = malloc(sizeof(float) * K);
	for (i = 0; i < K; i++) initd[i] = (float)ran1();
	z = calloc((size_t)N * L * P4, sizeof(int)); geno = calloc((size_t)N * L * P4, sizeof(int)); state = calloc(K, sizeof(int));
	freq = calloc((size_t)K * L * Amax, sizeof(double)); freq2 = calloc((size_t)K * L * Amax, sizeof(double)); qq = calloc((size_t)N * K, sizeof(double)); qqnum = calloc((size_t)N * K, sizeof(double));
	S = calloc(K, sizeof(double)); indvlkh = calloc(N, sizeof(double));
	vflat = malloc(sizeof(int) * N * L); gflat = malloc(sizeof(int) * N * L * P4); cflat = malloc(sizeof(int) * K * L * Amax);
	for (i = 0; i < N; i++) for (j = 0; j < L; j++) vflat[i * L + j] = VALID(i, j);
	D.N = N; D.L = L; D.P = P4; D.K = K; D.Amax = Amax; D.allelenum = allelenum; D.valid = vflat;
	fprintf(G, "# instruct golden v1 ploidy 4 (generated by oracle/ref_dump_poly.c from the reference sweeps)\n");
	fprintf(G, "cfg N=%d L=%d K=%d P=4 Amax=%d e=%d u=%ld b=%ld t=%d r=%d j=%d s=%d,%d,%d%s\n", N, L, K, Amax, e, u, b, t, r, jj, s1, s2, s3, g_allo ? " ap=0" : "");
	{
		uint64_t h = fnv_init();
		for (i = 0; i < N; i++) for (j = 0; j < L; j++) for (k = 0; k < P4; k++) h = fnv_i32(h, k < alleleid[i * L + j] ? OBS(i, j, k) : -1);
		fprintf(G, "data hobs=%016llx halleleid=%016llx hallelenum=%016llx\n", (unsigned long long)h,
			(unsigned long long)hash_i32v(alleleid, (long)N * L), (unsigned long long)hash_i32v(allelenum, L));
	}
	fprintf(G, "initd 0");
	for (i = 0; i < K; i++) fprintf(G, " %a", (double)initd[i]);
	fprintf(G, "\ninit seeds=%ld %ld %ld\n", sd1, sd2, sd3);
	convg = calloc(r, sizeof(double));
	c_indv = malloc(sizeof(double) * N); c_S = malloc(sizeof(double) * K); c_qq = malloc(sizeof(double) * N * K);
	steps = (long)((u - b) / t);
#define SEEDS() fprintf(G, " seeds=%ld %ld %ld\n", sd1, sd2, sd3)
#define GFLAT() do { for (i = 0; i < N; i++) for (j = 0; j < L; j++) for (k = 0; k < P4; k++) gflat[((long)i * L + j) * P4 + k] = VALID(i, j) ? GENO(i, j, k) : -1; } while (0)
	gen_polyinfo();
	{ /* keyed layout; the chain origin is the stream state when the chain starts (after read_init's draws) */
		uint64_t amb = 0;
		for (i = 0; i < N * L; i++) amb += (alleleid[i] == 2 || alleleid[i] == 3 || (g_allo && alleleid[i] == 4)); /* loci whose genotype is drawn */
		isg_wh_tables_init(&g_tab);
		g_origin.s1 = (uint32_t)sd1; g_origin.s2 = (uint32_t)sd2; g_origin.s3 = (uint32_t)sd3;
		ky_SP = (g_allo ? 2 : 1) * (16 * (uint64_t)Amax + 16); ky_SZ = 4 * (uint64_t)L + 16 * (uint64_t)K + 16;
		ky_G0 = 1; ky_ZI0 = 1 + amb; ky_B0 = ky_ZI0 + (uint64_t)N * ky_SZ;
		ky_offS = (uint64_t)K * L * ky_SP; ky_offZ = ky_offS + 4 * (uint64_t)K; ky_offGE = ky_offZ + (uint64_t)N * ky_SZ;
		ky_BLK = ky_offGE + amb + 4;
	}
	alpha = ran1() * 10;
	fprintf(G, "chain init alpha=%a", alpha); SEEDS();
	initial_geno();
	GFLAT();
	fprintf(G, "chain genoinit hgeno=%016llx", (unsigned long long)hash_i32v(gflat, (long)N * L * P4)); SEEDS();
	for (i = 0; i < K; i++) { S[i] = initd[i]; if (back_refl == 0) state[i] = dt_stat(S[i]); }
	update_ZQ(1);
	fprintf(G, "chain zqinit hz=%016llx hqq=%016llx", (unsigned long long)hash_z(&D, z), (unsigned long long)hash_f64v(qq, (long)N * K)); SEEDS();
	for (step = 0; step < u; step++) {
		ky_iter = (uint64_t)step;
		if (g_allo) update_P_allo();
		else update_P_auto();
		GFLAT();
		count_alleles_plain(&D, gflat, z, cflat);
		fprintf(G, "it %ld P hcnt=%016llx hfreq=%016llx", step, (unsigned long long)hash_counts(&D, cflat), (unsigned long long)hash_freq(&D, freq));
		if (g_allo) fprintf(G, " hfreq2=%016llx", (unsigned long long)hash_freq(&D, freq2));
		SEEDS();
		calc_exfreq_auto();
		fprintf(G, "it %ld X hexfreq=%016llx\n", step, (unsigned long long)hash_tables(exfreq));
		update_S_POP();
		fprintf(G, "it %ld S", step);
		for (i = 0; i < K; i++) fprintf(G, " %a", S[i]);
		if (back_refl == 0) for (i = 0; i < K; i++) fprintf(G, " st%d", state[i]);
		fprintf(G, " hgenofreq=%016llx", (unsigned long long)hash_tables(genofreq)); SEEDS();
		update_ZQ(0);
		fprintf(G, "it %ld ZQ hz=%016llx hqq=%016llx hqqnum=%016llx", step, (unsigned long long)hash_z(&D, z),
			(unsigned long long)hash_f64v(qq, (long)N * K), (unsigned long long)hash_f64v(qqnum, (long)N * K)); SEEDS();
		update_geno();
		GFLAT();
		fprintf(G, "it %ld GE hgeno=%016llx", step, (unsigned long long)hash_i32v(gflat, (long)N * L * P4)); SEEDS();
		totallkh = cal_lkd();
		fprintf(G, "it %ld L totallkh=%a hindv=%016llx\n", step, totallkh, (unsigned long long)hash_f64v(indvlkh, N));
		if (step == b - 1) {
			c_step = 0; c_tot = 1; c_tot2 = 1;
			for (i = 0; i < N; i++) c_indv[i] = 1;
			for (i = 0; i < K; i++) c_S[i] = 1;
			for (i = 0; i < N * K; i++) c_qq[i] = 1;
		}
		if (step >= b && (step + 1 - b) % t == 0) {
#define RM(m, x) do { double *m_ = &(m), x_ = (x); if (*m_ != 0) *m_ = *m_ * ((c_step + x_ / *m_) / (1 + c_step)); else *m_ = x_ / (1 + c_step); } while (0)
			RM(c_tot, totallkh); RM(c_tot2, totallkh * totallkh);
			for (i = 0; i < N; i++) RM(c_indv[i], indvlkh[i]);
			for (i = 0; i < N * K; i++) RM(c_qq[i], qq[i]);
			for (i = 0; i < K; i++) RM(c_S[i], S[i]);
			c_step++;
			if (cnt_step < r) convg[cnt_step] = totallkh;
			cnt_step++;
		}
		if (cnt_step == jj) {
			for (k = 0; k < K && !flag_empty; k++) {
				double sum = 0;
				for (i = 0; i < N; i++) sum += qq[(long)i * K + k];
				if (sum < 0.01) flag_empty = 1;
			}
			if (flag_empty) { fprintf(G, "empty_cluster at step %ld\n", step); break; }
		}
	}
	fprintf(G, "chain done"); SEEDS();
	fprintf(G, "chain steps=%ld step=%ld flag_empty=%d totallkh=%a totallkh2=%a\n", steps, c_step, flag_empty, c_tot, c_tot2);
	dump_vec(G, "chain indvlkh", c_indv, N);
	dump_vec(G, "chain self_rates", c_S, K);
	for (i = 0; i < N; i++) {
		fprintf(G, "chain qq %d", i);
		for (k = 0; k < K; k++) fprintf(G, " %a", c_qq[i * K + k]);
		fprintf(G, "\n");
	}
	dump_vec(G, "convg", convg, r);
	fclose(G);
	return err_flag;
}
