/*
 * isg_poly_tables.h -- genotype-class tables of the autotetraploid sampler (reference poly_geno.c).
 *
 * For a locus with n alleles a tetraploid genotype falls in one of five classes (iiii, iiij, iijj, iijk,
 * ijkl; poly_geno.c:1698-1713); all G(n) genotypes are listed in a fixed order, encoded as base-n
 * numbers (auto_geno_list, poly_geno.c:1716-1800).  Two float tables per (cluster, locus) drive the
 * sampler:
 *   exfreq   log expected (panmictic) genotype frequency          calc_exfreq_auto  poly_geno.c:1515-1590
 *   genfreq  log genotype frequency at selfing rate s, solved class by class from quadri-allelic down
 *            to mono-allelic genotypes (3x3 float Gauss-Jordan for the tri-allelic triples)
 *                                                                   auto_genfreq      poly_geno.c:1803-2028
 * The arithmetic keeps the reference's float/double promotion pattern (the tables are float,
 * poly_geno.h:19-20; libm calls are double), including its quirks (the repeated test at :1990-1996).
 *
 * This is a TEMPLATE header: include it with PT_NAME(x), PT_LOG(x), PT_EXP(x) defined.  The product
 * (host and device) instantiates it with the bit-reproducible isg_math.h functions; the CPU oracle
 * instantiates it a second time with glibc's log/exp, and THAT instance is pinned byte-for-byte to the
 * reference's golden trajectories (tests/golden/t*.golden), which is what validates this code.
 */
#include "isg_math.h"

#ifndef ISG_POLY_COMMON
#define ISG_POLY_COMMON
typedef struct {
	int n;        /* alleles at the locus */
	int G;        /* genotypes in total */
	int g[6];     /* g[1..5]: mono, simplex, duplex, tri, quadri (genonum row, poly_geno.c:1706-1713) */
	const int *list; /* [G] base-n codes in table order */
} isg_polyclass;

ISG_HD int isg_poly_G(int i) { return i + i * (i - 1) * 3 / 2 + i * (i - 1) * (i - 2) / 2 + i * (i - 1) * (i - 2) * (i - 3) / 24; }

/* auto_geno_num + auto_geno_list for one allele count (host side table construction) */
static inline void isg_poly_build(int i, int g[6], int *list)
{
	int j, k, m, n, cnt, tmp;
	g[1] = i;
	g[2] = i * (i - 1);
	g[3] = i * (i - 1) / 2;
	g[4] = i * (i - 1) * (i - 2) / 2;
	g[5] = i * (i - 1) * (i - 2) * (i - 3) / 24;
	g[0] = isg_poly_G(i);
	for (j = 0; j < g[1]; j++) list[j] = j * (i * i * i + i * i + i + 1);
	tmp = g[1];
	cnt = 0;
	for (j = 0; j < i - 1; j++)
		for (k = j + 1; k < i; k++) {
			list[tmp + 2 * cnt] = j * (i * i * i + i * i + i) + k;
			list[tmp + 2 * cnt + 1] = i * (i * i + i + 1) * k + j;
			cnt++;
		}
	tmp += g[2];
	cnt = 0;
	for (j = 0; j < i - 1; j++)
		for (k = j + 1; k < i; k++) list[tmp + cnt++] = j * (i * i * i + i * i) + k * (i + 1);
	tmp += g[3];
	cnt = 0;
	for (j = 0; j < i - 2; j++)
		for (k = j + 1; k < i - 1; k++)
			for (m = k + 1; m < i; m++) {
				list[tmp + 3 * cnt] = j * (i * i * i + i * i) + k * i + m;
				list[tmp + 3 * cnt + 1] = k * (i * i * i + i * i) + j * i + m;
				list[tmp + 3 * cnt + 2] = m * (i * i * i + i * i) + j * i + k;
				cnt++;
			}
	tmp += g[4];
	cnt = 0;
	for (j = 0; j < i - 3; j++)
		for (k = j + 1; k < i - 2; k++)
			for (m = k + 1; m < i - 1; m++)
				for (n = m + 1; n < i; n++) list[tmp + cnt++] = j * i * i * i + k * i * i + i * m + n;
}

ISG_HD int isg_poly_exists(int value, const int *vec, int leng) /* data_interface.c:865-877 */
{
	int i, flag = 0;
	for (i = 0; i < leng; i++)
		if (value == vec[i]) flag = 1;
	return flag;
}
ISG_HD int isg_poly_find(int num, const int *array, int len, int *err) /* find_id, poly_geno.c:2367-2381 */
{
	int i;
	for (i = 0; i < len; i++)
		if (array[i] == num) return i;
	*err |= 1;
	return 0;
}
ISG_HD int isg_poly_calc_val(const int *num, int val, int i) /* poly_geno.c:2305-2330 */
{
	int temp = 0;
	if (val < num[0]) temp = val * i * i * i + num[0] * i * i + num[1] * i + num[2];
	else if (val > num[0] && val < num[1]) temp = num[0] * i * i * i + val * i * i + num[1] * i + num[2];
	else if (val > num[1] && val < num[2]) temp = num[0] * i * i * i + num[1] * i * i + val * i + num[2];
	else if (val > num[2]) temp = num[0] * i * i * i + num[1] * i * i + num[2] * i + val;
	return temp;
}
ISG_HD int isg_poly_calc_val2(const int *num, int val1, int val2, int i) /* poly_geno.c:2332-2365 */
{
	int temp = 0;
	if (val2 < num[1]) temp = val1 * i * i * i + val2 * i * i + num[1] * i + num[0];
	else if (val2 > num[1] && val2 < num[0] && val1 < num[1]) temp = val1 * i * i * i + num[1] * i * i + val2 * i + num[0];
	else if (val1 > num[1] && val2 < num[0]) temp = num[1] * i * i * i + val1 * i * i + val2 * i + num[0];
	else if (val1 > num[1] && val1 < num[0] && val2 > num[0]) temp = num[1] * i * i * i + val1 * i * i + num[0] * i + val2;
	else if (val1 > num[0]) temp = num[1] * i * i * i + num[0] * i * i + val1 * i + val2;
	else if (val1 < num[1] && val2 > num[0]) temp = val1 * i * i * i + num[1] * i * i + i * num[0] + val2;
	return temp;
}
/* gaussj (poly_geno.c:2384-2435) for the 3x3 system with one right-hand side, float, 1-based */
ISG_HD void isg_poly_gaussj3(float a[4][4], float b[4], int *err)
{
	int indxc[4], indxr[4], ipiv[4], i, icol = 1, irow = 1, j, k, l, ll;
	const int n = 3;
	float big, dum, pivinv, temp;
	for (j = 1; j <= n; j++) ipiv[j] = 0;
	for (i = 1; i <= n; i++) {
		big = 0.0f;
		for (j = 1; j <= n; j++)
			if (ipiv[j] != 1)
				for (k = 1; k <= n; k++)
					if (ipiv[k] == 0) {
						const float av = a[j][k] < 0 ? -a[j][k] : a[j][k];
						if (av >= big) { big = av; irow = j; icol = k; }
					}
		++(ipiv[icol]);
		if (irow != icol) {
			for (l = 1; l <= n; l++) { temp = a[irow][l]; a[irow][l] = a[icol][l]; a[icol][l] = temp; }
			temp = b[irow]; b[irow] = b[icol]; b[icol] = temp;
		}
		indxr[i] = irow;
		indxc[i] = icol;
		if (a[icol][icol] == 0.0f) { *err |= 2; return; }
		pivinv = (float)(1.0 / a[icol][icol]);
		a[icol][icol] = 1.0f;
		for (l = 1; l <= n; l++) a[icol][l] *= pivinv;
		b[icol] *= pivinv;
		for (ll = 1; ll <= n; ll++)
			if (ll != icol) {
				dum = a[ll][icol];
				a[ll][icol] = 0.0f;
				for (l = 1; l <= n; l++) a[ll][l] -= a[icol][l] * dum;
				b[ll] -= b[icol] * dum;
			}
	}
	for (l = n; l >= 1; l--)
		if (indxr[l] != indxc[l])
			for (k = 1; k <= n; k++) { temp = a[k][indxr[l]]; a[k][indxr[l]] = a[k][indxc[l]]; a[k][indxc[l]] = temp; }
}

/* genotype category of a 4-copy genotype (get_cat_auto, poly_geno.c:1313-1339): 0 iiii, 1 iiij, 2 iijj, 3 iijk, 4 ijkl */
ISG_HD int isg_poly_cat(const int *g)
{
	int i, cnt = 0, tmp[4], c0 = 0;
	tmp[cnt++] = g[0];
	for (i = 1; i < 4; i++)
		if (!isg_poly_exists(g[i], tmp, cnt)) tmp[cnt++] = g[i];
	if (cnt == 1) return 0;
	if (cnt == 3) return 3;
	if (cnt == 4) return 4;
	for (i = 0; i < 4; i++) c0 += (g[i] == tmp[0]);
	return c0 == 2 ? 2 : 1;
}

/*
 * Allotetraploid (-ap 0): a genotype is a pair of diploid genotypes, one per subgenome -- copies 0, 1 from the first
 * (allele frequencies freq), copies 2, 3 from the second (freq2).  Four classes for a locus with n alleles, in table order
 * (allo_geno_num / allo_geno_list, poly_geno.c:2031-2119):  0 iikk (both homozygous), 1 iikl (second heterozygous),
 * 2 ijkk (first heterozygous), 3 ijkl (both).  A canonical genotype (g0 <= g1, g2 <= g3) has a closed-form row:
 * with C = n (n - 1) / 2 and pair(a, b) = the rank of a < b among the pairs in lexicographic order,
 *   iikk  g0 n + g2 | iikl  n^2 + g0 C + pair(g2, g3) | ijkk  n^2 + n C + pair(g0, g1) n + g2 | ijkl  n^2 + 2 n C + pair(g0, g1) C + pair(g2, g3)
 * -- the reference finds rows by linear search of the base-n code (find_id, poly_geno.c:2367-2381); same rows.
 */
ISG_HD int isg_allo_G(int n) { return n * n + n * (n - 1) * n + n * (n - 1) * n * (n - 1) / 4; }
ISG_HD int isg_allo_pair(int a, int b, int n) { return a * n - a * (a + 1) / 2 + (b - a - 1); } /* a < b */
ISG_HD int isg_allo_row(int n, int g0, int g1, int g2, int g3) /* canonical genotype -> table row */
{
	const int C = n * (n - 1) / 2, hetA = g0 != g1, hetB = g2 != g3;
	if (!hetA && !hetB) return g0 * n + g2;
	if (!hetA) return n * n + g0 * C + isg_allo_pair(g2, g3, n);
	if (!hetB) return n * n + n * C + isg_allo_pair(g0, g1, n) * n + g2;
	return n * n + 2 * n * C + isg_allo_pair(g0, g1, n) * C + isg_allo_pair(g2, g3, n);
}
ISG_HD int isg_allo_row_any(int n, int a, int b, int c, int d) /* each pair in either order */
{
	return isg_allo_row(n, a < b ? a : b, a < b ? b : a, c < d ? c : d, c < d ? d : c);
}
/* get_cat_allo (poly_geno.c:1341-1372) */
ISG_HD int isg_allo_cat(const int *g) { return (g[0] != g[1] ? 2 : 0) + (g[2] != g[3] ? 1 : 0); }
/* class sizes and the base-n codes in table order (g[1..4]; g[5] = 0) */
static inline void isg_allo_build(int n, int g[6], int *list)
{
	int a, b, c, d, r = 0;
	const int C = n * (n - 1) / 2;
	g[1] = n * n; g[2] = n * C; g[3] = n * C; g[4] = C * C; g[5] = 0;
	g[0] = isg_allo_G(n);
	for (a = 0; a < n; a++)
		for (c = 0; c < n; c++) list[r++] = ((a * n + a) * n + c) * n + c;
	for (a = 0; a < n; a++)
		for (c = 0; c < n - 1; c++)
			for (d = c + 1; d < n; d++) list[r++] = ((a * n + a) * n + c) * n + d;
	for (a = 0; a < n - 1; a++)
		for (b = a + 1; b < n; b++)
			for (c = 0; c < n; c++) list[r++] = ((a * n + b) * n + c) * n + c;
	for (a = 0; a < n - 1; a++)
		for (b = a + 1; b < n; b++)
			for (c = 0; c < n - 1; c++)
				for (d = c + 1; d < n; d++) list[r++] = ((a * n + b) * n + c) * n + d;
}
#endif /* ISG_POLY_COMMON */

/* ---- instantiated part: needs PT_NAME, PT_LOG, PT_EXP ---- */

/* calc_exfreq_auto for one (cluster, locus): f = allele frequencies of the cluster at the locus */
ISG_HD void PT_NAME(exfreq_row)(const isg_polyclass *pc, const double *f, float *ex)
{
	const int n = pc->n, P = 4;
	int j, m, digit[4], temp, tmp;
	for (j = 0; j < pc->g[1]; j++) {
		tmp = pc->list[j];
		digit[0] = tmp % n;
		ex[j] = (float)PT_LOG(f[digit[0]]) * (float)P;
	}
	temp = pc->g[1];
	for (j = temp; j < temp + pc->g[2]; j++) {
		tmp = pc->list[j];
		digit[0] = tmp % n;
		tmp /= n;
		digit[1] = tmp % n;
		ex[j] = (float)(PT_LOG(4.0) + PT_LOG(f[digit[1]]) * (float)(P - 1) + PT_LOG(f[digit[0]]));
	}
	temp += pc->g[2];
	for (j = temp; j < temp + pc->g[3]; j++) {
		tmp = pc->list[j];
		digit[0] = tmp % n;
		tmp /= (n * n);
		digit[1] = tmp % n;
		ex[j] = (float)(PT_LOG(6.0) + (PT_LOG(f[digit[1]]) + PT_LOG(f[digit[0]])) * (P / 2));
	}
	temp += pc->g[3];
	for (j = temp; j < temp + pc->g[4]; j++) {
		tmp = pc->list[j];
		for (m = 0; m < P - 1; m++) { digit[m] = tmp % n; tmp /= n; }
		ex[j] = (float)(PT_LOG(12.0) + PT_LOG(f[digit[2]]) * (P / 2) + PT_LOG(f[digit[0]]) + PT_LOG(f[digit[1]]));
	}
	temp += pc->g[4];
	for (j = temp; j < temp + pc->g[5]; j++) {
		tmp = pc->list[j];
		for (m = 0; m < P; m++) { digit[m] = tmp % n; tmp /= n; }
		ex[j] = (float)PT_LOG(24.0);
		for (m = 0; m < P; m++) ex[j] += (float)PT_LOG(f[digit[m]]);
	}
}

/* auto_genfreq for one (cluster, locus): ex = its exfreq row, fr = output row; *err |= 4 when a log
 * frequency comes out positive (the reference aborts: "Genotype frequencies can not be greater than 1!") */
ISG_HD void PT_NAME(genfreq_row)(float self, const isg_polyclass *pc, const float *ex, float *fr, int *err)
{
	const int n = pc->n, G = pc->G, tri = 3, P = 4;
	const int *gl = pc->list;
	int i, j, k, l, tmp, digit[3], num = 0;
	float temp, matr[4][4], vec[4];
	tmp = G;
	if (n >= 4)
		for (i = tmp - pc->g[5]; i < tmp; i++) {
			fr[i] = (float)(PT_LOG((double)(1 - self)) + ex[i] - PT_LOG((double)(1 - self / 6)));
			if (fr[i] > 0) *err |= 4;
		}
	if (n >= 3) {
		tmp -= pc->g[5];
		for (i = 0; i < pc->g[4] / tri; i++) {
			num = gl[tmp - pc->g[4] + i * 3];
			for (j = P - 2; j >= 0; j--) { digit[j] = num % n; num /= n; }
			temp = 0;
			if (n >= 4) {
				for (l = 0; l < n; l++)
					if (isg_poly_exists(l, digit, tri) == 0) {
						num = isg_poly_find(isg_poly_calc_val(digit, l, n), gl, G, err);
						temp = (float)(temp + PT_EXP((double)fr[num]));
					}
				if (temp > 1) *err |= 4;
			}
			for (j = 1; j <= tri; j++) {
				for (k = 1; k <= tri; k++) {
					if (j == k) matr[j][k] = (float)(1 - self * 10.0 / 36.0);
					else matr[j][k] = (float)(-self / 9.0);
				}
				vec[j] = (float)(self / 18.0 * temp + (1.0 - self) * PT_EXP((double)ex[tmp - pc->g[4] + i * 3 + j - 1]));
			}
			temp = vec[1];
			for (j = 1; j <= tri; j++) vec[j] /= temp;
			isg_poly_gaussj3(matr, vec, err);
			for (j = 0; j < tri; j++) {
				fr[tmp - pc->g[4] + i * 3 + j] = (float)(PT_LOG((double)vec[j + 1]) + PT_LOG((double)temp));
				if (fr[tmp - pc->g[4] + i * 3 + j] > 0) *err |= 4;
			}
		}
	}
	tmp -= pc->g[4];
	for (i = tmp - pc->g[3]; i < tmp; i++) { /* duplex iijj */
		num = gl[i];
		digit[0] = num % n;
		num /= (n * n);
		digit[1] = num % n;
		temp = 0;
		if (n >= 3)
			for (j = 0; j < n; j++)
				if (isg_poly_exists(j, digit, 2) == 0) {
					if (digit[0] < j) num = isg_poly_find(digit[1] * n * n * (n + 1) + digit[0] * n + j, gl, G, err);
					else if (digit[0] > j) num = isg_poly_find(digit[1] * n * n * (n + 1) + j * n + digit[0], gl, G, err);
					temp = (float)(temp + PT_EXP((double)fr[num]) / 9.0 * self);
					if (digit[1] < j) num = isg_poly_find(digit[0] * n * n * (n + 1) + digit[1] * n + j, gl, G, err);
					else if (digit[1] > j) num = isg_poly_find(digit[0] * n * n * (n + 1) + j * n + digit[1], gl, G, err);
					temp = (float)(temp + PT_EXP((double)fr[num]) / 9.0 * self);
					num = isg_poly_find(j * n * n * (n + 1) + digit[1] * n + digit[0], gl, G, err);
					temp = (float)(temp + PT_EXP((double)fr[num]) / 36.0 * self);
					if (n >= 4)
						for (k = j + 1; k < n; k++)
							if (isg_poly_exists(k, digit, 2) == 0) {
								num = isg_poly_find(isg_poly_calc_val2(digit, j, k, n), gl, G, err);
								temp = (float)(temp + PT_EXP((double)fr[num]) / 36.0 * self);
							}
				}
		fr[i] = (float)(PT_LOG((1 - self) * PT_EXP((double)ex[i]) + temp) - PT_LOG(1 - self / 2.0));
		if (fr[i] > 0) *err |= 4;
	}
	tmp -= pc->g[3];
	for (i = tmp - pc->g[2]; i < tmp; i++) { /* simplex iiij */
		num = gl[i];
		digit[0] = num % n;
		num /= n;
		digit[1] = num % n;
		if (digit[0] < digit[1]) num = isg_poly_find((digit[0] * n * n + digit[1]) * (n + 1), gl, G, err);
		else if (digit[0] > digit[1]) num = isg_poly_find((digit[1] * n * n + digit[0]) * (n + 1), gl, G, err);
		temp = (float)(8.0 / 36.0 * PT_EXP((double)fr[num]) * self);
		if (n >= 3)
			for (j = 0; j < n; j++)
				if (isg_poly_exists(j, digit, 2) == 0) {
					if (digit[0] < j) num = isg_poly_find(digit[1] * n * n * (n + 1) + digit[0] * n + j, gl, G, err);
					else if (digit[0] > j) num = isg_poly_find(digit[1] * n * n * (n + 1) + j * n + digit[0], gl, G, err);
					temp = (float)(temp + PT_EXP((double)fr[num]) / 9.0 * self);
				}
		fr[i] = (float)(PT_LOG((1 - self) * PT_EXP((double)ex[i]) + temp) - PT_LOG(1 - self / 2.0));
		if (fr[i] > 0) *err |= 4;
	}
	tmp -= pc->g[2];
	for (i = tmp - pc->g[1]; i < tmp; i++) { /* mono iiii */
		num = gl[i];
		digit[0] = num % n;
		temp = 0;
		for (j = 0; j < n; j++)
			if (j != digit[0]) {
				num = isg_poly_find(digit[0] * n * (n * n + n + 1) + j, gl, G, err);
				temp = (float)(temp + PT_EXP((double)fr[num]) / 4.0 * self);
				/* sic: for digit[0] > j the reference repeats the digit[0] < j test in its else branch, so the
				 * duplex term reuses the simplex index found above (poly_geno.c:1990-1996) */
				if (digit[0] < j) num = isg_poly_find(digit[0] * n * n * (n + 1) + j * (n + 1), gl, G, err);
				temp = (float)(temp + PT_EXP((double)fr[num]) / 36.0 * self);
				if (n >= 3)
					for (k = j + 1; k < n; k++)
						if (k != digit[0]) {
							num = isg_poly_find(digit[0] * n * n * (n + 1) + j * n + k, gl, G, err);
							temp = (float)(temp + PT_EXP((double)fr[num]) / 36.0 * self);
						}
			}
		fr[i] = (float)(PT_LOG((1 - self) * PT_EXP((double)ex[i]) + temp) - PT_LOG((double)(1 - self)));
		if (fr[i] > 0) *err |= 4;
	}
}

/* calc_exfreq_allo (poly_geno.c:1592-1670) for one (cluster, locus): f / f2 = the cluster's allele frequencies in the two
 * subgenomes.  Copies 0, 1 (digits 3, 2 of the code) take f, copies 2, 3 take f2; float / double exactly as the reference. */
ISG_HD void PT_NAME(exfreq_row_allo)(const isg_polyclass *pc, const double *f, const double *f2, float *ex)
{
	const int n = pc->n;
	int r;
	for (r = 0; r < pc->G; r++) {
		const int code = pc->list[r], d0 = code % n, d1 = (code / n) % n, d2 = (code / n / n) % n, d3 = code / n / n / n;
		if (r < pc->g[1]) ex[r] = (float)((PT_LOG(f[d2]) + PT_LOG(f2[d0])) * 2);
		else if (r < pc->g[1] + pc->g[2]) ex[r] = (float)(PT_LOG(2.0) + PT_LOG(f[d2]) * 2 + PT_LOG(f2[d0]) + PT_LOG(f2[d1]));
		else if (r < pc->g[1] + pc->g[2] + pc->g[3]) ex[r] = (float)(PT_LOG(2.0) + PT_LOG(f2[d1]) * 2 + PT_LOG(f[d3]) + PT_LOG(f[d2]));
		else {
			ex[r] = (float)PT_LOG(4.0);
			ex[r] += (float)PT_LOG(f2[d0]);
			ex[r] += (float)PT_LOG(f2[d1]);
			ex[r] += (float)PT_LOG(f[d2]);
			ex[r] += (float)PT_LOG(f[d3]);
		}
	}
}

/* allo_genfreq (poly_geno.c:2122-2304) for one (cluster, locus): log genotype frequencies at selfing rate `self`, class by
 * class from ijkl down to iikk (each class only needs the ones solved before it); *err |= 4 when one comes out positive */
ISG_HD void PT_NAME(genfreq_row_allo)(float self, const isg_polyclass *pc, const float *ex, float *fr, int *err)
{
	const int n = pc->n, b1 = pc->g[1], b2 = b1 + pc->g[2], b3 = b2 + pc->g[3];
	int r, v, w;
	float temp;
	for (r = b3; r < pc->G; r++) { /* ijkl */
		fr[r] = (float)(PT_LOG((double)(1 - self)) + ex[r] - PT_LOG((double)(1 - self / 4)));
		if (fr[r] > 0) *err |= 4;
	}
	for (r = b2; r < b3; r++) { /* ijkk: the second subgenome's homozygote comes from selfed heterozygotes k v */
		const int code = pc->list[r], k = code % n, b = (code / n / n) % n, a = code / n / n / n;
		temp = 0;
		for (v = 0; v < n; v++)
			if (v != k) temp = (float)(temp + PT_EXP((double)fr[isg_allo_row_any(n, a, b, k, v)]) * self / 8.0);
		fr[r] = (float)(PT_LOG((1 - self) * PT_EXP((double)ex[r]) + temp) - PT_LOG(1 - self / 2.0));
		if (fr[r] > 0) *err |= 4;
	}
	for (r = b1; r < b2; r++) { /* iikl */
		const int code = pc->list[r], d = code % n, c = (code / n) % n, a = (code / n / n) % n;
		temp = 0;
		for (v = 0; v < n; v++)
			if (v != a) temp = (float)(temp + PT_EXP((double)fr[isg_allo_row_any(n, a, v, c, d)]) * self / 8.0);
		fr[r] = (float)(PT_LOG((1 - self) * PT_EXP((double)ex[r]) + temp) - PT_LOG(1 - self / 2.0));
		if (fr[r] > 0) *err |= 4;
	}
	for (r = 0; r < b1; r++) { /* iikk */
		const int code = pc->list[r], k = code % n, a = (code / n / n) % n;
		temp = 0;
		for (v = 0; v < n; v++)
			if (v != k) temp = (float)(temp + PT_EXP((double)fr[isg_allo_row_any(n, a, a, k, v)]) * self / 4.0);
		for (v = 0; v < n; v++)
			if (v != a) temp = (float)(temp + PT_EXP((double)fr[isg_allo_row_any(n, a, v, k, k)]) * self / 4.0);
		for (v = 0; v < n; v++)
			for (w = 0; w < n; w++)
				if (v != a && w != k) temp = (float)(temp + PT_EXP((double)fr[isg_allo_row_any(n, a, v, k, w)]) * self / 16.0);
		fr[r] = (float)(PT_LOG((1 - self) * PT_EXP((double)ex[r]) + temp) - PT_LOG((double)(1 - self)));
		if (fr[r] > 0) *err |= 4;
	}
}
