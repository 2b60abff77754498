/*
 * synth_fast.c -- the benchmark's synthetic tetraploid population in C: the same arrays instruct_amd/synth.py builds with numpy
 * (raw_alleles + code_tetraploid: counter-based splitmix64 uniforms, allele labels from each cluster's cumulative frequencies, the
 * coding rules of transform_data2 / get_missing_tetra, data_interface.c:571-669, 722-741), sized for config 5 (8e8 allele copies,
 * where the numpy coder takes minutes).  Benchmark / test input only: nothing of the sampler depends on it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t x)
{
	uint64_t z = x + 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
static inline double uniform(uint64_t key, uint64_t idx) { return (double)(splitmix64(idx ^ key) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint64_t stream_key(uint64_t seed, uint64_t stream) { return splitmix64(seed + stream * 0xD1B54A32D192ED03ULL); }

/*
 * cum: [K][L][A] cumulative allele frequencies per (cluster, locus) (formed by the caller exactly as synth.raw_alleles does).
 * Out: obs int32 [N][L][4] sorted distinct allele codes (-1 padded), alleleid int32 [N][L] (0 = missing), allelenum int32 [L].
 * Returns 0, or 1 when memory runs out / a label is out of range.
 */
int isg_synth_tetraploid(long N, long L, int K, int A, double missing_frac, uint64_t seed, const double *cum, int32_t *obs, int32_t *alleleid, int32_t *allelenum)
{
	const int P = 4;
	const uint64_t k2 = stream_key(seed, 2), k4 = stream_key(seed, 4);
	int8_t *raw = (int8_t *)malloc((size_t)N * L * P);
	int64_t *first = (int64_t *)malloc(sizeof(int64_t) * (size_t)A * L);
	int32_t *code = (int32_t *)malloc(sizeof(int32_t) * (size_t)A * L);
	long i, j;
	int k, a;
	const int64_t big = (int64_t)N * P;
	if (!raw || !first || !code || A > 30) { free(raw); free(first); free(code); return 1; }
	for (j = 0; j < (long)A * L; j++) first[j] = big;
	/* labels 0 .. A-1 (synth.py's are 1-based), -1 missing; first appearance per (label, locus) scanning individuals, then copies
	 * (threads take individuals in turn and keep their own "first" tables, merged by minimum: the same result in any order) */
	int bad = 0;
#pragma omp parallel
	{
		int64_t *mine = (int64_t *)malloc(sizeof(int64_t) * (size_t)A * L);
		long ii, jj;
		int kk, aa;
		if (!mine) {
#pragma omp atomic write
			bad = 1;
		} else {
			for (jj = 0; jj < (long)A * L; jj++) mine[jj] = big;
#pragma omp for schedule(static)
			for (ii = 0; ii < N; ii++) {
				const double *c = cum + (size_t)(ii % K) * L * A;
				for (jj = 0; jj < L; jj++) {
					int8_t *r = raw + ((size_t)ii * L + jj) * P;
					const int miss = missing_frac > 0.0 && uniform(k4, (uint64_t)ii * (uint64_t)L + (uint64_t)jj) < missing_frac;
					for (kk = 0; kk < P; kk++) {
						const double u = uniform(k2, ((uint64_t)ii * (uint64_t)L + (uint64_t)jj) * P + (uint64_t)kk);
						int v = 0;
						for (aa = 0; aa < A; aa++) v += (u >= c[jj * A + aa]);
						if (v >= A) {
#pragma omp atomic write
							bad = 1;
							v = A - 1;
						}
						r[kk] = miss ? -1 : (int8_t)v;
						if (!miss && (int64_t)ii * P + kk < mine[(size_t)v * L + jj]) mine[(size_t)v * L + jj] = (int64_t)ii * P + kk;
					}
				}
			}
#pragma omp critical
			for (jj = 0; jj < (long)A * L; jj++)
				if (mine[jj] < first[jj]) first[jj] = mine[jj];
			free(mine);
		}
	}
	if (bad) { free(raw); free(first); free(code); return 1; }
	/* codes in order of first appearance (stable: ties cannot occur among labels that appear) */
	for (j = 0; j < L; j++) {
		int n = 0;
		for (a = 0; a < A; a++) {
			int rank = 0, b;
			if (first[(size_t)a * L + j] < big) n++;
			for (b = 0; b < A; b++)
				if (first[(size_t)b * L + j] < first[(size_t)a * L + j] || (first[(size_t)b * L + j] == first[(size_t)a * L + j] && b < a)) rank++;
			code[(size_t)a * L + j] = rank;
		}
		allelenum[j] = n;
	}
#pragma omp parallel for schedule(static) private(j, k, a)
	for (i = 0; i < N; i++)
		for (j = 0; j < L; j++) {
			const int8_t *r = raw + ((size_t)i * L + j) * P;
			int32_t *o = obs + ((size_t)i * L + j) * P;
			unsigned mask = 0;
			int n = 0;
			for (k = 0; k < P; k++)
				if (r[k] >= 0) mask |= 1u << code[(size_t)r[k] * L + j];
			for (a = 0; a < A; a++)
				if ((mask >> a) & 1u) o[n++] = a;
			alleleid[(size_t)i * L + j] = n;
			for (; n < P; n++) o[n] = -1;
		}
	free(raw);
	free(first);
	free(code);
	return 0;
}
