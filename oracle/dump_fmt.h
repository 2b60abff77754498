/*
 * dump_fmt.h -- TEST INFRASTRUCTURE (oracle/): the line format of the golden trajectory files
 * under tests/golden/.  Shared by oracle/ref_dump.c (which drives the REAL reference sweeps,
 * built only in this container into oracle/_ref/) and by the CPU restatement's own dump tool
 * (oracle/orc_dump.c), so that "the restatement reproduces the reference" is a byte-for-byte
 * file comparison.
 *
 * All state is handed over as flat arrays:
 *   z      int  [N][L][P]   (entries of invalid loci are hashed as -1: the reference leaves
 *                            them uninitialised, mcmc.c:536 + mcmc.c:1137)
 *   freq   double [K][L][Amax]  (only a < allelenum[j] hashed)
 *   qq     double [N][K], qqnum double [N][K], gen int [N], S double [K], indv double [N]
 */
#ifndef ISG_DUMP_FMT_H
#define ISG_DUMP_FMT_H
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline uint64_t fnv_init(void) { return 0xcbf29ce484222325ULL; }
static inline uint64_t fnv_bytes(uint64_t h, const void *p, size_t n)
{
	const unsigned char *b = (const unsigned char *)p;
	size_t i;
	for (i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ULL; }
	return h;
}
static inline uint64_t fnv_i32(uint64_t h, int32_t v) { return fnv_bytes(h, &v, 4); }
static inline uint64_t fnv_f64(uint64_t h, double v)
{
	uint64_t u;
	memcpy(&u, &v, 8);
	return fnv_bytes(h, &u, 8);
}

typedef struct {
	int N, L, P, K, Amax;
	const int *allelenum; /* [L] */
	const int *valid;     /* [N][L] 1 = locus used for this individual */
} dump_dims;

static inline uint64_t hash_z(const dump_dims *d, const int *z)
{
	uint64_t h = fnv_init();
	long i, j, k;
	for (i = 0; i < d->N; i++)
		for (j = 0; j < d->L; j++)
			for (k = 0; k < d->P; k++)
				h = fnv_i32(h, d->valid[i * d->L + j] ? z[(i * d->L + j) * d->P + k] : -1);
	return h;
}
static inline uint64_t hash_freq(const dump_dims *d, const double *freq)
{
	uint64_t h = fnv_init();
	long k, j, a;
	for (k = 0; k < d->K; k++)
		for (j = 0; j < d->L; j++)
			if (d->allelenum[j] > 1)
				for (a = 0; a < d->allelenum[j]; a++)
					h = fnv_f64(h, freq[(k * d->L + j) * d->Amax + a]);
	return h;
}
static inline uint64_t hash_counts(const dump_dims *d, const int *cnt)
{
	uint64_t h = fnv_init();
	long k, j, a;
	for (k = 0; k < d->K; k++)
		for (j = 0; j < d->L; j++)
			for (a = 0; a < d->allelenum[j]; a++)
				h = fnv_i32(h, cnt[(k * d->L + j) * d->Amax + a]);
	return h;
}
static inline uint64_t hash_f64v(const double *v, long n)
{
	uint64_t h = fnv_init();
	long i;
	for (i = 0; i < n; i++) h = fnv_f64(h, v[i]);
	return h;
}
static inline uint64_t hash_i32v(const int *v, long n)
{
	uint64_t h = fnv_init();
	long i;
	for (i = 0; i < n; i++) h = fnv_i32(h, v[i]);
	return h;
}

/* allele counts seqpop[K][L][Amax] from z and geno (definition of mcmc.c:815-845) */
static inline void count_alleles_plain(const dump_dims *d, const int *geno, const int *z, int *cnt)
{
	long i, j, k;
	memset(cnt, 0, sizeof(int) * (size_t)d->K * d->L * d->Amax);
	for (i = 0; i < d->N; i++)
		for (j = 0; j < d->L; j++)
			if (d->valid[i * d->L + j])
				for (k = 0; k < d->P; k++) {
					int zz = z[(i * d->L + j) * d->P + k];
					int a = geno[(i * d->L + j) * d->P + k];
					if (zz >= 0 && zz < d->K && a >= 0 && a < d->allelenum[j])
						cnt[((long)zz * d->L + j) * d->Amax + a]++;
				}
}

static inline void dump_vec(FILE *f, const char *tag, const double *v, int n)
{
	int i;
	fprintf(f, "%s", tag);
	for (i = 0; i < n; i++) fprintf(f, " %a", v[i]);
	fprintf(f, "\n");
}
#endif
