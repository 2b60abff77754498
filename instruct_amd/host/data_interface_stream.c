/*
 * data_interface_stream.c -- drop-in replacement for the reference's data_interface.c:
 * same exported symbols (data_interface.h:58-62: read_data, sort, exists), same SEQDATA contents, same messages on
 * stdout, but the file is STREAMED: the reference first copies every token into a 100-byte cell
 * (data_interface.c:18,109-115: ~100 B per allele copy, 10 GB at N=10000 L=5000, 80 GB at BASELINE config 5) and
 * then codes it; here each token is coded the moment it is read (4 B per allele copy, the size of SEQDATA.seqdata
 * itself) against a small per-locus list of the allele strings seen so far.
 *
 * Coding rules kept (cited per function): allele strings are numbered per locus in order of first appearance
 * scanning individuals, then copies (transform_data :489-569, transform_data2 :571-669); diploid loci with fewer than
 * two allele types are dropped and the later ones move up (:524-552); ploidy 4 keeps every locus and stores the sorted
 * distinct codes of each (individual, locus) plus their number (:617-640); a token equal to the missing-data string
 * marks the copy missing (-9).  The -L / -N corrections and their messages (cnt_loci :356-388, cnt_lines :427-457
 * and the *2 variants) are reproduced, as is the dump of the coded data on stdout.
 *
 * Layouts read: `-af 0` (ploidy lines per individual, read_data_from_file :133-245) and `-af 1` / ploidy 4 (one line
 * per individual, ploidy adjacent tokens per locus, read_data_from_file2 :247-350), with the optional label,
 * population, extra columns and the marker-name line.
 *
 * Not reproduced: fgets' 1e6-character line limit (MAXLEN, :17) -- lines may be longer here -- and the reference's
 * behaviour on malformed files beyond its own error messages (e.g. trailing blank lines: the reference miscounts the
 * individuals; this reader stops with the reference's token-count error).
 *
 * Plain C; links against the host program's nrutil objects like the reference's data_interface.o does.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "instruct_types.h"

#define ELMLEN 100   /* data_interface.c:18 */
#define INCRE_POP 5  /* data_interface.c:19 */

/* the host program's allocators and error exit (nrutil.h:49-63, nrutil.c:9-16) */
void nrerror(char error_text[]);
char *cvector(long nl, long nh);
char **cmatrix(long nrl, long nrh, long ncl, long nch);
char ***c3tensor(long nrl, long nrh, long ncl, long nch, long ndl, long ndh);
int *ivector(long nl, long nh);
int **imatrix(long nrl, long nrh, long ncl, long nch);
int ***i3tensor(long nrl, long nrh, long ncl, long nch, long ndl, long ndh);

void sort(int leng, int *vec) /* data_interface.c:709-720 */
{
	int i, j, t;
	for (i = 0; i < leng - 1; i++)
		for (j = i + 1; j < leng; j++)
			if (vec[i] > vec[j]) { t = vec[i]; vec[i] = vec[j]; vec[j] = t; }
}

int exists(int value, int *vec, int leng) /* data_interface.c:865-880 */
{
	int i, flag = 0;
	for (i = 0; i < leng; i++)
		if (value == vec[i]) flag = 1;
	return flag;
}

static int word_cnt(const char *s) /* data_interface.c:765-783 */
{
	int cnt = 0;
	while (*s != '\0') {
		while (isspace((unsigned char)*s)) ++s;
		if (*s != '\0') {
			++cnt;
			while (!isspace((unsigned char)*s) && *s != '\0') ++s;
		}
	}
	return cnt;
}

/* line reader with fgets' end-of-file behaviour as the reference's loops see it: a failed read leaves the previous
 * line in place (data_interface.c:236, 340) */
typedef struct {
	FILE *f;
	char *line;
	size_t cap;
	int eof; /* feof() after the last read */
} reader;

static void next_line(reader *r)
{
	ssize_t n = getline(&r->line, &r->cap, r->f);
	(void)n;
	r->eof = feof(r->f);
}

/* per-locus list of allele strings in order of first appearance */
typedef struct {
	char **s;
	int n, cap;
} strlist;

static int list_find(const strlist *l, const char *tok, size_t len)
{
	int i;
	for (i = 0; i < l->n; i++)
		if (strncmp(l->s[i], tok, len) == 0 && l->s[i][len] == '\0') return i;
	return -1;
}
static int list_add(strlist *l, const char *tok, size_t len)
{
	if (l->n == l->cap) {
		l->cap = l->cap ? 2 * l->cap : 4;
		l->s = (char **)realloc(l->s, sizeof(char *) * (size_t)l->cap);
		if (!l->s) nrerror("Memory allocation for the allele lists in read_data()!\n");
	}
	l->s[l->n] = (char *)malloc(len + 1);
	if (!l->s[l->n]) nrerror("Memory allocation for the allele lists in read_data()!\n");
	memcpy(l->s[l->n], tok, len);
	l->s[l->n][len] = '\0';
	return l->n++;
}

/* copy of token [tok, tok+len) into a fixed ELMLEN cell, as word_split + strcpy do (tokens of 100+ characters
 * overflow the reference's cells; here they are refused) */
static void cell_copy(char *dst, const char *tok, size_t len)
{
	if (len >= ELMLEN) nrerror("A token of the input file is longer than 99 characters!\n");
	memcpy(dst, tok, len);
	dst[len] = '\0';
}

/* next token of *p (NULL at the end of the line) */
static const char *next_tok(const char **p, size_t *len)
{
	const char *s = *p, *b;
	while (isspace((unsigned char)*s)) ++s;
	if (*s == '\0') { *p = s; return NULL; }
	b = s;
	while (!isspace((unsigned char)*s) && *s != '\0') ++s;
	*len = (size_t)(s - b);
	*p = s;
	return b;
}

static int isnew_cell(const char *str, int len, char **array) /* data_interface.c:744-763 */
{
	int i;
	for (i = 0; i < len; i++)
		if (strcmp(str, array[i]) == 0) return i;
	return -1;
}

/* label, population and extra columns of the individual's (first) line; returns the rest of the line */
static const char *read_individual_columns(SEQDATA *d, const char *p, int count, int *pop_cnt, int *max_pop, int first, int cnt_token)
{
	size_t len;
	const char *tok;
	char cell[ELMLEN];
	int j, indx;
	(void)cnt_token;
	if (d->label == 1) {
		tok = next_tok(&p, &len);
		cell_copy(cell, tok, len);
		if (first) strcpy(d->indvname[count], cell);
		else if (strcmp(d->indvname[count], cell) != 0) nrerror("Some individuals have different number of haplotypes!\n");
	}
	if (d->popdata == 1) {
		tok = next_tok(&p, &len);
		if (first) {
			cell_copy(cell, tok, len);
			indx = isnew_cell(cell, *pop_cnt, d->poptype);
			if (indx == -1) {
				if (*max_pop <= *pop_cnt) {
					d->poptype = (char **)realloc(d->poptype, (size_t)(*max_pop += INCRE_POP) * sizeof(char *));
					if (d->poptype == NULL) nrerror("Memory reallocation for variable \'data->poptype\' in function read_data_from_file()!\n");
					for (j = *max_pop - INCRE_POP; j < *max_pop; j++)
						if ((d->poptype[j] = (char *)malloc(ELMLEN)) == NULL)
							nrerror("Memory reallocation for variable \'data->poptype[i]\' in function read_data_from_file()!\n");
				}
				(*pop_cnt)++;
				strcpy(d->poptype[*pop_cnt - 1], cell);
				d->popindx[count] = *pop_cnt - 1;
			} else {
				d->popindx[count] = indx;
			}
		}
	}
	for (j = 0; j < d->n_extra_col; j++) {
		tok = next_tok(&p, &len);
		if (first) cell_copy(d->extra_col[count][j], tok, len);
	}
	return p;
}

SEQDATA read_data(char *infilename, int ploid, int totalsize, int popnum, int nloci, char *missingdata, int label, int popdata,
		  double siglevel, int back_refl, int type_freq, int nstep_check_empty_cluster, int prior_flag, int mode, int n_extra_col,
		  int markername_flag, double alpha_dpm, int print_iter, int print_freq, int inf_K, int distr_fmt, int autopoly, int datafmt,
		  double max_mem) /* data_interface.c:36-84 */
{
	SEQDATA data;
	reader r;
	strlist *lists;
	int *code; /* [N][L][P] index into the locus' list, -9 = missing */
	int fmt2, cnt_token, lead, cnt_line = 0, i, j, k, m, count = 0, pop_cnt = 0, max_pop = 0, L, P, N;
	size_t misslen;
	const int missing_num = -9;

	memset(&data, 0, sizeof(data));
	data.label = label; data.popdata = popdata; data.n_extra_col = n_extra_col; data.markername_flag = markername_flag;
	data.ploid = ploid; data.siglevel = siglevel; data.back_refl = back_refl; data.type_freq = type_freq;
	data.nstep_check_empty_cluster = nstep_check_empty_cluster; data.prior_flag = prior_flag; data.mode = mode;
	data.popnum = popnum; data.missingdata = missingdata; data.totalsize = totalsize; data.locinum = nloci;
	data.alpha_dpm = alpha_dpm; data.print_iter = print_iter; data.print_freq = print_freq; data.inf_K = inf_K;
	data.distr_fmt = distr_fmt; data.autopoly = autopoly; data.datafmt = datafmt; data.max_mem = max_mem;
	if (ploid != 2 && ploid != 4) { /* the reference reads nothing then (data_interface.c:72-82) */
		nrerror("read_data: ploidy must be 2 or 4");
	}
	if (ploid == 2 && datafmt != 0 && datafmt != 1) nrerror("read_data: -af must be 0 or 1");
	fmt2 = (ploid == 4) || (datafmt == 1);
	P = ploid;
	lead = label + popdata + n_extra_col;
	misslen = strlen(missingdata);

	memset(&r, 0, sizeof(r));
	if ((r.f = fopen(infilename, "r")) == NULL) nrerror("Cannot open input file!\n");

	/* ---- cnt_loci / cnt_loci2 (:356-425): the first line decides the number of loci ---- */
	next_line(&r);
	if (r.line == NULL) nrerror("Cannot open input file!\n");
	cnt_token = word_cnt(r.line);
	if (markername_flag == 1) {
		const char *p = r.line, *tok;
		size_t len;
		data.locinum = cnt_token;
		data.marker_names = cmatrix(0, cnt_token - 1, 0, ELMLEN - 1);
		for (i = 0; (tok = next_tok(&p, &len)) != NULL; i++) cell_copy(data.marker_names[i], tok, len);
		fprintf(stdout, "The number of loci is %d now!\n", data.locinum);
	} else {
		const int found = fmt2 ? (cnt_token - lead) / P : cnt_token - lead;
		if (found != data.locinum) {
			data.locinum = found;
			fprintf(stdout, "The Input Number of Loci is wrong!\nThe number of loci is %d now!\n", data.locinum);
		}
	}
	L = data.locinum;

	/* ---- cnt_lines / cnt_lines2 (:427-487): the number of lines decides the number of individuals ---- */
	rewind(r.f);
	for (;;) {
		ssize_t n = getline(&r.line, &r.cap, r.f);
		if (n < 0) break;
		/* a last line without newline that is short and starts with a non-blank is not counted (:439, :472) */
		if (feof(r.f) && isspace((unsigned char)r.line[0]) == 0 && strlen(r.line) <= (size_t)L) break;
		cnt_line++;
	}
	if (!fmt2) {
		if ((cnt_line - markername_flag) % P != 0) nrerror("Some individuals do not have two copies of haplotype!\n");
		if (data.totalsize != (cnt_line - markername_flag) / P) {
			data.totalsize = (cnt_line - markername_flag) / P;
			fprintf(stdout, "The input population size is incorrect!\nThe population size is %d\n", data.totalsize);
		}
	} else if (data.totalsize != cnt_line - markername_flag) {
		data.totalsize = cnt_line - markername_flag;
		fprintf(stdout, "The input population size is incorrect!\ncnt_line is %d.\nThe population size is %d\n", cnt_line, data.totalsize);
	}
	N = data.totalsize;

	/* ---- read_data_from_file / read_data_from_file2: per-individual columns, tokens coded on the fly ---- */
	cnt_token = fmt2 ? lead + L * P : lead + L;
	if (popdata == 1) {
		data.popindx = ivector(0, N - 1);
		if ((data.poptype = (char **)malloc(INCRE_POP * sizeof(char *))) == NULL)
			nrerror("Memory allocation for variable \'data->poptype\' in function read_data_from_file()!\n");
		for (i = 0; i < INCRE_POP; i++)
			if ((data.poptype[i] = (char *)malloc(ELMLEN)) == NULL)
				nrerror("Memory allocation for variable \'data->poptype[i]\' in function read_data_from_file()!\n");
		max_pop = INCRE_POP;
	}
	if (label == 1) data.indvname = cmatrix(0, N - 1, 0, ELMLEN - 1);
	if (n_extra_col > 0) data.extra_col = c3tensor(0, N - 1, 0, n_extra_col - 1, 0, ELMLEN - 1);
	lists = (strlist *)calloc((size_t)(L > 0 ? L : 1), sizeof(strlist));
	code = (int *)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1) * (size_t)(L > 0 ? L : 1) * (size_t)P);
	if (!lists || !code) nrerror("Memory allocation for variable \'allele\' in function read_seqs()!\n");

	rewind(r.f);
	next_line(&r);
	if (markername_flag == 1) next_line(&r);
	while (!r.eof && count < N) {
		const int nlines = fmt2 ? 1 : P;
		for (i = 0; i < nlines; i++) {
			const char *p = r.line, *tok;
			size_t len;
			if (word_cnt(r.line) != cnt_token) {
				if (fmt2) nrerror("The number of tokens in a line is different from the number given !\n");
				nrerror("The number of tokens in one line does not match the parameters input from the commandline");
			}
			p = read_individual_columns(&data, p, count, &pop_cnt, &max_pop, i == 0, cnt_token);
			for (j = 0; j < L; j++) {
				const int ncopy = fmt2 ? P : 1;
				for (m = 0; m < ncopy; m++) {
					const int kk = fmt2 ? m : i;
					int c;
					tok = next_tok(&p, &len);
					if (len >= ELMLEN) nrerror("A token of the input file is longer than 99 characters!\n");
					if (len == misslen && strncmp(tok, missingdata, len) == 0) c = missing_num;
					else if ((c = list_find(&lists[j], tok, len)) < 0) c = list_add(&lists[j], tok, len);
					code[((size_t)count * L + j) * P + kk] = c;
				}
			}
			next_line(&r);
			if (fmt2) {
				if (strlen(r.line) < (size_t)cnt_token) break; /* :341-345 */
				if (word_cnt(r.line) != cnt_token) nrerror("The lines of input files do not have the same number of tokens!\n");
			} else if (word_cnt(r.line) != cnt_token) {
				nrerror("The lines of input files do not have the same number of tokens!\n");
			}
		}
		count++;
		if (fmt2 && strlen(r.line) < (size_t)cnt_token) break;
	}
	if (count != N) nrerror("The lines of input files do not have the same number of tokens!\n");
	data.pop_count = pop_cnt;
	fclose(r.f);
	free(r.line);

	/* ---- transform_data (:489-569) / transform_data2 (:571-669) ---- */
	data.seqdata = i3tensor(0, N - 1, 0, L - 1, 0, P - 1);
	data.alleletype = (char ***)malloc((size_t)(L > 0 ? L : 1) * sizeof(char **));
	if (data.alleletype == NULL) nrerror("Memory allocation for variable \'(*data)->alleletype\' in function transform_data()!\n");
	data.allelenum = ivector(0, L - 1);
	if (P == 2) {
		int allele_cnt = 0;
		for (j = 0; j < L; j++) {
			const int cnt = lists[j].n;
			if (cnt >= 2) {
				data.allelenum[allele_cnt] = cnt;
				data.alleletype[allele_cnt] = cmatrix(0, cnt - 1, 0, ELMLEN - 1);
				for (k = 0; k < cnt; k++) strcpy(data.alleletype[allele_cnt][k], lists[j].s[k]);
				for (i = 0; i < N; i++)
					for (k = 0; k < P; k++) data.seqdata[i][allele_cnt][k] = code[((size_t)i * L + j) * P + k];
				allele_cnt++;
			} else {
				fprintf(stdout, "The locus %d is not polymorphic.\n", j + 1);
			}
		}
		data.missingnum = missing_num;
		data.locinum = allele_cnt;
		fprintf(stdout, "The number of polymorphic loci is %d now.\n", allele_cnt);
	} else {
		data.alleleid = imatrix(0, N - 1, 0, L - 1);
		for (j = 0; j < L; j++) {
			const int cnt = lists[j].n;
			data.allelenum[j] = cnt;
			data.alleletype[j] = cmatrix(0, cnt - 1, 0, ELMLEN - 1);
			for (k = 0; k < cnt; k++) strcpy(data.alleletype[j][k], lists[j].s[k]);
			for (i = 0; i < N; i++) {
				int flag = 0;
				for (k = 0; k < P; k++) data.seqdata[i][j][k] = -1;
				for (k = 0; k < P; k++) {
					const int c = code[((size_t)i * L + j) * P + k];
					if (c != missing_num && exists(c, data.seqdata[i][j], flag) == 0) data.seqdata[i][j][flag++] = c;
				}
				sort(flag, data.seqdata[i][j]);
				data.alleleid[i][j] = flag;
				if (flag == 0) data.seqdata[i][j][0] = missing_num;
			}
		}
		data.missingnum = missing_num;
		fprintf(stdout, "Print the number of alleles per individual per locus:\n");
		for (i = 0; i < N; i++) {
			for (j = 0; j < L; j++) fprintf(stdout, "%d ", data.alleleid[i][j]);
			fprintf(stdout, "\n");
		}
	}
	fprintf(stdout, "Print the transformed allele data:\n");
	for (i = 0; i < N; i++)
		for (k = 0; k < P; k++) {
			for (j = 0; j < data.locinum; j++) fprintf(stdout, "%d ", data.seqdata[i][j][k]);
			fprintf(stdout, "\n");
		}
	fprintf(stdout, "End the printing of the transformed allele data.\n");
	for (j = 0; j < L; j++) {
		for (k = 0; k < lists[j].n; k++) free(lists[j].s[k]);
		free(lists[j].s);
	}
	free(lists);
	free(code);

	/* ---- get_missing (:812-847) / get_missing_tetra (:722-741) ---- */
	data.missvec = ivector(0, N - 1);
	data.missindx = imatrix(0, N - 1, 0, data.locinum - 1);
	for (i = 0; i < N; i++) {
		data.missvec[i] = 0;
		for (j = 0; j < data.locinum; j++) {
			data.missindx[i][j] = 0;
			if (P == 2) {
				for (k = 0; k < P; k++)
					if (data.seqdata[i][j][k] == data.missingnum) data.missindx[i][j] = 1;
			} else if (data.alleleid[i][j] == 0) {
				data.missindx[i][j] = 1;
			}
			data.missvec[i] += data.missindx[i][j];
		}
	}
	/* find_max (:849-863) */
	data.allelenum_max = data.allelenum[0];
	for (i = 1; i < data.locinum; i++)
		if (data.allelenum_max < data.allelenum[i]) data.allelenum_max = data.allelenum[i];
	return data;
}
