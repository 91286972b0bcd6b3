/*
 * mc_reader.c -- STRUCTURE-format reader (reference read_file.c:38-300, 443-663), default allele-code mode.
 *
 * Same observable result as the reference: L from the header line (minus 2 with -R), optional "-1" line skipped,
 * interleaved layout detected from the first two names, `ploidy` consecutive lines per individual otherwise,
 * locales numbered in order of first appearance, per-locus ascending allele lists, the phantom trailing allele slot
 * for loci that have missing data (uniquealleles counts it, L_alleles does not; read_file.c:527-533 vs 581-585),
 * and the genotype in allele-index form.  Not the same cost: one pass over the file in memory and one O(n log n)
 * sort per locus instead of two O(n^2) bubble sorts per locus (read_file.c:518,577).
 */
#include "mc_cli.h"

#include <stdlib.h>
#include <string.h>

static int fail(const char *fn, int line, const char *msg, const char *arg)
{
	fprintf(stderr, "ERROR [mc_reader.c::%s(%d)]: ", fn, line);
	fprintf(stderr, msg, arg ? arg : "");
	fprintf(stderr, "\n");
	return 1;
}
#define FAIL(msg, arg) fail(__func__, __LINE__, msg, arg)

static int is_space(char c) { return c == ' ' || c == '\t' || c == '\r'; }

/* next token on the line ending at `end`; returns NULL when the line is exhausted */
static char *next_token(char **cur, char *end, size_t *len)
{
	char *p = *cur;
	while (p < end && is_space(*p)) p++;
	if (p >= end) { *cur = p; return NULL; }
	char *s = p;
	while (p < end && !is_space(*p)) p++;
	*len = (size_t)(p - s);
	*cur = p;
	return s;
}

static int count_tokens(char *s, char *end)
{
	int n = 0;
	size_t len;
	while (next_token(&s, end, &len)) n++;
	return n;
}

static char *dup_token(const char *s, size_t len)
{
	char *r = malloc(len + 1);
	if (r) { memcpy(r, s, len); r[len] = 0; }
	return r;
}

static int cmp_int(const void *a, const void *b)
{
	int x = *(const int *)a, y = *(const int *)b;
	return (x > y) - (x < y);
}

int mc_read_structure(const mc_cli_options *opt, mc_cli_data *dat)
{
	FILE *f = fopen(opt->filename, "rb");
	if (!f) return FAIL("could not open file '%s'", opt->filename);
	fseek(f, 0, SEEK_END);
	long fsz = ftell(f);
	fseek(f, 0, SEEK_SET);
	char *buf = malloc((size_t)fsz + 2);
	if (!buf) { fclose(f); return FAIL("out of memory reading '%s'", opt->filename); }
	if (fread(buf, 1, (size_t)fsz, f) != (size_t)fsz) { fclose(f); free(buf); return FAIL("short read on '%s'", opt->filename); }
	fclose(f);
	buf[fsz] = '\n';
	buf[fsz + 1] = 0;

	/* split into non-empty lines */
	size_t nlines = 0, cap = 1024;
	char **ls = malloc(cap * sizeof *ls), **le = malloc(cap * sizeof *le);
	for (char *p = buf, *eof = buf + fsz + 1; p < eof;) {
		char *q = memchr(p, '\n', (size_t)(eof - p));
		if (!q) q = eof;
		char *s = p;
		while (s < q && is_space(*s)) s++;
		if (s < q) {
			if (nlines == cap) { cap *= 2; ls = realloc(ls, cap * sizeof *ls); le = realloc(le, cap * sizeof *le); }
			ls[nlines] = p; le[nlines] = q; nlines++;
		}
		p = q + 1;
	}
	int rc = 1;
	memset(dat, 0, sizeof *dat);
	dat->ploidy = opt->ploidy;
	if (nlines < 3) { FAIL("file '%s' has no data lines", opt->filename); goto DONE; }

	int L = count_tokens(ls[0], le[0]);		/* read_file.c:56 */
	if (opt->R_format) L -= 2;
	size_t first = 1, len;
	{	/* optional inter-marker distance line (read_file.c:70-82) */
		char *c = ls[1];
		char *t = next_token(&c, le[1], &len);
		if (t && len == 2 && !strncmp(t, "-1", 2)) first = 2;
	}
	if (nlines < first + 2) { FAIL("file '%s' has fewer than two data lines", opt->filename); goto DONE; }
	{	/* interleaved iff the first two data lines carry different names (read_file.c:89-90) */
		char *c1 = ls[first], *c2 = ls[first + 1];
		size_t l1, l2;
		char *n1 = next_token(&c1, le[first], &l1), *n2 = next_token(&c2, le[first + 1], &l2);
		dat->interleaved = !(n1 && n2 && l1 == l2 && !strncmp(n1, n2, l1));
	}
	const int pl = dat->ploidy;
	int ncol = count_tokens(ls[first + 1], le[first + 1]) - 2;	/* allele columns of a data line (read_file.c:100) */
	if (dat->interleaved && ncol != L && ncol != pl * L) { FAIL("number of allele columns in '%s' is not a multiple of ploidy", opt->filename); goto DONE; }
	if (!dat->interleaved && ncol != L) { FAIL("number of locus names in '%s' does not match the alleles of the first individual (check -R)", opt->filename); goto DONE; }
	if (dat->interleaved && ncol == L) L /= pl;	/* header names every allele column (read_file.c:115-116) */
	/* read_file.c:119: I = (remaining lines) + 2 - skip_line_two.  With the "-1" line present the reference counts one
	 * data line too few: it drops the last individual of an interleaved file and rejects a non-interleaved one whose
	 * line count was right.  Kept, so that the same file gives the same fit. */
	size_t ndata = nlines - first - (first == 2 ? 1 : 0);
	if (!dat->interleaved && (ndata % (size_t)pl)) { FAIL("number of lines in '%s' is not a multiple of ploidy", opt->filename); goto DONE; }
	const int I = dat->interleaved ? (int)ndata : (int)(ndata / (size_t)pl);
	const int nhap = I * pl;
	if (L <= 0 || I <= 0) { FAIL("no loci or no individuals in '%s'", opt->filename); goto DONE; }
	dat->I = I; dat->L = L;
	dat->IL = malloc(sizeof(int) * (size_t)nhap * L);
	dat->names = calloc((size_t)I, sizeof *dat->names);
	dat->locale = calloc((size_t)I, sizeof *dat->locale);
	if (!dat->IL || !dat->names || !dat->locale) { FAIL("out of memory%s", NULL); goto DONE; }

	for (size_t ln = 0; ln < ndata; ln++) {
		char *c = ls[first + ln], *end = le[first + ln];
		const int i = dat->interleaved ? (int)ln : (int)(ln / (size_t)pl);
		const int h0 = dat->interleaved ? i * pl : (int)ln;
		char *name = next_token(&c, end, &len);
		size_t llen, nlen = len;
		char *loc = next_token(&c, end, &llen);
		if (!name || !loc) { FAIL("line without name/locale columns in '%s'", opt->filename); goto DONE; }
		if (dat->interleaved || !(ln % (size_t)pl)) {
			dat->names[i] = dup_token(name, nlen);
			int found = -1;				/* add_to_string_set: order of first appearance */
			for (int n = 0; n < dat->numpops; n++)
				if (strlen(dat->pops[n]) == llen && !strncmp(dat->pops[n], loc, llen)) { found = n; break; }
			if (found < 0) {
				dat->pops = realloc(dat->pops, sizeof *dat->pops * (size_t)(dat->numpops + 1));
				dat->pops[dat->numpops] = dup_token(loc, llen);
				found = dat->numpops++;
			}
			dat->locale[i] = found;
		}
		for (int l = 0; l < L; l++)
			for (int j = 0; j < (dat->interleaved ? pl : 1); j++) {
				char *t = next_token(&c, end, &len), *endp;
				if (!t) { FAIL("failed to read a locus in file '%s'.  Check option -R.", opt->filename); goto DONE; }
				long v = strtol(t, &endp, 10);
				if (endp == t) { FAIL("non-integer allele in file '%s'", opt->filename); goto DONE; }
				if ((int)v == opt->missing_value) v = MC_MISSING;	/* change_missing_value, read_file.c:266-268 */
				dat->IL[(size_t)(h0 + j) * L + l] = (int)v;
			}
	}
	dat->i_p = calloc((size_t)dat->numpops, sizeof *dat->i_p);
	for (int i = 0; i < I; i++) dat->i_p[dat->locale[i]]++;

	/* summarize_alleles (read_file.c:443-600) */
	dat->uniquealleles = calloc((size_t)L, sizeof *dat->uniquealleles);
	dat->L_alleles = calloc((size_t)L, sizeof *dat->L_alleles);
	dat->toff = calloc((size_t)L + 1, sizeof *dat->toff);
	dat->geno = malloc((size_t)I * L * pl);
	int *col = malloc(sizeof(int) * (size_t)nhap);
	if (!dat->uniquealleles || !dat->L_alleles || !dat->toff || !dat->geno || !col) { FAIL("out of memory%s", NULL); goto DONE; }
	for (int l = 0; l < L; l++) {
		for (int h = 0; h < nhap; h++) col[h] = dat->IL[(size_t)h * L + l];
		qsort(col, (size_t)nhap, sizeof(int), cmp_int);
		int nreal = 0, has_missing = 0;
		for (int h = 0; h < nhap; h++) {
			if (col[h] == MC_MISSING) { has_missing = 1; continue; }
			if (!nreal || col[h] != col[nreal - 1]) col[nreal++] = col[h];	/* compact uniques in place */
		}
		if (!nreal) { free(col); FAIL("a locus of '%s' has no observed allele", opt->filename); goto DONE; }
		if (nreal > 254) { free(col); FAIL("a locus of '%s' has more than 254 alleles", opt->filename); goto DONE; }
		if (has_missing) dat->missing_data = 1;
		dat->uniquealleles[l] = nreal + (has_missing ? 1 : 0);
		dat->L_alleles[l] = malloc(sizeof(int) * (size_t)nreal);
		memcpy(dat->L_alleles[l], col, sizeof(int) * (size_t)nreal);
		if (dat->uniquealleles[l] > dat->M) dat->M = dat->uniquealleles[l];
		dat->toff[l + 1] = dat->toff[l] + dat->uniquealleles[l];
		for (int i = 0; i < I; i++)
			for (int a = 0; a < pl; a++) {
				const int v = dat->IL[(size_t)(i * pl + a) * L + l];
				uint8_t idx = MCHIP_MISSING;
				if (v != MC_MISSING) {
					int lo = 0, hi = nreal - 1;
					while (lo < hi) {
						int mid = (lo + hi) / 2;
						if (dat->L_alleles[l][mid] < v) lo = mid + 1; else hi = mid;
					}
					idx = (uint8_t)lo;
				}
				dat->geno[((size_t)i * L + l) * pl + a] = idx;
			}
	}
	free(col);
	dat->T = dat->toff[L];
	rc = 0;
DONE:
	free(ls); free(le); free(buf);
	if (rc) mc_free_data(dat);
	return rc;
}

void mc_free_data(mc_cli_data *dat)
{
	if (!dat) return;
	free(dat->IL); free(dat->uniquealleles); free(dat->geno); free(dat->toff); free(dat->locale); free(dat->i_p);
	if (dat->L_alleles) for (int l = 0; l < dat->L; l++) free(dat->L_alleles[l]);
	free(dat->L_alleles);
	if (dat->names) for (int i = 0; i < dat->I; i++) free(dat->names[i]);
	free(dat->names);
	if (dat->pops) for (int n = 0; n < dat->numpops; n++) free(dat->pops[n]);
	free(dat->pops);
	memset(dat, 0, sizeof *dat);
}
