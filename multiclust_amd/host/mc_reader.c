/*
 * mc_reader.c -- STRUCTURE-format reader (reference read_file.c:38-300, 443-663), default allele-code mode.
 *
 * Same observable result as the reference: L from the header line (minus 2 with -R), optional "-1" line skipped,
 * interleaved layout detected from the first two names, `ploidy` consecutive lines per individual otherwise,
 * locales numbered in order of first appearance, per-locus ascending allele lists, the phantom trailing allele slot
 * for loci that have missing data (uniquealleles counts it, L_alleles does not; read_file.c:527-533 vs 581-585),
 * and the genotype in allele-index form.  Not the same cost: the file is parsed in memory by several threads (lines are
 * independent), and the per-locus allele lists are built by threads over blocks of loci with one small sorted insert
 * list per locus instead of two O(n^2) bubble sorts over all haplotypes (read_file.c:518,577).
 */
#include "mc_cli.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* prints and returns the exit status the reference's reader returns for this kind of failure (message.h:21-41: out of memory 3,
 * file cannot be opened 5, anything wrong with its contents 7) */
static int fail(const char *fn, int line, const char *msg, const char *arg)
{
	fprintf(stderr, "ERROR [mc_reader.c::%s(%d)]: ", fn, line);
	fprintf(stderr, msg, arg ? arg : "");
	fprintf(stderr, "\n");
	if (!strncmp(msg, "out of memory", 13)) return MC_EXIT_MEMORY_ALLOCATION;
	if (!strncmp(msg, "could not open", 14)) return MC_EXIT_FILE_OPEN_ERROR;
	return MC_EXIT_FILE_FORMAT_ERROR;
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

/* ---- worker threads ---- */
enum { RD_OK = 0, RD_SHORT_LINE, RD_NOT_INT, RD_TOO_MANY, RD_NOMEM };

typedef struct parse_job {
	char **cur, **end;		/* per data line: first allele token / end of line */
	size_t ln0, ln1;		/* data lines [ln0, ln1) */
	int L, pl, interleaved, missing_value;
	int *IL;
	int err;
} parse_job;

/* the numeric part of data lines ln0..ln1-1 into IL (strtol semantics for well-formed tokens: optional sign, digits) */
static void *parse_main(void *arg)
{
	parse_job *j = arg;
	const int per_line = j->interleaved ? j->pl : 1;
	for (size_t ln = j->ln0; ln < j->ln1; ln++) {
		const char *c = j->cur[ln], *end = j->end[ln];
		const size_t h0 = j->interleaved ? ln * (size_t)j->pl : ln;
		for (int l = 0; l < j->L; l++)
			for (int x = 0; x < per_line; x++) {
				while (c < end && is_space(*c)) c++;
				if (c >= end) { j->err = RD_SHORT_LINE; return NULL; }
				int neg = 0;
				long v = 0;
				if (*c == '-' || *c == '+') { neg = *c == '-'; c++; }
				if (c >= end || *c < '0' || *c > '9') { j->err = RD_NOT_INT; return NULL; }
				while (c < end && *c >= '0' && *c <= '9') v = v * 10 + (*c++ - '0');
				while (c < end && !is_space(*c)) c++;		/* trailing characters of the token, as strtol leaves them */
				if (neg) v = -v;
				if ((int)v == j->missing_value) v = MC_MISSING;	/* change_missing_value, read_file.c:266-268 */
				j->IL[(h0 + (size_t)x) * (size_t)j->L + (size_t)l] = (int)v;
			}
	}
	return NULL;
}

typedef struct locus_job {
	const int *IL;
	int I, L, pl, l0, l1;
	int32_t *uniquealleles;
	int **L_alleles;
	uint8_t *geno;
	int missing_data, M, err;
} locus_job;

#define LOCUS_BLOCK 1024
/* summarize_alleles (read_file.c:443-600) for loci l0..l1-1: ascending list of observed alleles, phantom slot when the
 * locus has missing data, genotype bytes.  Loci are taken LOCUS_BLOCK at a time so that IL rows are read in 4 KB runs. */
static void *locus_main(void *arg)
{
	locus_job *j = arg;
	const int nhap = j->I * j->pl, L = j->L;
	/* per locus of the block: its sorted allele list (only the first few entries of a row are ever touched for the usual
	 * handful of alleles), its length, whether a copy is missing, the last allele seen */
	int (*uniq)[256] = malloc(sizeof(int[256]) * LOCUS_BLOCK);
	int *nu = malloc(sizeof(int) * 3 * LOCUS_BLOCK), *miss = nu + LOCUS_BLOCK, *last = nu + 2 * LOCUS_BLOCK;
	if (!uniq || !nu) { free(uniq); free(nu); j->err = RD_NOMEM; return NULL; }
	for (int b0 = j->l0; b0 < j->l1; b0 += LOCUS_BLOCK) {
		const int nb = j->l1 - b0 < LOCUS_BLOCK ? j->l1 - b0 : LOCUS_BLOCK;
		for (int x = 0; x < nb; x++) { nu[x] = 0; miss[x] = 0; last[x] = MC_MISSING; }
		for (int h = 0; h < nhap; h++) {
			const int *row = j->IL + (size_t)h * L + b0;
			for (int x = 0; x < nb; x++) {
				const int v = row[x];
				if (v == MC_MISSING) { miss[x] = 1; continue; }
				if (v == last[x] && nu[x]) continue;	/* same as the previous observed allele: already known */
				last[x] = v;
				int lo = 0, hi = nu[x];			/* sorted insert */
				while (lo < hi) {
					const int mid = (lo + hi) / 2;
					if (uniq[x][mid] < v) lo = mid + 1; else hi = mid;
				}
				if (lo < nu[x] && uniq[x][lo] == v) continue;
				if (nu[x] >= 255) { j->err = RD_TOO_MANY; free(uniq); free(nu); return NULL; }
				memmove(&uniq[x][lo + 1], &uniq[x][lo], sizeof(int) * (size_t)(nu[x] - lo));
				uniq[x][lo] = v;
				nu[x]++;
			}
		}
		for (int x = 0; x < nb; x++) {
			const int l = b0 + x;
			if (nu[x] > 254) { j->err = RD_TOO_MANY; free(uniq); free(nu); return NULL; }
			/* every haplotype missing at this locus: the reference leaves uniquealleles = 0 and moves on before it would set
			 * missing_data (read_file.c:524-527): a locus without any allele column (golden allmiss_*) */
			if (miss[x] && nu[x]) j->missing_data = 1;
			j->uniquealleles[l] = nu[x] ? nu[x] + (miss[x] ? 1 : 0) : 0;
			if (!(j->L_alleles[l] = malloc(sizeof(int) * (size_t)(nu[x] ? nu[x] : 1)))) { j->err = RD_NOMEM; free(uniq); free(nu); return NULL; }
			memcpy(j->L_alleles[l], uniq[x], sizeof(int) * (size_t)nu[x]);
			if (j->uniquealleles[l] > j->M) j->M = j->uniquealleles[l];
		}
		for (int i = 0; i < j->I; i++)
			for (int a = 0; a < j->pl; a++) {
				const int *row = j->IL + (size_t)(i * j->pl + a) * L + b0;
				for (int x = 0; x < nb; x++) {
					const int v = row[x];
					uint8_t idx = MCHIP_MISSING;
					if (v != MC_MISSING) {
						int lo = 0, hi = nu[x] - 1;
						while (lo < hi) {
							const int mid = (lo + hi) / 2;
							if (uniq[x][mid] < v) lo = mid + 1; else hi = mid;
						}
						idx = (uint8_t)lo;
					}
					j->geno[((size_t)i * L + (size_t)(b0 + x)) * (size_t)j->pl + (size_t)a] = idx;
				}
			}
	}
	free(uniq);
	free(nu);
	return NULL;
}

static double now_s(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
#define PHASE(name) do { mchip_progress_note("mc_read_structure: after " name); if (timing) { double t_ = now_s(); fprintf(stderr, "INFO [mc_reader.c]: %-18s %.3f s\n", name, t_ - t_phase); t_phase = t_; } } while (0)

static int n_threads(size_t work)
{
	long cpus = sysconf(_SC_NPROCESSORS_ONLN);
	int nt = cpus > 16 ? 16 : (cpus < 1 ? 1 : (int)cpus);
	if (getenv("MC_READER_THREADS") && atoi(getenv("MC_READER_THREADS")) > 0) nt = atoi(getenv("MC_READER_THREADS"));
	if (nt > 64) nt = 64;
	if (work < ((size_t)1 << 20)) nt = 1;	/* small files: not worth a thread */
	return nt;
}

int mc_read_structure(const mc_cli_options *opt, mc_cli_data *dat)
{
	const int timing = getenv("MC_READER_TIMING") != NULL;
	double t_phase = now_s();
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
	if (!ls || !le) { free(ls); free(le); free(buf); return FAIL("out of memory reading '%s'", opt->filename); }
	for (char *p = buf, *eof = buf + fsz + 1; p < eof;) {
		char *q = memchr(p, '\n', (size_t)(eof - p));
		if (!q) q = eof;
		char *s = p;
		while (s < q && is_space(*s)) s++;
		if (s < q) {
			if (nlines == cap) {
				char **ls2 = realloc(ls, 2 * cap * sizeof *ls), **le2 = ls2 ? realloc(le, 2 * cap * sizeof *le) : NULL;
				if (ls2) ls = ls2;
				if (le2) le = le2;
				if (!ls2 || !le2) { free(ls); free(le); free(buf); return FAIL("out of memory reading '%s'", opt->filename); }
				cap *= 2;
			}
			ls[nlines] = p; le[nlines] = q; nlines++;
		}
		p = q + 1;
	}
	int rc = MC_EXIT_FILE_FORMAT_ERROR;
	PHASE("file into memory");
	memset(dat, 0, sizeof *dat);
	dat->ploidy = opt->ploidy;
	if (nlines < 3) { rc = FAIL("file '%s' has no data lines", opt->filename); goto DONE; }

	int L = count_tokens(ls[0], le[0]);		/* read_file.c:56 */
	if (opt->R_format) L -= 2;
	size_t first = 1, len;
	{	/* optional inter-marker distance line (read_file.c:70-82) */
		char *c = ls[1];
		char *t = next_token(&c, le[1], &len);
		if (t && len == 2 && !strncmp(t, "-1", 2)) first = 2;
	}
	if (nlines < first + 2) { rc = FAIL("file '%s' has fewer than two data lines", opt->filename); goto DONE; }
	{	/* interleaved iff the first two data lines carry different names (read_file.c:89-90) */
		char *c1 = ls[first], *c2 = ls[first + 1];
		size_t l1, l2;
		char *n1 = next_token(&c1, le[first], &l1), *n2 = next_token(&c2, le[first + 1], &l2);
		dat->interleaved = !(n1 && n2 && l1 == l2 && !strncmp(n1, n2, l1));
	}
	const int pl = dat->ploidy;
	int ncol = count_tokens(ls[first + 1], le[first + 1]) - 2;	/* allele columns of a data line (read_file.c:100) */
	if (dat->interleaved && ncol != L && ncol != pl * L) { rc = FAIL("number of allele columns in '%s' is not a multiple of ploidy", opt->filename); goto DONE; }
	if (!dat->interleaved && ncol != L) { rc = FAIL("number of locus names in '%s' does not match the alleles of the first individual (check -R)", opt->filename); goto DONE; }
	if (dat->interleaved && ncol == L) L /= pl;	/* header names every allele column (read_file.c:115-116) */
	/* read_file.c:119: I = (remaining lines) + 2 - skip_line_two.  With the "-1" line present the reference counts one
	 * data line too few: it drops the last individual of an interleaved file and rejects a non-interleaved one whose
	 * line count was right.  Kept, so that the same file gives the same fit. */
	size_t ndata = nlines - first - (first == 2 ? 1 : 0);
	if (!dat->interleaved && (ndata % (size_t)pl)) { rc = FAIL("number of lines in '%s' is not a multiple of ploidy", opt->filename); goto DONE; }
	const int I = dat->interleaved ? (int)ndata : (int)(ndata / (size_t)pl);
	const int nhap = I * pl;
	if (L <= 0 || I <= 0) { rc = FAIL("no loci or no individuals in '%s'", opt->filename); goto DONE; }
	dat->I = I; dat->L = L;
	dat->IL = malloc(sizeof(int) * (size_t)nhap * L);
	dat->names = calloc((size_t)I, sizeof *dat->names);
	dat->locale = calloc((size_t)I, sizeof *dat->locale);
	if (!dat->IL || !dat->names || !dat->locale) { rc = FAIL("out of memory%s", NULL); goto DONE; }

	for (size_t ln = 0; ln < ndata; ln++) {
		char *c = ls[first + ln], *end = le[first + ln];
		const int i = dat->interleaved ? (int)ln : (int)(ln / (size_t)pl);
		char *name = next_token(&c, end, &len);
		size_t llen, nlen = len;
		char *loc = next_token(&c, end, &llen);
		if (!name || !loc) { rc = FAIL("line without name/locale columns in '%s'", opt->filename); goto DONE; }
		if (dat->interleaved || !(ln % (size_t)pl)) {
			dat->names[i] = dup_token(name, nlen);
			int found = -1;				/* add_to_string_set: order of first appearance */
			for (int n = 0; n < dat->numpops; n++)
				if (strlen(dat->pops[n]) == llen && !strncmp(dat->pops[n], loc, llen)) { found = n; break; }
			if (found < 0) {
				char **pops2 = realloc(dat->pops, sizeof *dat->pops * (size_t)(dat->numpops + 1));
				if (!pops2) { rc = FAIL("out of memory%s", NULL); goto DONE; }
				dat->pops = pops2;
				dat->pops[dat->numpops] = dup_token(loc, llen);
				found = dat->numpops++;
			}
			dat->locale[i] = found;
		}
		ls[first + ln] = c;		/* from here on the line is numbers: parsed by the worker threads below */
	}
	PHASE("names + locales");
	{
		const int nt = n_threads((size_t)nhap * (size_t)L);
		parse_job jobs[64];
		pthread_t th[64];
		int joinable[64] = { 0 };	/* (a pthread_t has no "none" value) */
		const size_t per = (ndata + (size_t)nt - 1) / (size_t)nt;
		int perr = RD_OK;
		for (int t = 0; t < nt; t++) {
			const size_t lo = (size_t)t * per, hi = lo + per < ndata ? lo + per : ndata;
			jobs[t] = (parse_job){ ls + first, le + first, lo < ndata ? lo : ndata, hi, L, pl, dat->interleaved, opt->missing_value, dat->IL, RD_OK };
			if (nt == 1 || pthread_create(&th[t], NULL, parse_main, &jobs[t])) parse_main(&jobs[t]);
			else joinable[t] = 1;
		}
		for (int t = 0; t < nt; t++) {
			if (joinable[t]) pthread_join(th[t], NULL);
			if (!perr) perr = jobs[t].err;
		}
		if (perr == RD_SHORT_LINE) { rc = FAIL("failed to read a locus in file '%s'.  Check option -R.", opt->filename); goto DONE; }
		if (perr) { rc = FAIL("non-integer allele in file '%s'", opt->filename); goto DONE; }
	}
	PHASE("parse");
	dat->i_p = calloc((size_t)dat->numpops, sizeof *dat->i_p);
	for (int i = 0; i < I; i++) dat->i_p[dat->locale[i]]++;

	/* summarize_alleles (read_file.c:443-600) */
	dat->uniquealleles = calloc((size_t)L, sizeof *dat->uniquealleles);
	dat->L_alleles = calloc((size_t)L, sizeof *dat->L_alleles);
	dat->toff = calloc((size_t)L + 1, sizeof *dat->toff);
	dat->geno = malloc((size_t)I * L * pl);
	if (!dat->uniquealleles || !dat->L_alleles || !dat->toff || !dat->geno) { rc = FAIL("out of memory%s", NULL); goto DONE; }
	{
		const int nt = n_threads((size_t)nhap * (size_t)L);
		locus_job jobs[64];
		pthread_t th[64];
		int joinable[64] = { 0 };
		const int per = ((L + nt - 1) / nt + LOCUS_BLOCK - 1) / LOCUS_BLOCK * LOCUS_BLOCK;
		int lerr = RD_OK;
		for (int t = 0; t < nt; t++) {
			const int lo = t * per < L ? t * per : L, hi = lo + per < L ? lo + per : L;
			jobs[t] = (locus_job){ dat->IL, I, L, pl, lo, hi, dat->uniquealleles, dat->L_alleles, dat->geno, 0, 0, RD_OK };
			if (nt == 1 || pthread_create(&th[t], NULL, locus_main, &jobs[t])) locus_main(&jobs[t]);
			else joinable[t] = 1;
		}
		for (int t = 0; t < nt; t++) {
			if (joinable[t]) pthread_join(th[t], NULL);
			if (!lerr) lerr = jobs[t].err;
			if (jobs[t].missing_data) dat->missing_data = 1;
			if (jobs[t].M > dat->M) dat->M = jobs[t].M;
		}
		if (lerr == RD_TOO_MANY) { rc = FAIL("a locus of '%s' has more than 254 alleles", opt->filename); goto DONE; }
		if (lerr) { rc = FAIL("out of memory%s", NULL); goto DONE; }
	}
	for (int l = 0; l < L; l++) dat->toff[l + 1] = dat->toff[l] + dat->uniquealleles[l];
	dat->T = dat->toff[L];
	PHASE("alleles + genotype");
	/* the codes as read (4 bytes per allele copy: 8 GB at config-3 size) have served their purpose: nothing downstream
	 * reads them once the index form and the allele lists exist */
	free(dat->IL);
	dat->IL = NULL;
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
