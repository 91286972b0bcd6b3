/*
 * mc_cli.h -- the drop-in surface around the EM hot path: command line, STRUCTURE reader, result writers and the
 * model-selection driver of MULTICLUST, written from scratch as plain host C.  None of it touches the GPU
 * directly; all array arithmetic goes through mc_host.h -> include/multiclust_hip.h.
 * Each function cites the reference code whose observable behaviour (flags, file formats, stdout lines) it keeps.
 */
#ifndef MC_CLI_H
#define MC_CLI_H

#include <stdio.h>
#include "mc_host.h"

#define MC_MISSING (-9)		/* reference multiclust.h:140 */

/* exit status of the program: the reference returns its error enum from main() (message.h:21-41, multiclust.c:157-163), so a
 * script that tests $? sees these values */
enum { MC_EXIT_OK = 0, MC_EXIT_HELP = 1 /* -h: usage() is reached through the error path (multiclust.c:1500-1502) */, MC_EXIT_MEMORY_ALLOCATION = 3, MC_EXIT_FILE_OPEN_ERROR = 5, MC_EXIT_FILE_FORMAT_ERROR = 7,
       MC_EXIT_INVALID_CMDLINE = 8, MC_EXIT_INVALID_CMD_OPTION = 9, MC_EXIT_INVALID_CMD_ARGUMENT = 10, MC_EXIT_INVALID_USER_SETUP = 11, MC_EXIT_INTERNAL_ERROR = 13 };

typedef struct mc_cli_options {
	mc_options em;			/* the EM-layer options (mc_host.h) */
	const char *filename;		/* -f */
	const char *filename_file;	/* file part of filename (multiclust.c:1483-1492) */
	const char *path;		/* -d, default "./" */
	const char *outfile_name;	/* -o */
	int min_K, max_K;		/* -1 / -2 / -k, default 6 */
	int n_init;			/* -n, default 50 */
	int n_bootstrap;		/* -b */
	int n_rand_em_init;		/* -m (stored, never selects Rand-EM in the reference either) */
	int missing_value;		/* --missing */
	int R_format;			/* -R */
	int ploidy;			/* -p, default 2 */
	int seed_given;			/* -r seen: the reference only calls srand() then (multiclust.c:1592-1596) */
	int target_ll, target_revisit;	/* -u l / -u n */
	double desired_ll;
	int n_repeat;			/* -w n */
	unsigned int repeat_seconds, max_repeat_seconds;	/* -w t / -w m (minutes on the command line) */
	int write_files;
	int compact;
	int parallel;			/* -M */
	int device;			/* --device (extension): HIP device index */
	int n_gpus;			/* --gpus (extension): shard initialisations over devices device..device+n_gpus-1 */
	int n_streams;			/* --streams (extension): concurrent fits per device, each with its own context and
					 * stream; small data sets do not fill a GPU with one fit */
	const char *pfile, *qfile;	/* -P / -Q: initial parameters of the admixture model from files (read_file.c:880-959) */
	const char *afile;		/* -A: a partition of the individuals to compare the fitted one with (multiclust.c:1416-1418) */
} mc_cli_options;

typedef struct mc_cli_data {
	int I, L, ploidy, M;		/* M = max alleles at a locus */
	int missing_data;
	int interleaved;
	int *IL;			/* [I*ploidy][L] allele codes as read (reference dat->IL); released by the reader once geno exists */
	int32_t *uniquealleles;		/* [L] */
	int **L_alleles;		/* [L][..] ascending real alleles (phantom slot excluded) */
	uint8_t *geno;			/* [I][L][ploidy] allele indices, 0xFF missing */
	char **names;			/* [I] */
	int *locale;			/* [I] index into pops */
	char **pops;
	int numpops;
	int *i_p;			/* [numpops] individuals per locale */
	int T;
	int32_t *toff;			/* [L+1] */
} mc_cli_data;

/* read_file + summarize_alleles + sufficient_statistics (read_file.c:38-300,443-663), default (allele-code) mode */
int mc_read_structure(const mc_cli_options *opt, mc_cli_data *dat);
void mc_free_data(mc_cli_data *dat);

/* what a finished initialisation hands to the writers */
typedef struct mc_fit_view {
	int K, converged;
	double logL, aic, bic;
	const double *q;	/* [I][K] or [K]: slot pindex */
	const double *p;	/* [K][T] */
	const double *sik;	/* [I][K]: sum_lm d_iklm of the last E step, or vik */
} mc_fit_view;

/* read_afile (read_file.c:970-999): I cluster labels 1, 2, ... (white space between them) -> labels[i] - 1 and their number
 * *pK = largest label; returns 0 or the reference's exit status for the failure (file cannot be opened 5, contents 7) */
int mc_read_afile(const char *path, int I, int **labels, int *pK);
/* adj_rand(..., ADJUSTED_RAND_INDEX) (multiclust.c:1903-1985): adjusted Rand index of two labelings of n observations with k1
 * and k2 classes, the reference's sums in the reference's order; NaN (0 / 0) where it has it */
double mc_adjusted_rand(int n, int k1, int k2, const int *cl1, const int *cl2);
/* partition_admixture / partition_mixture (write_file.c:350-382, 585-603): MAP cluster per individual, count_K */
void mc_partition(const mc_cli_data *dat, const mc_fit_view *fit, int *I_K, int *count_K);
/* write_file_detail + popq_* + indivq_* (write_file.c:203-348, 398-475, 492-569, 618-732) */
int mc_write_results(const mc_cli_options *opt, const mc_cli_data *dat, const mc_fit_view *fit, const int *count_K);

#endif
