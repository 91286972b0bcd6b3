/*
 * ref_bind.c -- the binding of INTEGRATION.md EXECUTED: the five EM-layer entry points the reference's driver calls
 * (`nm` of multiclust.o + cline.o + read_file.o + write_file.o + message.o + bootstrap.o leaves exactly these undefined:
 * initialize_model, em, converged, aic, bic), implemented on ref_glue.c's functions, i.e. on libmulticlust_host.so /
 * libmulticlust_hip.so.
 *
 * TEST INFRASTRUCTURE (oracle side).  oracle/Makefile links this file and ref_glue.c with the reference's OWN, unmodified
 * multiclust.c (its main(), option parser, maximize_likelihood, run_bootstrap), cline.c, read_file.c, write_file.c, message.c
 * and bootstrap.c where they lie under /root/reference -- and WITHOUT its em_alg.c, accel_em.c, log_likelihood.c, simplex.c
 * and rnd_init.c -- into oracle/_ref/multiclust_ref_hip: the reference program with its EM layer replaced by the MI355X
 * path, which is what a maintainer following INTEGRATION.md would build.  The binary travels to the GPU box (the sources do
 * not); tests/test_gpu_refbind.py runs it on the command lines of the reference's own goldens.  No GPU: every fit fails.
 *
 * The libc rand() stream stays the program's one random stream: the reference seeds it (srand in parse_options,
 * multiclust.c:1592-1596) and anything else in the program may keep drawing from it; an initialisation borrows its
 * state (initstate/setstate expose glibc's TYPE_3 table: word 0 = 5 * rear + type, words 1..31 = the table), lets the device
 * draw the partition from it and hands the advanced state back.
 */
#define _DEFAULT_SOURCE		/* initstate / setstate under -std=c17 */
#include "multiclust.h"
#include "mc_host.h"
#include "ref_glue.h"

static struct {
	const data *dat;	/* data set the flat genotype was made from */
	int ***ilm;		/* its count array at that time: parametric_bootstrap() swaps dat->ILM (bootstrap.c:35-41) */
	uint8_t *flat;		/* observed genotype, allele indices */
	mc_options mo;
	mc_data md;
	mc_model *mm;
	int K;
	double *sik;
} B;

static void bind_fatal(const char *what)
{
	fprintf(stderr, "multiclust_ref_hip: %s\n", what);
	exit(EXIT_FAILURE);
}

static void unbind(void)
{
	if (B.mm) mc_model_free(B.mm);
	B.mm = NULL;
}

static void bind(options *opt, data *dat, model *mod)
{
	int fresh = 0;
	if (opt->pfile || opt->qfile) bind_fatal("starting values from files (-p/-q) are not bound");
	if (B.dat != dat) {
		unbind();
		free(B.flat);
		B.flat = mcamd_flatten_genotypes(dat);
		if (!B.flat) bind_fatal("out of memory");
		B.dat = dat;
		B.ilm = dat->ILM;
		atexit(unbind);
		fresh = 1;
	}
	/* parametric_bootstrap() swaps dat->ILM (bootstrap.c:35-41).  Not bound here: the reference's own -b run aborts in its
	 * second model (free(): invalid pointer, tests/test_gpu_cli.py), so there is nothing to compare an executed binding with;
	 * mcamd_bootstrap_model() is the documented call and the replicate generator is pinned in tests/test_gpu_bootstrap.py */
	if (dat->ILM != B.ilm) bind_fatal("bootstrap replicates (-b) are not bound in this test program");
	const uint8_t *geno = B.flat;
	/* after the reference's synchronize() (multiclust.c:807): opt->lower_bound is already MIN(bound, 0.5 / I / ploidy), and
	 * mc_synchronize applies the same rule again without changing it */
	if (mcamd_options(opt, dat, geno, opt->lower_bound, &B.mo, &B.md)) bind_fatal("options do not carry over");
	if (fresh || !B.mm || B.K != mod->K) {
		unbind();
		const char *dev = getenv("MCAMD_DEVICE");
		if (mcamd_allocate_model_for_k(&B.mo, &B.md, mod, dev ? atoi(dev) : 0, &B.mm))
			bind_fatal("no device context (there is no CPU fallback)");
		B.K = mod->K;
		free(B.sik);
		if (!(B.sik = malloc(sizeof(double) * (size_t)dat->I * mod->K))) bind_fatal("out of memory");
	}
}

/* parse_options() tests errno after every numeric argument without ever clearing it (multiclust.c:1538 and the like): whatever
 * the constructors of the GPU runtime's libraries left there must be gone before main() runs */
__attribute__((constructor)) static void clear_errno_before_main(void) { errno = 0; }

/* ---- the libc stream on loan ---- */
static int32_t *libc_words;

static void stream_take(mc_rng *g)
{
	static char parked[128];
	libc_words = (int32_t *)initstate(1, parked, sizeof parked);	/* switching away makes glibc record the rear index */
	if (!libc_words || libc_words[0] % 5 != 3) bind_fatal("libc's rand() is not glibc's TYPE_3 generator");
	memcpy(g->r, libc_words + 1, sizeof g->r);
	g->b = libc_words[0] / 5;
	g->f = (g->b + 3) % 31;
}

static void stream_give(const mc_rng *g)
{
	memcpy(libc_words + 1, g->r, sizeof g->r);
	libc_words[0] = 5 * g->b + 3;
	setstate((char *)libc_words);
}

/* ---- the five entry points (prototypes: multiclust.h:371-388) ---- */
int initialize_model(options *opt, data *dat, model *mod)		/* rnd_init.c:54 */
{
	mc_rng rng;
	const int errno_before = errno;
	/* the stream is taken BEFORE anything touches the GPU runtime: creating the first HIP context draws from (or reseeds) libc's
	 * generator -- found when this program first ran on a GPU box: same seed, different initialisations from run to run.
	 * While the state is on loan libc runs on a parked table, so such draws cannot disturb the program's stream */
	stream_take(&rng);
	bind(opt, dat, mod);
	const int err = mcamd_initialize_model(&B.mo, &B.md, mod, B.mm, &rng);
	stream_give(&rng);
	errno = errno_before;
	return err;
}

void em(options *opt, data *dat, model *mod)				/* em_alg.c:44 */
{
	mc_rng held;
	/* the reference reads errno as "did my last allocation fail" without clearing it first (write_file.c:417,424,...: the popq /
	 * indivq writers return silently when it is non-zero): what the GPU runtime's system calls leave there must not reach it */
	const int errno_before = errno;
	if (!B.mm || B.dat != dat || B.K != mod->K) bind_fatal("em() without initialize_model()");
	stream_take(&held);						/* parked for the duration of the GPU work, handed back as it was */
	mcamd_em(&B.mo, &B.md, mod, B.mm);
	if (mcamd_fetch_parameters(opt, dat, mod, B.mm) || mcamd_fetch_expected_counts(opt, dat, mod, B.mm, B.sik))
		bind_fatal("cannot read the fit back");
	stream_give(&held);
	errno = errno_before;
	if (opt->admixture) {
		/* the writers only ever sum diklm over (l, m) (write_file.c:366,455,540): the sums go into the first cell */
		int l0 = 0;
		while (l0 < dat->L && dat->uniquealleles[l0] < 1) l0++;
		if (l0 == dat->L) bind_fatal("no locus with an allele");
		for (int i = 0; i < dat->I; i++)
			for (int k = 0; k < mod->K; k++) {
				for (int l = 0; l < dat->L; l++)
					for (int m = 0; m < dat->uniquealleles[l]; m++) mod->diklm[i][k][l][m] = 0.0;
				mod->diklm[i][k][l0][0] = B.sik[(size_t)i * mod->K + k];
			}
	}
}

int converged(options *opt, model *mod, double loglik)		/* em_alg.c:163 */
{
	mc_options mo;
	mc_model t;
	mc_make_options(&mo);
	mo.abs_error = opt->abs_error;
	mo.rel_error = opt->rel_error;
	memset(&t, 0, sizeof t);
	t.logL = mod->logL;
	const int stop = mc_converged(&mo, &t, loglik);
	if (t.converged) mod->converged = 1;
	return stop;
}

double aic(model *mod) { return mc_aic(mod->max_logL, mod->no_parameters); }			/* log_likelihood.c:70 */
double bic(data *dat, model *mod) { return mc_bic(mod->max_logL, mod->no_parameters, dat->I); }	/* log_likelihood.c:82 */
