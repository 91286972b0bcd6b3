/*
 * mc_oracle.c -- TEST INFRASTRUCTURE ONLY (see mc_oracle.h).
 *
 * CPU restatement of MULTICLUST's EM hot path on flat arrays.  Every routine cites the reference
 * file:line it follows and keeps that code's operation order, so that on the golden vectors
 * (tests/golden/, dumped from the reference by oracle/ref_harness.c) it agrees bit for bit.
 * Parity status: PINNED (tests/test_oracle_golden.py).
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off (no FMA contraction: the reference's stock build has none).
 */
#include "mc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct mco_data {
	int I, L, ploidy, T;
	int32_t *ua;		/* uniquealleles[L] */
	int32_t *toff;		/* [L+1] prefix sums */
	uint8_t *geno;		/* [I][L][ploidy] */
	int32_t *ilm;		/* [I][T] allele counts (dat->ILM flattened) */
};

struct mco_model {
	const mco_data *dat;
	int K, nq, constrained_or_mix;
	double *p[3], *q[3];
	double *d;		/* [I][K][T] (admixture, fused==0) */
	double *sik;		/* [I][K] */
	double *u_p[3], *v_p[3], *u_q[3], *v_q[3];
	int qn;			/* options::q */
	int pindex, findex, tindex, delta_index;
	double logL;
	int n_iter, converged, stopped, accel_step, iter_stop, fatal;
	double A[9], Ainv[9], cutu[3];
};

/* ------------------------------------------------------------------ utilities */

double mco_lower_bound(double lb, int I, int ploidy)
{
	/* multiclust.c:812-813: MIN(lower_bound, 1.0/I/ploidy - 0.5/I/ploidy) */
	double alt = 1.0 / I / ploidy - 0.5 / I / ploidy;
	return lb < alt ? lb : alt;
}

void mco_srand(mco_rng *g, unsigned int seed)
{
	/* glibc stdlib/random_r.c __srandom_r, TYPE_3 (degree 31, separation 3) */
	int32_t word;
	if (seed == 0) seed = 1;
	g->r[0] = (int32_t)seed;
	word = (int32_t)seed;
	for (int i = 1; i < 31; i++) {
		long hi = word / 127773, lo = word % 127773;
		word = (int32_t)(16807 * lo - 2836 * hi);
		if (word < 0) word += 2147483647;
		g->r[i] = word;
	}
	g->f = 3; g->b = 0;
	for (int i = 0; i < 310; i++) (void)mco_rand(g);
}

int mco_rand(mco_rng *g)
{
	uint32_t v = (uint32_t)g->r[g->f] + (uint32_t)g->r[g->b];
	g->r[g->f] = (int32_t)v;
	if (++g->f >= 31) g->f = 0;
	if (++g->b >= 31) g->b = 0;
	return (int)(v >> 1);
}

mco_data *mco_data_create(int I, int L, int ploidy, const int32_t *ua, const uint8_t *geno)
{
	mco_data *d = calloc(1, sizeof *d);
	d->I = I; d->L = L; d->ploidy = ploidy;
	d->ua = malloc(sizeof(int32_t) * L);
	d->toff = malloc(sizeof(int32_t) * (L + 1));
	memcpy(d->ua, ua, sizeof(int32_t) * L);
	d->toff[0] = 0;
	for (int l = 0; l < L; l++) d->toff[l + 1] = d->toff[l] + ua[l];
	d->T = d->toff[L];
	size_t ng = (size_t)I * L * ploidy;
	d->geno = malloc(ng);
	memcpy(d->geno, geno, ng);
	/* read_file.c:633-663 sufficient_statistics (L_alleles != NULL branch): ILM[i][l][m] = #copies equal to allele m */
	d->ilm = calloc((size_t)I * d->T, sizeof(int32_t));
	for (int i = 0; i < I; i++)
		for (int l = 0; l < L; l++)
			for (int a = 0; a < ploidy; a++) {
				uint8_t m = geno[((size_t)i * L + l) * ploidy + a];
				if (m != MCO_MISSING_IDX && m < ua[l])
					d->ilm[(size_t)i * d->T + d->toff[l] + m]++;
			}
	return d;
}

void mco_data_free(mco_data *d)
{
	if (!d) return;
	free(d->ua); free(d->toff); free(d->geno); free(d->ilm); free(d);
}
int mco_data_T(const mco_data *d) { return d->T; }
const int32_t *mco_data_ilm(const mco_data *d) { return d->ilm; }

mco_model *mco_model_create(const mco_data *dat, const mco_options *o, int K)
{
	/* multiclust.c:1181-1265 allocate_model_for_k, flat */
	mco_model *m = calloc(1, sizeof *m);
	size_t KT = (size_t)K * dat->T;
	m->dat = dat; m->K = K;
	m->constrained_or_mix = (!o->admixture || o->eta_constrained);
	m->nq = m->constrained_or_mix ? K : dat->I * K;
	m->qn = o->accel_scheme >= MCO_QN ? o->accel_scheme - MCO_SQS3 : 1;	/* multiclust.c:820 */
	for (int s = 0; s < 3; s++) {
		m->p[s] = calloc(KT, sizeof(double));
		m->q[s] = calloc(m->nq, sizeof(double));
		m->u_p[s] = calloc(KT, sizeof(double));
		m->v_p[s] = calloc(KT, sizeof(double));
		m->u_q[s] = calloc(m->nq, sizeof(double));
		m->v_q[s] = calloc(m->nq, sizeof(double));
	}
	if (o->admixture && !o->fused)
		m->d = calloc((size_t)dat->I * KT, sizeof(double));
	m->sik = calloc((size_t)dat->I * K, sizeof(double));
	mco_model_reset(m);
	return m;
}

void mco_model_free(mco_model *m)
{
	if (!m) return;
	for (int s = 0; s < 3; s++) {
		free(m->p[s]); free(m->q[s]); free(m->u_p[s]); free(m->v_p[s]); free(m->u_q[s]); free(m->v_q[s]);
	}
	free(m->d); free(m->sik); free(m);
}

double *mco_model_p(mco_model *m, int s) { return m->p[s]; }
double *mco_model_q(mco_model *m, int s) { return m->q[s]; }
double *mco_model_sik(mco_model *m) { return m->sik; }
double *mco_model_u_p(mco_model *m, int j) { return m->u_p[j]; }
double *mco_model_u_q(mco_model *m, int j) { return m->u_q[j]; }
double *mco_model_v_q(mco_model *m, int j) { return m->v_q[j]; }
int mco_model_delta_index(const mco_model *m) { return m->delta_index; }
void mco_model_set_delta_index(mco_model *m, int delta_index) { m->delta_index = delta_index; }	/* model::delta_index (em_alg.c:1171) */
double *mco_model_v_p(mco_model *m, int j) { return m->v_p[j]; }
int mco_model_q_len(const mco_model *m) { return m->nq; }
double mco_model_logL(const mco_model *m) { return m->logL; }
int mco_model_n_iter(const mco_model *m) { return m->n_iter; }
int mco_model_converged(const mco_model *m) { return m->converged; }
int mco_model_pindex(const mco_model *m) { return m->pindex; }
int mco_model_findex(const mco_model *m) { return m->findex; }
int mco_model_tindex(const mco_model *m) { return m->tindex; }
int mco_model_fatal(const mco_model *m) { return m->fatal; }

void mco_model_reset(mco_model *m)
{
	/* rnd_init.c:58-71 (initialize_model) and multiclust.c:518-524 (maximize_likelihood) */
	m->n_iter = 0;
	m->logL = -INFINITY;
	m->converged = 0;
	m->stopped = 0;
	m->iter_stop = 0;
	m->accel_step = 0;
	m->fatal = 0;
	m->pindex = m->findex = m->tindex = 0;
	m->delta_index = 0;
}

/* ------------------------------------------------------------------ simplex.c:109-143 */

void mco_michelot_project(double *x, int len, double sum, double min)
{
	int fixed[len];
	int n = len;
	for (int i = 0; i < len; i++) fixed[i] = 0;
	while (n) {
		double csum = 0.0;
		for (int i = 0; i < len; i++) csum += x[i];
		double a = (csum - sum) / n;
		int can_terminate = 1;
		for (int i = 0; i < len; i++)
			if (!fixed[i]) {
				x[i] -= a;
				if (x[i] < min) {
					x[i] = min;
					fixed[i] = 1;
					n--;
					can_terminate = 0;
				}
			}
		if (can_terminate) break;
	}
}

static void project_q(const mco_options *o, mco_model *m, double *q, int i)
{
	/* simplex.c:29-45 */
	if (o->admixture && !o->eta_constrained)
		mco_michelot_project(q + (size_t)i * m->K, m->K, 1.0, o->lower_bound);
	else
		mco_michelot_project(q, m->K, 1.0, o->lower_bound);
}

static void project_p(const mco_options *o, mco_model *m, double *p, int k, int l)
{
	/* simplex.c:47-69 */
	const mco_data *d = m->dat;
	mco_michelot_project(p + (size_t)k * d->T + d->toff[l], d->ua[l], 1.0, o->lower_bound);
}

/* ------------------------------------------------------------------ admixture E step */

/* em_alg.c:291-486 e_step_admixture_orig (default mode: L_alleles != NULL, m_start == 0) */
static double e_step_admixture_ref(const mco_options *o, mco_model *m)
{
	const mco_data *dat = m->dat;
	const int I = dat->I, L = dat->L, K = m->K, T = dat->T;
	const double *P = m->p[m->findex], *Q = m->q[m->findex];
	double ld[K];
	double loglik = 0;

	for (int i = 0; i < I; i++) {
		const double *qi = o->eta_constrained ? Q : Q + (size_t)i * K;
		for (int l = 0; l < L; l++)
			for (int mm = 0; mm < dat->ua[l]; mm++) {
				int c = dat->toff[l] + mm;
				int n = dat->ilm[(size_t)i * T + c];
				if (n == 0) {
					for (int k = 0; k < K; k++)
						m->d[((size_t)i * K + k) * T + c] = 0;
					continue;
				}
				double tmp = 0;
				for (int k = 0; k < K; k++) {
					ld[k] = qi[k] * P[(size_t)k * T + c];
					tmp += ld[k];
				}
				for (int k = 0; k < K; k++)
					m->d[((size_t)i * K + k) * T + c] = n * ld[k] / tmp;
				loglik += n * log(tmp);
			}
	}
	/* what the writers read (write_file.c:359-381): sum_lm d_iklm, in M-step order */
	for (int i = 0; i < I; i++)
		for (int k = 0; k < K; k++) {
			double e = 0;
			const double *row = m->d + ((size_t)i * K + k) * T;
			for (int c = 0; c < T; c++) e += row[c];
			m->sik[(size_t)i * K + k] = e;
		}
	return loglik;
}

/* em_alg.c:592-754 m_step_admixture_orig */
static void m_step_admixture_ref(const mco_options *o, mco_model *m)
{
	const mco_data *dat = m->dat;
	const int I = dat->I, L = dat->L, K = m->K, T = dat->T;
	double *P = m->p[m->tindex], *Q = m->q[m->tindex];
	double temp;

	if (o->eta_constrained) {		/* em_alg.c:604-648 */
		temp = 0.0;
		for (int k = 0; k < K; k++) {
			Q[k] = 0;
			for (int l = 0; l < L; l++)
				for (int mm = 0; mm < dat->ua[l]; mm++)
					for (int i = 0; i < I; i++)
						Q[k] += m->d[((size_t)i * K + k) * T + dat->toff[l] + mm];
			temp += Q[k];
		}
		for (int k = 0; k < K; k++) Q[k] /= temp;
		if (o->do_projection) project_q(o, m, Q, 0);
	} else {				/* em_alg.c:650-702 */
		for (int i = 0; i < I; i++) {
			temp = 0.0;
			for (int k = 0; k < K; k++) {
				double e = 0;
				const double *row = m->d + ((size_t)i * K + k) * T;
				for (int c = 0; c < T; c++) e += row[c];
				Q[(size_t)i * K + k] = e;
				temp += e;
			}
			for (int k = 0; k < K; k++) Q[(size_t)i * K + k] /= temp;
			if (o->do_projection) project_q(o, m, Q, i);
		}
	}
	for (int k = 0; k < K; k++)		/* em_alg.c:706-752 */
		for (int l = 0; l < L; l++) {
			temp = 0.0;
			for (int mm = 0; mm < dat->ua[l]; mm++) {
				int c = dat->toff[l] + mm;
				double s = 0;
				for (int i = 0; i < I; i++)
					s += m->d[((size_t)i * K + k) * T + c];
				P[(size_t)k * T + c] = s;
				temp += s;
			}
			for (int mm = 0; mm < dat->ua[l]; mm++)
				P[(size_t)k * T + dat->toff[l] + mm] /= temp;
			if (o->do_projection) project_p(o, m, P, k, l);
		}
}

/* Fused E+M (never materialises d_iklm): the re-association the HIP kernels use:
 * r = n/tmp; S_ik = q_ik * sum_c P_kc r_ic; N_kc = P_kc * sum_i q_ik r_ic.  Same normalisation and
 * projection as em_alg.c:685-701,734-750.  Returns logL of the findex parameters. */
static double em_step_admixture_fused(const mco_options *o, mco_model *m, int do_mstep)
{
	const mco_data *dat = m->dat;
	const int I = dat->I, L = dat->L, K = m->K, T = dat->T, pl = dat->ploidy;
	const double *P = m->p[m->findex], *Q = m->q[m->findex];
	double *Pt = malloc(sizeof(double) * (size_t)T * K);	/* [T][K] for locality */
	double *At = calloc((size_t)T * K, sizeof(double));
	double *etac = calloc(K, sizeof(double));
	double acc[K];
	double loglik = 0;

	for (int k = 0; k < K; k++)
		for (int c = 0; c < T; c++) Pt[(size_t)c * K + k] = P[(size_t)k * T + c];

	for (int i = 0; i < I; i++) {
		const double *qi = o->eta_constrained ? Q : Q + (size_t)i * K;
		for (int k = 0; k < K; k++) acc[k] = 0;
		for (int l = 0; l < L; l++) {
			const uint8_t *g = dat->geno + ((size_t)i * L + l) * pl;
			for (int a = 0; a < pl; a++) {
				uint8_t mm = g[a];
				if (mm == MCO_MISSING_IDX || mm >= dat->ua[l]) continue;
				int seen = 0;		/* visit each distinct allele once, with its count */
				for (int b = 0; b < a; b++) if (g[b] == mm) seen = 1;
				if (seen) continue;
				int n = 0;
				for (int b = a; b < pl; b++) n += (g[b] == mm);
				int c = dat->toff[l] + mm;
				const double *pc = Pt + (size_t)c * K;
				double tmp = 0;
				for (int k = 0; k < K; k++) tmp += qi[k] * pc[k];
				double r = n / tmp;
				for (int k = 0; k < K; k++) {
					acc[k] += pc[k] * r;
					At[(size_t)c * K + k] += qi[k] * r;
				}
				loglik += n * log(tmp);
			}
		}
		for (int k = 0; k < K; k++) {
			double s = qi[k] * acc[k];
			m->sik[(size_t)i * K + k] = s;
			etac[k] += s;
		}
	}
	if (do_mstep) {
		double *Pn = m->p[m->tindex], *Qn = m->q[m->tindex];
		double temp;
		if (o->eta_constrained) {
			temp = 0;
			for (int k = 0; k < K; k++) temp += etac[k];
			for (int k = 0; k < K; k++) Qn[k] = etac[k] / temp;
			if (o->do_projection) project_q(o, m, Qn, 0);
		} else {
			for (int i = 0; i < I; i++) {
				temp = 0;
				for (int k = 0; k < K; k++) temp += m->sik[(size_t)i * K + k];
				for (int k = 0; k < K; k++) Qn[(size_t)i * K + k] = m->sik[(size_t)i * K + k] / temp;
				if (o->do_projection) project_q(o, m, Qn, i);
			}
		}
		for (int k = 0; k < K; k++)
			for (int l = 0; l < L; l++) {
				temp = 0;
				for (int mm = 0; mm < dat->ua[l]; mm++) {
					int c = dat->toff[l] + mm;
					double s = Pt[(size_t)c * K + k] * At[(size_t)c * K + k];
					Pn[(size_t)k * T + c] = s;
					temp += s;
				}
				for (int mm = 0; mm < dat->ua[l]; mm++)
					Pn[(size_t)k * T + dat->toff[l] + mm] /= temp;
				if (o->do_projection) project_p(o, m, Pn, k, l);
			}
	}
	free(Pt); free(At); free(etac);
	return loglik;
}

/* log_likelihood.c:96-147 logL_admixture */
static double logL_admixture(const mco_options *o, mco_model *m, int which)
{
	const mco_data *dat = m->dat;
	const int I = dat->I, L = dat->L, K = m->K, T = dat->T;
	const double *P = m->p[which], *Q = m->q[which];
	double loglt1 = 0.0;
	for (int i = 0; i < I; i++) {
		const double *qi = o->eta_constrained ? Q : Q + (size_t)i * K;
		for (int l = 0; l < L; l++)
			for (int mm = 0; mm < dat->ua[l]; mm++) {
				int c = dat->toff[l] + mm;
				int n = dat->ilm[(size_t)i * T + c];
				if (n == 0) continue;
				double temp = 0.0;
				for (int k = 0; k < K; k++) temp += qi[k] * P[(size_t)k * T + c];
				loglt1 += n * log(temp);
			}
	}
	return loglt1;
}

/* ------------------------------------------------------------------ mixture model */

/* em_alg.c:763-897 e_step_mixture; vik lives in m->sik */
static double e_step_mixture(mco_model *m)
{
	const mco_data *dat = m->dat;
	const int I = dat->I, L = dat->L, K = m->K, T = dat->T;
	const double *P = m->p[m->findex], *eta = m->q[m->findex];
	double log_etak[K];
	double loglik = 0;
	for (int k = 0; k < K; k++) log_etak[k] = log(eta[k]);
	for (int i = 0; i < I; i++) {
		double max_ll = -INFINITY;
		double *v = m->sik + (size_t)i * K;
		for (int k = 0; k < K; k++) {
			v[k] = log_etak[k];
			for (int l = 0; l < L; l++)
				for (int mm = 0; mm < dat->ua[l]; mm++) {
					int c = dat->toff[l] + mm;
					int n = dat->ilm[(size_t)i * T + c];
					if (n == 0 || P[(size_t)k * T + c] == 0.0) continue;
					v[k] += n * log(P[(size_t)k * T + c]);
				}
			if (v[k] > max_ll) max_ll = v[k];
		}
		double temp = 0;
		for (int k = 0; k < K; k++) {
			v[k] = exp(v[k] - max_ll);
			temp += v[k];
		}
		for (int k = 0; k < K; k++) v[k] /= temp;
		loglik += log(temp) + max_ll;
	}
	return loglik;
}

/* em_alg.c:907-1011 m_step_mixture */
static void m_step_mixture(const mco_options *o, mco_model *m)
{
	const mco_data *dat = m->dat;
	const int I = dat->I, L = dat->L, K = m->K, T = dat->T;
	double *P = m->p[m->tindex], *eta = m->q[m->tindex];
	double temp = 0.0;
	for (int k = 0; k < K; k++) {
		eta[k] = 0;
		for (int i = 0; i < I; i++) eta[k] += m->sik[(size_t)i * K + k];
		temp += eta[k];
	}
	for (int k = 0; k < K; k++) eta[k] /= temp;
	if (o->do_projection) project_q(o, m, eta, 0);
	for (int k = 0; k < K; k++)
		for (int l = 0; l < L; l++) {
			temp = 0.0;
			for (int mm = 0; mm < dat->ua[l]; mm++) {
				int c = dat->toff[l] + mm;
				double s = o->lower_bound;	/* em_alg.c:972: p_lower_bound */
				for (int i = 0; i < I; i++) {
					int n = dat->ilm[(size_t)i * T + c];
					if (n) s += m->sik[(size_t)i * K + k] * n;
				}
				P[(size_t)k * T + c] = s;
				temp += s;
			}
			for (int mm = 0; mm < dat->ua[l]; mm++)
				P[(size_t)k * T + dat->toff[l] + mm] /= temp;
			if (o->do_projection) project_p(o, m, P, k, l);
		}
}

/* log_likelihood.c:157-232 logL_mixture */
static double logL_mixture(mco_model *m, int which)
{
	const mco_data *dat = m->dat;
	const int I = dat->I, L = dat->L, K = m->K, T = dat->T;
	const double *P = m->p[which], *eta = m->q[which];
	double temp_vik[K], log_etak[K];
	double loglt1 = 0.0;
	for (int k = 0; k < K; k++) log_etak[k] = log(eta[k]);
	for (int i = 0; i < I; i++) {
		double max_exp = -INFINITY;
		for (int k = 0; k < K; k++) {
			temp_vik[k] = 0.0;
			for (int l = 0; l < L; l++)
				for (int mm = 0; mm < dat->ua[l]; mm++) {
					int c = dat->toff[l] + mm;
					int n = dat->ilm[(size_t)i * T + c];
					if (n == 0) continue;
					temp_vik[k] += n * log(P[(size_t)k * T + c]);
				}
			temp_vik[k] += log_etak[k];
			if (temp_vik[k] > max_exp) max_exp = temp_vik[k];
		}
		double temp_exp = exp(max_exp), scale_exp = 0.0;
		int flag_out_range = 0;
		/* NOT the reference: with every temp_vik[k] NaN or -inf (or one +inf) max_exp is infinite and the reference's loop below
		 * never ends (inf / 2 = inf, log_likelihood.c:212-221): no reference answer exists, so nothing pins this case.  The
		 * individual's term is NaN here, as in k_mix_finalize; a checker that spins is of no use to a test. */
		if (!(max_exp > -INFINITY && max_exp < INFINITY)) {
			loglt1 = loglt1 + NAN;
			continue;
		}
		if (temp_exp == 0.0 || temp_exp == HUGE_VAL) {
			flag_out_range = 1;
			scale_exp = (temp_exp == HUGE_VAL) ? max_exp : -max_exp;
			do {
				scale_exp *= 0.5;
				temp_exp = exp(scale_exp);
			} while (temp_exp == HUGE_VAL);
			scale_exp = max_exp - scale_exp;
		}
		if (flag_out_range)
			for (int k = 0; k < K; k++) temp_vik[k] -= scale_exp;
		temp_exp = 0.0;
		for (int k = 0; k < K; k++) temp_exp = temp_exp + exp(temp_vik[k]);
		loglt1 = loglt1 + log(temp_exp) + scale_exp;
	}
	return loglt1;
}

double mco_log_likelihood(const mco_data *d, const mco_options *o, mco_model *m, int which)
{
	(void)d;
	return o->admixture ? logL_admixture(o, m, which) : logL_mixture(m, which);
}

/* ------------------------------------------------------------------ initialisation */

void mco_initialize_from_partition(const mco_data *dat, const mco_options *o, mco_model *m, const uint8_t *assign)
{
	/* rnd_init.c:456-482: d[i][k][l][m] = 1 (not += 1) for each copy a of allele m assigned to k; missing
	 * copies match nothing.  Then m_step_admixture (rnd_init.c:356).  Sums of 0/1 are exact in any order. */
	const int I = dat->I, L = dat->L, K = m->K, T = dat->T, pl = dat->ploidy;
	if (!o->fused) {
		memset(m->d, 0, sizeof(double) * (size_t)I * K * T);
		for (int i = 0; i < I; i++)
			for (int l = 0; l < L; l++)
				for (int a = 0; a < pl; a++) {
					size_t gi = ((size_t)i * L + l) * pl + a;
					uint8_t mm = dat->geno[gi];
					if (mm == MCO_MISSING_IDX || mm >= dat->ua[l]) continue;
					m->d[((size_t)i * K + assign[gi]) * T + dat->toff[l] + mm] = 1;
				}
		for (int i = 0; i < I; i++)
			for (int k = 0; k < K; k++) {
				double e = 0;
				for (int c = 0; c < T; c++) e += m->d[((size_t)i * K + k) * T + c];
				m->sik[(size_t)i * K + k] = e;
			}
		m_step_admixture_ref(o, m);
		return;
	}
	/* fused: same integers without the dense array */
	double *N = calloc((size_t)K * T, sizeof(double));
	double *etac = calloc(K, sizeof(double));
	double *P = m->p[m->tindex], *Q = m->q[m->tindex];
	memset(m->sik, 0, sizeof(double) * (size_t)I * K);
	for (int i = 0; i < I; i++)
		for (int l = 0; l < L; l++)
			for (int a = 0; a < pl; a++) {
				size_t gi = ((size_t)i * L + l) * pl + a;
				uint8_t mm = dat->geno[gi];
				if (mm == MCO_MISSING_IDX || mm >= dat->ua[l]) continue;
				int dup = 0;
				for (int b = 0; b < a; b++) {
					size_t gb = ((size_t)i * L + l) * pl + b;
					if (dat->geno[gb] == mm && assign[gb] == assign[gi]) dup = 1;
				}
				if (dup) continue;
				m->sik[(size_t)i * K + assign[gi]] += 1;
				N[(size_t)assign[gi] * T + dat->toff[l] + mm] += 1;
			}
	if (o->eta_constrained) {
		double temp = 0;
		for (int i = 0; i < I; i++) for (int k = 0; k < K; k++) etac[k] += m->sik[(size_t)i * K + k];
		for (int k = 0; k < K; k++) temp += etac[k];
		for (int k = 0; k < K; k++) Q[k] = etac[k] / temp;
		if (o->do_projection) project_q(o, m, Q, 0);
	} else {
		for (int i = 0; i < I; i++) {
			double temp = 0;
			for (int k = 0; k < K; k++) temp += m->sik[(size_t)i * K + k];
			for (int k = 0; k < K; k++) Q[(size_t)i * K + k] = m->sik[(size_t)i * K + k] / temp;
			if (o->do_projection) project_q(o, m, Q, i);
		}
	}
	for (int k = 0; k < K; k++)
		for (int l = 0; l < L; l++) {
			double temp = 0;
			for (int mm = 0; mm < dat->ua[l]; mm++) temp += N[(size_t)k * T + dat->toff[l] + mm];
			for (int mm = 0; mm < dat->ua[l]; mm++)
				P[(size_t)k * T + dat->toff[l] + mm] = N[(size_t)k * T + dat->toff[l] + mm] / temp;
			if (o->do_projection) project_p(o, m, P, k, l);
		}
	free(N); free(etac);
}

void mco_random_initialize_admixture(const mco_data *dat, const mco_options *o, mco_model *m, mco_rng *g)
{
	/* rnd_init.c:460-467: loops i, l, a; k = rand() % K for every copy (missing ones too) */
	size_t n = (size_t)dat->I * dat->L * dat->ploidy;
	uint8_t *assign = malloc(n);
	for (size_t j = 0; j < n; j++) assign[j] = (uint8_t)(mco_rand(g) % m->K);
	mco_initialize_from_partition(dat, o, m, assign);
	free(assign);
}

/* ------------------------------------------------------------------ Rand-EM (rnd_init.c:123-160, 412-444) */

void mco_random_allele_center(const mco_data *dat, int K, mco_rng *g, uint8_t *ilk)
{
	/* rnd_init.c:496-583, default mode (L_alleles != NULL, m_start = 0): per locus the center alleles -- every allele slot when
	 * the locus has fewer than K (the phantom slot of a locus with missing data included: it matches nothing), else K distinct
	 * slots by rejection -- then every copy in i, a order: the first cluster whose center it carries, else rand() % K */
	const int I = dat->I, L = dat->L, pl = dat->ploidy;
	int center[K > 0 ? K : 1];
	if (K == 1) {
		memset(ilk, 0, (size_t)I * L * pl);
		return;
	}
	for (int l = 0; l < L; l++) {
		const int M = dat->ua[l];
		if (M < K) {
			for (int k = 0; k < M; k++) center[k] = k;
			for (int k = M; k < K; k++) center[k] = -1;
		} else {
			for (int k = 0; k < K; k++) {
				int flag;
				center[k] = mco_rand(g) % M;
				do {
					flag = 0;
					for (int j = 0; j < k; j++)
						if (center[k] == center[j]) {
							center[k] = mco_rand(g) % M;
							flag = 1;
							break;
						}
				} while (flag == 1);
			}
		}
		for (int i = 0; i < I; i++)
			for (int a = 0; a < pl; a++) {
				const size_t gi = ((size_t)i * L + l) * pl + a;
				const uint8_t mm = dat->geno[gi];
				int flag = 0;
				for (int k = 0; k < K; k++) {
					if (center[k] == -1) break;
					if (mm != MCO_MISSING_IDX && mm == center[k]) {
						ilk[gi] = (uint8_t)k;
						flag = 1;
						break;
					}
				}
				if (!flag) ilk[gi] = (uint8_t)(mco_rand(g) % K);
			}
	}
}

void mco_initialize_parameters_admixture(const mco_data *dat, const mco_options *o, mco_model *m, const uint8_t *ilk)
{
	/* rnd_init.c:603-705: eta = (1 + copies assigned to k, missing ones too) / (ploidy L [I] + K); p = (1 + copies of allele m
	 * assigned to k) / sum over m; no projection */
	const int I = dat->I, L = dat->L, pl = dat->ploidy, K = m->K, T = dat->T;
	double *P = m->p[m->tindex], *Q = m->q[m->tindex];
	if (o->eta_constrained) {
		const double temp = pl * L * I + K;
		for (int k = 0; k < K; k++) Q[k] = 1;
		for (size_t j = 0; j < (size_t)I * L * pl; j++) Q[ilk[j]]++;
		for (int k = 0; k < K; k++) Q[k] /= temp;
	} else {
		const double temp = pl * L + K;
		for (int i = 0; i < I; i++) {
			for (int k = 0; k < K; k++) Q[(size_t)i * K + k] = 1;
			for (int l = 0; l < L; l++)
				for (int a = 0; a < pl; a++) Q[(size_t)i * K + ilk[((size_t)i * L + l) * pl + a]]++;
			for (int k = 0; k < K; k++) Q[(size_t)i * K + k] /= temp;
		}
	}
	for (size_t x = 0; x < (size_t)K * T; x++) P[x] = 1.0;
	for (int i = 0; i < I; i++)
		for (int l = 0; l < L; l++)
			for (int a = 0; a < pl; a++) {
				const size_t gi = ((size_t)i * L + l) * pl + a;
				const uint8_t mm = dat->geno[gi];
				if (mm != MCO_MISSING_IDX && mm < dat->ua[l]) P[(size_t)ilk[gi] * T + dat->toff[l] + mm]++;
			}
	for (int k = 0; k < K; k++)
		for (int l = 0; l < L; l++) {
			double temp = 0;
			for (int mm = 0; mm < dat->ua[l]; mm++) temp += P[(size_t)k * T + dat->toff[l] + mm];
			for (int mm = 0; mm < dat->ua[l]; mm++) P[(size_t)k * T + dat->toff[l] + mm] /= temp;
		}
}

void mco_random_individual_center(const mco_data *dat, int K, mco_rng *g, int *I_K)
{
	/* rnd_init.c:192-259 */
	const int I = dat->I, T = dat->T;
	int center[K > 0 ? K : 1];
	if (K == 1) {
		for (int i = 0; i < I; i++) I_K[i] = 0;
		return;
	}
	for (int k = 0; k < K; k++) {
		int flag;
		center[k] = mco_rand(g) % I;
		do {
			flag = 0;
			for (int j = 0; j < k; j++)
				if (center[k] == center[j]) {
					center[k] = mco_rand(g) % I;
					flag = 1;
					break;
				}
		} while (flag == 1);
	}
	for (int i = 0; i < I; i++) {
		I_K[i] = 0;
		if (i == center[0]) continue;
		double min_count_diff = INFINITY;
		for (int k = 0; k < K; k++) {
			if (i == center[k]) { I_K[i] = k; break; }
			double count_diff = 0;
			for (int c = 0; c < T; c++) count_diff += abs(dat->ilm[(size_t)i * T + c] - dat->ilm[(size_t)center[k] * T + c]);
			if (count_diff < min_count_diff) { I_K[i] = k; min_count_diff = count_diff; }
		}
	}
}

void mco_initialize_parameters_mixture(const mco_data *dat, mco_model *m, const int *I_K)
{
	/* rnd_init.c:268-339, including its loop nest: p[k][l][m] = 1 inside the loop over k, then EVERY individual's count is
	 * added to its own cluster's entry, so cluster k collects its counts K - k times */
	const int I = dat->I, L = dat->L, K = m->K, T = dat->T;
	double *P = m->p[m->tindex], *Q = m->q[m->tindex];
	for (int k = 0; k < K; k++) Q[k] = 1;
	for (int i = 0; i < I; i++) Q[I_K[i]]++;
	for (int k = 0; k < K; k++) Q[k] /= I + K;
	for (int k = 0; k < K; k++)
		for (int l = 0; l < L; l++)
			for (int mm = 0; mm < dat->ua[l]; mm++) {
				const int c = dat->toff[l] + mm;
				P[(size_t)k * T + c] = 1.0;
				for (int i = 0; i < I; i++) {
					if (dat->ilm[(size_t)i * T + c] == 0) continue;
					P[(size_t)I_K[i] * T + c] += dat->ilm[(size_t)i * T + c];
				}
			}
	for (int k = 0; k < K; k++)
		for (int l = 0; l < L; l++) {
			double temp = 0.0;
			for (int mm = 0; mm < dat->ua[l]; mm++) temp += P[(size_t)k * T + dat->toff[l] + mm];
			for (int mm = 0; mm < dat->ua[l]; mm++) P[(size_t)k * T + dat->toff[l] + mm] /= temp;
		}
}

double mco_em_e_step(const mco_data *d, const mco_options *o, mco_model *m);

void mco_randem_initialize(const mco_data *dat, const mco_options *o, mco_model *m, mco_rng *g, int n_rand_em_init, double *ll_out)
{
	/* randem_initialize_admixture (rnd_init.c:412-444) / randem_initialize_mixture (123-160) */
	const int n_init = m->K > 1 ? n_rand_em_init : 1;
	const size_t n = (size_t)dat->I * dat->L * dat->ploidy;
	double max_logL = -INFINITY;
	if (o->admixture) {
		uint8_t *ilk = malloc(n), *best = calloc(n, 1);
		for (int i = 0; i < n_init; i++) {
			mco_random_allele_center(dat, m->K, g, ilk);
			mco_initialize_parameters_admixture(dat, o, m, ilk);
			const double logL = mco_em_e_step(dat, o, m);
			if (ll_out) ll_out[i] = logL;
			if (logL > max_logL) {
				max_logL = logL;
				memcpy(best, ilk, n);
			}
		}
		mco_initialize_parameters_admixture(dat, o, m, best);
		free(ilk); free(best);
	} else {
		int *I_K = calloc(dat->I, sizeof(int)), *best = calloc(dat->I, sizeof(int));
		for (int i = 0; i < n_init; i++) {
			mco_random_individual_center(dat, m->K, g, I_K);
			mco_initialize_parameters_mixture(dat, m, I_K);
			const double logL = mco_em_e_step(dat, o, m);
			if (ll_out) ll_out[i] = logL;
			if (logL > max_logL) {
				max_logL = logL;
				memcpy(best, I_K, sizeof(int) * dat->I);
			}
		}
		mco_initialize_parameters_mixture(dat, m, best);
		free(I_K); free(best);
	}
}

/* ------------------------------------------------------------------ EM control flow */

/* em_alg.c:163-182 */
static int converged(const mco_options *o, mco_model *m, double loglik)
{
	int stop = 1;
	double abs_diff = 0, rel_diff = 0;
	if (o->abs_error) abs_diff = fabs(loglik - m->logL);
	if (o->rel_error) rel_diff = abs_diff / fabs(m->logL);
	if (o->abs_error && abs_diff > o->abs_error) stop &= 0;
	if (o->rel_error && rel_diff > o->rel_error) stop &= 0;
	if (stop) m->converged = 1;
	return stop;
}

/* em_alg.c:101-161 stop + stop_condition (time limit not restated: wall-clock dependent) */
static int stop_fn(const mco_options *o, mco_model *m, double loglik)
{
	m->n_iter++;
	if (isnan(loglik)) { m->fatal = 1; m->stopped = 1; return 1; }
	if (o->max_iter && m->n_iter > o->max_iter) {
		m->iter_stop = 1;
		m->stopped = 1;
	} else {
		m->stopped = converged(o, m, loglik);
	}
	if (loglik < m->logL && !m->stopped) { m->fatal = 2; m->stopped = 1; return 1; }
	m->accel_step = 0;
	m->logL = loglik;
	return m->stopped;
}

double mco_e_step(const mco_data *d, const mco_options *o, mco_model *m)
{
	(void)d;
	if (o->admixture)
		return o->fused ? em_step_admixture_fused(o, m, 0) : e_step_admixture_ref(o, m);
	return e_step_mixture(m);
}

int mco_em_step(const mco_data *d, const mco_options *o, mco_model *m)
{
	/* em_alg.c:195-207 */
	double ll;
	(void)d;
	if (o->admixture) {
		if (o->fused) {
			ll = em_step_admixture_fused(o, m, 1);
		} else {
			ll = e_step_admixture_ref(o, m);
			m_step_admixture_ref(o, m);
		}
	} else {
		ll = e_step_mixture(m);
		m_step_mixture(o, m);
	}
	return stop_fn(o, m, ll);
}

int mco_em_2_steps(const mco_data *d, const mco_options *o, mco_model *m)
{
	/* em_alg.c:1072-1211 (default build: neither OLDWAY nor NEWWAY) */
	const size_t KT = (size_t)m->K * m->dat->T;
	m->findex = m->pindex;
	m->tindex = (m->findex + 1) % 3;
	for (int j = 0; j < 2; j++) {
		if (mco_em_step(d, o, m)) return 1;
		double *dp = j ? m->v_p[m->delta_index] : m->u_p[m->delta_index];
		double *dq = j ? m->v_q[m->delta_index] : m->u_q[m->delta_index];
		for (size_t x = 0; x < KT; x++) dp[x] = m->p[m->tindex][x] - m->p[m->findex][x];
		for (int x = 0; x < m->nq; x++) dq[x] = m->q[m->tindex][x] - m->q[m->findex][x];
		m->findex = m->tindex;
		m->tindex = (m->findex + 1) % 3;
		if (m->tindex == m->pindex) m->tindex = (m->tindex + 1) % 3;
	}
	m->delta_index = (m->delta_index + 1) % m->qn;
	return 0;
}

double mco_step_size(const mco_data *d, const mco_options *o, mco_model *m)
{
	/* accel_em.c:130-243: eta terms first, then p in k,l,m order (flat [K][T] order is the same) */
	const size_t KT = (size_t)m->K * m->dat->T;
	const double *uq = m->u_q[m->delta_index], *vq = m->v_q[m->delta_index];
	const double *up = m->u_p[m->delta_index], *vp = m->v_p[m->delta_index];
	double utu = 0, utvu = 0, vutvu = 0, s;
	(void)d;
	for (int x = 0; x < m->nq; x++) {
		utu += uq[x] * uq[x];
		utvu += uq[x] * (vq[x] - uq[x]);
		vutvu += (vq[x] - uq[x]) * (vq[x] - uq[x]);
	}
	for (size_t x = 0; x < KT; x++) {
		utu += up[x] * up[x];
		utvu += up[x] * (vp[x] - up[x]);
		vutvu += (vp[x] - up[x]) * (vp[x] - up[x]);
	}
	if (o->accel_scheme == MCO_SQS1) s = utu / utvu;
	else if (o->accel_scheme == MCO_SQS2) s = utvu / vutvu;
	else if (o->accel_scheme == MCO_SQS3) {
		if (sqrt(utu) < 1e-8) return NAN;
		s = -sqrt(utu / vutvu);
	} else if (o->accel_scheme == MCO_QN) s = -utu / utvu;
	else s = -1;
	if (o->accel_scheme < MCO_QN && s > -1) s = -1;
	return s;
}

double mco_accelerated_update(const mco_data *d, const mco_options *o, mco_model *m, double s)
{
	/* accel_em.c:422-551 */
	const mco_data *dat = m->dat;
	const int K = m->K, T = dat->T;
	m->delta_index = m->delta_index ? m->delta_index - 1 : m->qn - 1;
	const double *up = m->u_p[m->delta_index], *vp = m->v_p[m->delta_index];
	const double *uq = m->u_q[m->delta_index], *vq = m->v_q[m->delta_index];
	double *Pt = m->p[m->tindex], *Pp = m->p[m->pindex];
	double *Qt = m->q[m->tindex], *Qp = m->q[m->pindex];
	for (int l = 0; l < dat->L; l++)
		for (int k = 0; k < K; k++) {
			for (int mm = 0; mm < dat->ua[l]; mm++) {
				size_t x = (size_t)k * T + dat->toff[l] + mm;
				if (o->accel_scheme == MCO_QN)
					Pt[x] = Pp[x] + up[x] + s * vp[x];
				else
					Pt[x] = Pp[x] - 2 * s * up[x] + s * s * (vp[x] - up[x]);
			}
			if (o->do_projection) project_p(o, m, Pt, k, l);
		}
	if (o->admixture && !o->eta_constrained) {
		for (int i = 0; i < dat->I; i++) {
			for (int k = 0; k < K; k++) {
				size_t x = (size_t)i * K + k;
				if (o->accel_scheme == MCO_QN)
					Qt[x] = Qp[x] + uq[x] + s * vq[x];
				else
					Qt[x] = Qp[x] - 2 * s * uq[x] + s * s * (vq[x] - uq[x]);
			}
			if (o->do_projection) project_q(o, m, Qt, i);
		}
	} else {
		for (int k = 0; k < K; k++) {
			if (o->accel_scheme == MCO_QN)
				Qt[k] = Qp[k] + uq[k] + s * vq[k];
			else
				Qt[k] = Qp[k] - 2 * s * uq[k] + s * s * (vq[k] - uq[k]);
		}
		if (o->do_projection) project_q(o, m, Qt, 0);
	}
	double ll = mco_log_likelihood(d, o, m, m->tindex);
	m->delta_index = (m->delta_index + 1) % m->qn;
	return ll;
}

/* accel_em.c:262-419 qn_accelerated_update (q = 1..3, closed-form inverse) */
static double qn_accelerated_update(const mco_data *d, const mco_options *o, mco_model *m)
{
	const mco_data *dat = m->dat;
	const int K = m->K, T = dat->T, I = dat->I, q = m->qn;
	const int indiv = (o->admixture && !o->eta_constrained);
	int vindex = m->delta_index ? m->delta_index - 1 : q - 1;
	int uindex = vindex ? vindex - 1 : q - 1;
	int q1, q2, j, n;
	double utu, utv, det;
	double *A = m->A, *Ainv = m->Ainv;

	q1 = m->delta_index;
	j = 0;
	do {
		q2 = m->delta_index;
		n = 0;
		do {
			utu = 0; utv = 0;
			for (int k = 0; k < K; k++) {
				if (indiv) {
					for (int i = 0; i < I; i++) {
						utu += m->u_q[q1][(size_t)i * K + k] * m->u_q[q2][(size_t)i * K + k];
						utv += m->u_q[q1][(size_t)i * K + k] * m->v_q[q2][(size_t)i * K + k];
					}
				} else {
					utu += m->u_q[q1][k] * m->u_q[q2][k];
					utv += m->u_q[q1][k] * m->v_q[q2][k];
				}
				for (int c = 0; c < T; c++) {
					utu += m->u_p[q1][(size_t)k * T + c] * m->u_p[q2][(size_t)k * T + c];
					utv += m->u_p[q1][(size_t)k * T + c] * m->v_p[q2][(size_t)k * T + c];
				}
			}
			m->cutu[n] = utu;
			A[j * q + n] = utu - utv;
			n++;
			q2 = (q2 + 1) % q;
		} while (q2 != m->delta_index);
		q1 = (q1 + 1) % q;
		j++;
	} while (q1 != m->delta_index);

	if (q == 1) {
		Ainv[0] = 1 / A[0];
	} else if (q == 2) {
		det = A[0] * A[3] - A[1] * A[2];
		Ainv[0] = A[3] / det;
		Ainv[3] = A[0] / det;
		Ainv[1] = -A[1] / det;
		Ainv[2] = -A[2] / det;
	} else {
		det = A[0] * (A[4] * A[8] - A[5] * A[7])
			- A[1] * (A[8] * A[3] - A[5] * A[6])
			+ A[2] * (A[3] * A[7] - A[4] * A[6]);
		Ainv[0] = (A[4] * A[8] - A[5] * A[7]) / det;
		Ainv[1] = (A[2] * A[7] - A[1] * A[8]) / det;
		Ainv[2] = (A[1] * A[5] - A[2] * A[4]) / det;
		Ainv[3] = (A[5] * A[6] - A[3] * A[8]) / det;
		Ainv[4] = (A[0] * A[8] - A[2] * A[6]) / det;
		Ainv[5] = (A[2] * A[3] - A[0] * A[5]) / det;
		Ainv[6] = (A[3] * A[7] - A[4] * A[6]) / det;
		Ainv[7] = (A[1] * A[6] - A[0] * A[7]) / det;
		Ainv[8] = (A[0] * A[4] - A[1] * A[3]) / det;
	}

	double *Pt = m->p[m->tindex], *Pp = m->p[m->pindex];
	double *Qt = m->q[m->tindex], *Qp = m->q[m->pindex];
	for (int x = 0; x < m->nq; x++) Qt[x] = Qp[x] + m->u_q[uindex][x];
	for (size_t x = 0; x < (size_t)K * T; x++) Pt[x] = Pp[x] + m->u_p[uindex][x];
	q1 = m->delta_index;
	j = 0;
	do {
		n = 0;
		q2 = m->delta_index;
		do {
			for (int x = 0; x < m->nq; x++)
				Qt[x] += m->v_q[q1][x] * Ainv[j * q + n] * m->cutu[n];
			for (size_t x = 0; x < (size_t)K * T; x++)
				Pt[x] += m->v_p[q1][x] * Ainv[j * q + n] * m->cutu[n];
			q2 = (q2 + 1) % q;
			n++;
		} while (q2 != m->delta_index);
		q1 = (q1 + 1) % q;
		j++;
	} while (q1 != m->delta_index);

	if (o->do_projection) {
		if (indiv)
			for (int i = 0; i < I; i++) project_q(o, m, Qt, i);
		else
			project_q(o, m, Qt, 0);
		for (int k = 0; k < K; k++)
			for (int l = 0; l < dat->L; l++) project_p(o, m, Pt, k, l);
	}
	return mco_log_likelihood(d, o, m, m->tindex);
}

int mco_accelerated_em_step(const mco_data *d, const mco_options *o, mco_model *m, double *trace4)
{
	/* accel_em.c:35-114 */
	int n_adjust = 0;
	double emll, ll = 0, s = 0;

	mco_em_2_steps(d, o, m);
	if (m->stopped) return 1;
	emll = mco_log_likelihood(d, o, m, m->findex);
	if (o->accel_scheme <= MCO_QN) {
		s = mco_step_size(d, o, m);
		if (isnan(s) || isinf(s)) goto EM_EXIT;
	}
	do {
		if (o->accel_scheme <= MCO_QN)
			ll = mco_accelerated_update(d, o, m, s);
		else
			ll = qn_accelerated_update(d, o, m);
		if (o->adjust_step && ll < emll)
			s = (s - 1) / 2;
	} while (n_adjust++ < o->adjust_step && ll < emll && s < -1);
	if (trace4) { trace4[0] = emll; trace4[1] = s; trace4[2] = ll; trace4[3] = ll > emll; }
	if (ll > emll) {
		m->pindex = m->tindex;
		m->accel_step = 1;
		return 0;
	}
EM_EXIT:
	if (trace4 && !(ll > emll)) { trace4[0] = emll; trace4[1] = s; trace4[2] = ll; trace4[3] = 0; }
	m->pindex = m->findex;
	return 0;
}

void mco_em(const mco_data *d, const mco_options *o, mco_model *m)
{
	/* em_alg.c:44-90 */
	int stop = 0;
	if (m->K == 1) {
		mco_em_step(d, o, m);
		m->logL = mco_log_likelihood(d, o, m, m->tindex);
		return;
	}
	while (m->n_iter < o->n_init_iter && !stop)
		stop = mco_em_step(d, o, m);
	for (int i = 1; i < m->qn; i++) {
		mco_em_2_steps(d, o, m);
		m->pindex = m->findex;
	}
	if (m->converged) return;
	do {
		if (!o->accel_scheme)
			stop = mco_em_step(d, o, m);
		else
			stop = mco_accelerated_em_step(d, o, m, NULL);
	} while (!stop);
}

/* ------------------------------------------------------------------ multiclust.c:471-656 */

void mco_summary_reset(mco_summary *s)
{
	/* multiclust.c:478-486; max_logL is reset by estimate_model (377) */
	s->first_max_logL = -INFINITY;
	s->max_logL = -INFINITY;
	s->n_init = 0;
	s->n_total_iter = 0;
	s->n_maxll_times = 0;
	s->n_maxll_init = -1;
	s->n_max_iter = 0;
	s->ever_converged = 0;
	s->best_unit = -1;
}

void mco_summary_add(const mco_options *o, mco_summary *s, int unit, double logL, int conv, int n_iter, int time_stop)
{
	/* multiclust.c:534-560 */
	if (conv) s->ever_converged = 1;
	if (conv || (!s->n_init && time_stop)) {
		s->n_total_iter += n_iter;
		if (s->n_max_iter < n_iter) s->n_max_iter = n_iter;
		s->n_init++;
	}
	/* converged(opt, mod, first_max_logL) with mod->logL = logL (em_alg.c:163-182) */
	int seen = 1;
	double abs_diff = 0, rel_diff = 0;
	if (o->abs_error) abs_diff = fabs(s->first_max_logL - logL);
	if (o->rel_error) rel_diff = abs_diff / fabs(logL);
	if (o->abs_error && abs_diff > o->abs_error) seen = 0;
	if (o->rel_error && rel_diff > o->rel_error) seen = 0;
	if (conv && seen) {
		s->n_maxll_times++;
	} else if (conv && logL > s->first_max_logL) {
		s->n_maxll_times = 1;
		s->first_max_logL = logL;
		s->n_maxll_init = s->n_init;
	}
	if (logL > s->max_logL) {
		s->max_logL = logL;
		s->best_unit = unit;
	}
}

void mco_maximize_likelihood(const mco_data *d, const mco_options *o, mco_model *m, mco_rng *g, int n_units,
			     double *per_unit, mco_summary *s)
{
	mco_summary_reset(s);
	for (int u = 0; u < n_units; u++) {
		/* multiclust.c:518-524; note delta_index is NOT reset between initialisations */
		int delta_keep = m->delta_index;
		mco_model_reset(m);
		m->delta_index = delta_keep;
		m->logL = 0.0;
		mco_random_initialize_admixture(d, o, m, g);
		m->logL = -INFINITY;	/* initialize_model, rnd_init.c:59 */
		mco_em(d, o, m);
		if (per_unit) {
			per_unit[4 * u + 0] = m->logL; per_unit[4 * u + 1] = m->converged;
			per_unit[4 * u + 2] = m->n_iter; per_unit[4 * u + 3] = m->pindex;
		}
		mco_summary_add(o, s, u, m->logL, m->converged, m->n_iter, 0);
	}
}

double mco_em_e_step(const mco_data *d, const mco_options *o, mco_model *m)
{
	/* em_alg.c:219-233: E, M, E; the second E step's log likelihood (no stop(), no iteration count) */
	if (o->admixture) {
		if (o->fused) {
			em_step_admixture_fused(o, m, 1);
		} else {
			e_step_admixture_ref(o, m);
			m_step_admixture_ref(o, m);
		}
	} else {
		e_step_mixture(m);
		m_step_mixture(o, m);
	}
	return mco_e_step(d, o, m);
}
