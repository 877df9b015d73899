/*
 * potus_oracle.c -- CPU fp64 restatement of the reference hot path.  TEST INFRASTRUCTURE
 * (see potus_oracle.h: parity unpinned; only tests/, smoke() and bench.py's cpu_baseline
 * may use it).
 *
 * Part 1 restates scripts/model/poll_model_2020.stan (and the
 * poll_model_2020_no_mode_adjustment.stan variant) -- every block cites its lines.
 * Part 2 restates Stan 2.24's adaptive diagonal-metric NUTS as invoked at
 * scripts/model/final_2016.R:533-541; those sources are third-party (CmdStan 2.24.1,
 * not in the reference tree) and are cited by their upstream file names.
 */
#define _POSIX_C_SOURCE 200809L
#include "potus_oracle.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------ */
/* model                                                                                */
/* ------------------------------------------------------------------------------------ */
struct oracle_model {
  int Nn, Ns, T, S, P, M, Pop, variant, D, ncols;
  int *state, *day_state, *day_nat, *poll_state, *poll_nat, *mode_state, *mode_nat, *pop_state,
      *pop_nat, *y_nat, *n_nat, *y_state, *n_state; /* 0-based copies */
  double *unadj_nat, *unadj_state, *prior, *w;
  double sigma_c, sigma_m, sigma_pop, sigma_nn, sigma_ns, sigma_e;
  double *LB, *LT, *LW; /* col-major SxS lower Cholesky factors (stan:52-54) */
  /* offsets of the parameter blocks inside q (stan:56-69, declaration order) */
  int o_zT, o_Z, o_c, o_m, o_pop, o_mue, o_rho, o_ze, o_nn, o_ns, o_zb;
  /* CSR of polls by day for the fast variant */
  int *day_ptr, *day_idx; /* state polls */
};

static void *xmalloc(size_t n) {
  void *p = calloc(n ? n : 1, 1);
  if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
  return p;
}
static int *dup_idx(const int32_t *src, int n, int shift) {
  int *d = (int *)xmalloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; i++) d[i] = src ? (int)src[i] - shift : 0;
  return d;
}
static double *dup_d(const double *src, int n) {
  double *d = (double *)xmalloc(sizeof(double) * (size_t)n);
  if (src) memcpy(d, src, sizeof(double) * (size_t)n);
  return d;
}

/* Eigen::LLT-equivalent dense Cholesky, lower, column-major (cholesky_decompose, stan:52-54) */
static int cholesky_lower(const double *A, double *L, int n) {
  memset(L, 0, sizeof(double) * (size_t)n * n);
  for (int j = 0; j < n; j++) {
    double s = A[j + (size_t)j * n];
    for (int k = 0; k < j; k++) s -= L[j + (size_t)k * n] * L[j + (size_t)k * n];
    if (!(s > 0.0)) return -1;
    double ljj = sqrt(s);
    L[j + (size_t)j * n] = ljj;
    for (int i = j + 1; i < n; i++) {
      double t = A[i + (size_t)j * n];
      for (int k = 0; k < j; k++) t -= L[i + (size_t)k * n] * L[j + (size_t)k * n];
      L[i + (size_t)j * n] = t / ljj;
    }
  }
  return 0;
}

#define FAIL(...) do { if (err) snprintf(err, (size_t)errlen, __VA_ARGS__); return NULL; } while (0)

static int check_range(const int32_t *v, int n, int lo, int hi) {
  if (n > 0 && !v) return -1;
  for (int i = 0; i < n; i++) if (v[i] < lo || v[i] > hi) return i + 1;
  return 0;
}

oracle_model *oracle_model_create(const potus_data *d, char *err, int errlen) {
  if (!d) FAIL("null data");
  int full = d->variant == POTUS_VARIANT_FULL;
  if (d->variant != POTUS_VARIANT_FULL && d->variant != POTUS_VARIANT_NO_MODE) FAIL("unknown variant");
  if (d->S < 1 || d->T < 2 || d->P < 1 || d->N_state_polls < 0 || d->N_national_polls < 0) FAIL("bad sizes");
  if (full && (d->M < 1 || d->Pop < 1)) FAIL("bad sizes M/Pop");
  /* declared bounds, stan:9-17 (state's declared upper bound is S+1, but S+1 indexes out of
     range at stan:97, where Stan would throw; reject it here) */
  if (check_range(d->state, d->N_state_polls, 1, d->S)) FAIL("state out of range [1,S]");
  if (check_range(d->day_state, d->N_state_polls, 1, d->T)) FAIL("day_state out of range [1,T]");
  if (check_range(d->day_national, d->N_national_polls, 1, d->T)) FAIL("day_national out of range [1,T]");
  if (check_range(d->poll_state, d->N_state_polls, 1, d->P)) FAIL("poll_state out of range [1,P]");
  if (check_range(d->poll_national, d->N_national_polls, 1, d->P)) FAIL("poll_national out of range [1,P]");
  if (full) {
    if (check_range(d->poll_mode_state, d->N_state_polls, 1, d->M)) FAIL("poll_mode_state out of range");
    if (check_range(d->poll_mode_national, d->N_national_polls, 1, d->M)) FAIL("poll_mode_national out of range");
    if (check_range(d->poll_pop_state, d->N_state_polls, 1, d->Pop)) FAIL("poll_pop_state out of range");
    if (check_range(d->poll_pop_national, d->N_national_polls, 1, d->Pop)) FAIL("poll_pop_national out of range");
    for (int i = 0; i < d->N_state_polls; i++)
      if (!(d->unadjusted_state[i] >= 0.0 && d->unadjusted_state[i] <= 1.0)) FAIL("unadjusted_state outside [0,1]");
    for (int i = 0; i < d->N_national_polls; i++)
      if (!(d->unadjusted_national[i] >= 0.0 && d->unadjusted_national[i] <= 1.0)) FAIL("unadjusted_national outside [0,1]");
  }
  /* binomial_logit argument checks (stan:130-131): 0 <= n <= N */
  for (int i = 0; i < d->N_state_polls; i++)
    if (d->n_democrat_state[i] < 0 || d->n_democrat_state[i] > d->n_two_share_state[i]) FAIL("n_democrat_state outside [0,N]");
  for (int i = 0; i < d->N_national_polls; i++)
    if (d->n_democrat_national[i] < 0 || d->n_democrat_national[i] > d->n_two_share_national[i]) FAIL("n_democrat_national outside [0,N]");
  int S = d->S;
  /* cov_matrix[S] (stan:37): symmetric, positive definite */
  for (int i = 0; i < S; i++)
    for (int j = 0; j < i; j++) {
      double a = d->state_covariance_0[i + (size_t)j * S], b = d->state_covariance_0[j + (size_t)i * S];
      if (fabs(a - b) > 1e-8 * fmax(1.0, fmax(fabs(a), fabs(b)))) FAIL("state_covariance_0 is not symmetric");
    }

  oracle_model *m = (oracle_model *)xmalloc(sizeof(*m));
  m->Nn = d->N_national_polls; m->Ns = d->N_state_polls; m->T = d->T; m->S = S; m->P = d->P;
  m->M = full ? d->M : 0; m->Pop = full ? d->Pop : 0; m->variant = d->variant;
  m->state = dup_idx(d->state, m->Ns, 1);
  m->day_state = dup_idx(d->day_state, m->Ns, 1);
  m->day_nat = dup_idx(d->day_national, m->Nn, 1);
  m->poll_state = dup_idx(d->poll_state, m->Ns, 1);
  m->poll_nat = dup_idx(d->poll_national, m->Nn, 1);
  m->mode_state = dup_idx(full ? d->poll_mode_state : NULL, m->Ns, 1);
  m->mode_nat = dup_idx(full ? d->poll_mode_national : NULL, m->Nn, 1);
  m->pop_state = dup_idx(full ? d->poll_pop_state : NULL, m->Ns, 1);
  m->pop_nat = dup_idx(full ? d->poll_pop_national : NULL, m->Nn, 1);
  m->y_nat = dup_idx(d->n_democrat_national, m->Nn, 0);
  m->n_nat = dup_idx(d->n_two_share_national, m->Nn, 0);
  m->y_state = dup_idx(d->n_democrat_state, m->Ns, 0);
  m->n_state = dup_idx(d->n_two_share_state, m->Ns, 0);
  m->unadj_nat = dup_d(full ? d->unadjusted_national : NULL, m->Nn);
  m->unadj_state = dup_d(full ? d->unadjusted_state : NULL, m->Ns);
  m->prior = dup_d(d->mu_b_prior, S);
  m->w = dup_d(d->state_weights, S);
  m->sigma_c = d->sigma_c; m->sigma_m = d->sigma_m; m->sigma_pop = d->sigma_pop;
  m->sigma_nn = d->sigma_measure_noise_national; m->sigma_ns = d->sigma_measure_noise_state;
  m->sigma_e = d->sigma_e_bias;

  /* transformed data, stan:42-55 */
  double nsd2 = 0.0;
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) nsd2 += m->w[i] * d->state_covariance_0[i + (size_t)j * S] * m->w[j];
  double nsd = sqrt(nsd2); /* national_cov_matrix_error_sd, stan:43 */
  double sc[3] = {d->polling_bias_scale / nsd, d->mu_b_T_scale / nsd, d->random_walk_scale / nsd};
  double **Ls[3] = {&m->LB, &m->LT, &m->LW};
  double *tmp = (double *)xmalloc(sizeof(double) * (size_t)S * S);
  for (int a = 0; a < 3; a++) {
    for (size_t i = 0; i < (size_t)S * S; i++) tmp[i] = d->state_covariance_0[i] * (sc[a] * sc[a]); /* stan:48-50 */
    *Ls[a] = (double *)xmalloc(sizeof(double) * (size_t)S * S);
    if (cholesky_lower(tmp, *Ls[a], S)) { free(tmp); oracle_model_free(m); FAIL("state_covariance_0 is not positive definite"); }
  }
  free(tmp);

  /* parameter layout, stan:56-69 */
  int o = 0;
  m->o_zT = o; o += S;
  m->o_Z = o; o += S * m->T;
  m->o_c = o; o += m->P;
  if (full) {
    m->o_m = o; o += m->M;
    m->o_pop = o; o += m->Pop;
    m->o_mue = o; o += 1;
    m->o_rho = o; o += 1;
    m->o_ze = o; o += m->T;
  } else { m->o_m = m->o_pop = m->o_mue = m->o_rho = m->o_ze = -1; }
  m->o_nn = o; o += m->Nn;
  m->o_ns = o; o += m->Ns;
  m->o_zb = o; o += S;
  m->D = o;
  /* output row: params + TPs (stan:72-83) + GQ (stan:135) */
  int tp = S * m->T + m->P + (full ? m->M + m->Pop + m->T : 0) + S + m->T + 1 + (full ? 1 : 0) + m->Ns + m->Nn;
  m->ncols = POTUS_N_SAMPLER_COLS + m->D + tp + m->T * S;

  /* state polls grouped by day (fast variant) */
  m->day_ptr = (int *)xmalloc(sizeof(int) * (size_t)(m->T + 1));
  m->day_idx = (int *)xmalloc(sizeof(int) * (size_t)m->Ns);
  for (int i = 0; i < m->Ns; i++) m->day_ptr[m->day_state[i] + 1]++;
  for (int t = 0; t < m->T; t++) m->day_ptr[t + 1] += m->day_ptr[t];
  int *fill = (int *)xmalloc(sizeof(int) * (size_t)m->T);
  for (int i = 0; i < m->Ns; i++) { int t = m->day_state[i]; m->day_idx[m->day_ptr[t] + fill[t]++] = i; }
  free(fill);
  return m;
}

void oracle_model_free(oracle_model *m) {
  if (!m) return;
  free(m->state); free(m->day_state); free(m->day_nat); free(m->poll_state); free(m->poll_nat);
  free(m->mode_state); free(m->mode_nat); free(m->pop_state); free(m->pop_nat);
  free(m->y_nat); free(m->n_nat); free(m->y_state); free(m->n_state);
  free(m->unadj_nat); free(m->unadj_state); free(m->prior); free(m->w);
  free(m->LB); free(m->LT); free(m->LW); free(m->day_ptr); free(m->day_idx);
  free(m);
}
int oracle_num_params(const oracle_model *m) { return m->D; }
int oracle_num_columns(const oracle_model *m) { return m->ncols; }
void oracle_cholesky_factors(const oracle_model *m, double *L_B, double *L_T, double *L_W) {
  size_t n = sizeof(double) * (size_t)m->S * m->S;
  memcpy(L_B, m->LB, n); memcpy(L_T, m->LT, n); memcpy(L_W, m->LW, n);
}

/* Stan Math log_inv_logit / log1m_inv_logit / inv_logit */
static double log_inv_logit(double x) { return x > 0 ? -log1p(exp(-x)) : x - log1p(exp(x)); }
static double inv_logit(double x) { return x >= 0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x)); }

/* y = L x (L lower, col-major) and y = L^T x */
static void lower_mv(const double *L, const double *x, double *y, int n) {
  for (int i = 0; i < n; i++) y[i] = 0.0;
  for (int k = 0; k < n; k++) { double xk = x[k]; for (int i = k; i < n; i++) y[i] += L[i + (size_t)k * n] * xk; }
}
static void lower_tmv(const double *L, const double *x, double *y, int n) {
  for (int k = 0; k < n; k++) { double s = 0.0; for (int i = k; i < n; i++) s += L[i + (size_t)k * n] * x[i]; y[k] = s; }
}

/* Everything the model block computes (stan:70-113), kept so write_array can reuse it. */
typedef struct {
  double *mu_b, *mu_c, *mu_m, *mu_pop, *e_bias, *pb, *nat_avg, *eta_s, *eta_n;
  double nat_pb, sigma_rho, mu_e, rho;
} tparams;

static tparams tp_alloc(const oracle_model *m) {
  tparams t; memset(&t, 0, sizeof(t));
  t.mu_b = (double *)xmalloc(sizeof(double) * (size_t)m->S * m->T);
  t.mu_c = (double *)xmalloc(sizeof(double) * (size_t)m->P);
  t.mu_m = (double *)xmalloc(sizeof(double) * (size_t)(m->M + 1));
  t.mu_pop = (double *)xmalloc(sizeof(double) * (size_t)(m->Pop + 1));
  t.e_bias = (double *)xmalloc(sizeof(double) * (size_t)m->T);
  t.pb = (double *)xmalloc(sizeof(double) * (size_t)m->S);
  t.nat_avg = (double *)xmalloc(sizeof(double) * (size_t)m->T);
  t.eta_s = (double *)xmalloc(sizeof(double) * (size_t)(m->Ns + 1));
  t.eta_n = (double *)xmalloc(sizeof(double) * (size_t)(m->Nn + 1));
  return t;
}
static void tp_free(tparams *t) {
  free(t->mu_b); free(t->mu_c); free(t->mu_m); free(t->mu_pop); free(t->e_bias); free(t->pb);
  free(t->nat_avg); free(t->eta_s); free(t->eta_n);
}

/* transformed parameters, literal: stan:70-113 */
static void transformed_parameters(const oracle_model *m, const double *q, tparams *t) {
  const int S = m->S, T = m->T, full = m->variant == POTUS_VARIANT_FULL;
  const double *zT = q + m->o_zT, *Z = q + m->o_Z, *zb = q + m->o_zb;
  lower_mv(m->LB, zb, t->pb, S);                                      /* stan:77 */
  t->nat_pb = 0.0; for (int s = 0; s < S; s++) t->nat_pb += t->pb[s] * m->w[s]; /* stan:79 */
  double *col = t->mu_b + (size_t)(T - 1) * S;
  lower_mv(m->LT, zT, col, S);
  for (int s = 0; s < S; s++) col[s] += m->prior[s];                  /* stan:85 */
  double *tmp = (double *)xmalloc(sizeof(double) * (size_t)S);
  for (int i = 1; i <= T - 1; i++) {                                  /* stan:86 */
    int tt = T - 1 - i;
    lower_mv(m->LW, Z + (size_t)tt * S, tmp, S);
    for (int s = 0; s < S; s++) t->mu_b[s + (size_t)tt * S] = tmp[s] + t->mu_b[s + (size_t)(tt + 1) * S];
  }
  free(tmp);
  for (int u = 0; u < T; u++) {                                       /* stan:87 */
    double a = 0.0; for (int s = 0; s < S; s++) a += t->mu_b[s + (size_t)u * S] * m->w[s];
    t->nat_avg[u] = a;
  }
  for (int p = 0; p < m->P; p++) t->mu_c[p] = q[m->o_c + p] * m->sigma_c; /* stan:88 */
  if (full) {
    for (int i = 0; i < m->M; i++) t->mu_m[i] = q[m->o_m + i] * m->sigma_m;       /* stan:89 */
    for (int i = 0; i < m->Pop; i++) t->mu_pop[i] = q[m->o_pop + i] * m->sigma_pop; /* stan:90 */
    t->mu_e = 0.0 + 0.02 * q[m->o_mue];                               /* stan:62 offset/multiplier */
    t->rho = inv_logit(q[m->o_rho]);                                  /* stan:63 lub_constrain(0,1) */
    const double *ze = q + m->o_ze;
    t->e_bias[0] = ze[0] * m->sigma_e;                                /* stan:91 */
    t->sigma_rho = sqrt(1.0 - t->rho * t->rho) * m->sigma_e;          /* stan:92 */
    for (int u = 1; u < T; u++)                                       /* stan:93 */
      t->e_bias[u] = t->mu_e + t->rho * (t->e_bias[u - 1] - t->mu_e) + ze[u] * t->sigma_rho;
  }
  for (int i = 0; i < m->Ns; i++) {                                   /* stan:95-104 */
    int s = m->state[i], d = m->day_state[i];
    double e = t->mu_b[s + (size_t)d * S] + t->mu_c[m->poll_state[i]];
    if (full) e += t->mu_m[m->mode_state[i]] + t->mu_pop[m->pop_state[i]] + m->unadj_state[i] * t->e_bias[d];
    e += q[m->o_ns + i] * m->sigma_ns + t->pb[s];
    t->eta_s[i] = e;
  }
  for (int j = 0; j < m->Nn; j++) {                                   /* stan:105-112 */
    int d = m->day_nat[j];
    double e = t->nat_avg[d] + t->mu_c[m->poll_nat[j]];
    if (full) e += t->mu_m[m->mode_nat[j]] + t->mu_pop[m->pop_nat[j]] + m->unadj_nat[j] * t->e_bias[d];
    e += q[m->o_nn + j] * m->sigma_nn + t->nat_pb;
    t->eta_n[j] = e;
  }
}

/* priors + Jacobians (stan:62-63, 117-128): value and gradient contributions that do not
   involve the likelihood.  `~` drops constants; Jacobian terms are kept in full. */
static double priors_and_jacobian(const oracle_model *m, const double *q, const tparams *t, double *grad) {
  const int full = m->variant == POTUS_VARIANT_FULL;
  double lp = 0.0;
  for (int i = 0; i < m->D; i++) {
    if (full && (i == m->o_mue || i == m->o_rho)) continue;
    lp += -0.5 * q[i] * q[i];   /* std_normal on every raw_* block, stan:117-121,125-128 */
    grad[i] += -q[i];
  }
  if (full) {
    double x = q[m->o_mue];
    lp += log(0.02);                                   /* offset_multiplier Jacobian, stan:62 */
    lp += -0.5 * (t->mu_e / 0.02) * (t->mu_e / 0.02); /* mu_e_bias ~ normal(0, 0.02), stan:123 */
    grad[m->o_mue] += -x;
    double r = t->rho;
    lp += log(r) + log1p(-r);                          /* lub_constrain Jacobian, stan:63 */
    lp += -0.5 * ((r - 0.7) / 0.1) * ((r - 0.7) / 0.1); /* rho_e_bias ~ normal(0.7, 0.1), stan:124 */
    grad[m->o_rho] += (1.0 - 2.0 * r) + (-(r - 0.7) / 0.01) * r * (1.0 - r);
  }
  return lp;
}

/* likelihood value and residuals r = y - N*inv_logit(eta) (stan:130-131) */
static double likelihood(const oracle_model *m, const tparams *t, double *r_s, double *r_n) {
  double lp = 0.0;
  for (int i = 0; i < m->Ns; i++) {
    double e = t->eta_s[i], y = m->y_state[i], N = m->n_state[i];
    lp += y * log_inv_logit(e) + (N - y) * log_inv_logit(-e);
    r_s[i] = y - N * inv_logit(e);
  }
  for (int j = 0; j < m->Nn; j++) {
    double e = t->eta_n[j], y = m->y_nat[j], N = m->n_nat[j];
    lp += y * log_inv_logit(e) + (N - y) * log_inv_logit(-e);
    r_n[j] = y - N * inv_logit(e);
  }
  return lp;
}

/* adjoints shared by both gradient variants: everything except the mu_b walk */
static void small_block_adjoints(const oracle_model *m, const double *q, const tparams *t,
                                 const double *r_s, const double *r_n, double *grad,
                                 double *adj_pb /*[S]*/) {
  const int S = m->S, T = m->T, full = m->variant == POTUS_VARIANT_FULL;
  double *adj_e = (double *)xmalloc(sizeof(double) * (size_t)T);
  double adj_nat_pb = 0.0;
  for (int s = 0; s < S; s++) adj_pb[s] = 0.0;
  for (int i = 0; i < m->Ns; i++) {
    double r = r_s[i];
    grad[m->o_c + m->poll_state[i]] += m->sigma_c * r;
    if (full) {
      grad[m->o_m + m->mode_state[i]] += m->sigma_m * r;
      grad[m->o_pop + m->pop_state[i]] += m->sigma_pop * r;
      adj_e[m->day_state[i]] += m->unadj_state[i] * r;
    }
    grad[m->o_ns + i] += m->sigma_ns * r;
    adj_pb[m->state[i]] += r;
  }
  for (int j = 0; j < m->Nn; j++) {
    double r = r_n[j];
    grad[m->o_c + m->poll_nat[j]] += m->sigma_c * r;
    if (full) {
      grad[m->o_m + m->mode_nat[j]] += m->sigma_m * r;
      grad[m->o_pop + m->pop_nat[j]] += m->sigma_pop * r;
      adj_e[m->day_nat[j]] += m->unadj_nat[j] * r;
    }
    grad[m->o_nn + j] += m->sigma_nn * r;
    adj_nat_pb += r;
  }
  for (int s = 0; s < S; s++) adj_pb[s] += m->w[s] * adj_nat_pb; /* adjoint of stan:79 */
  double *gzb = (double *)xmalloc(sizeof(double) * (size_t)S);
  lower_tmv(m->LB, adj_pb, gzb, S);                                /* adjoint of stan:77 */
  for (int s = 0; s < S; s++) grad[m->o_zb + s] += gzb[s];
  free(gzb);
  if (full) { /* reverse sweep of the AR(1) recursion, stan:91-93 */
    const double *ze = q + m->o_ze;
    double adj_mue = 0.0, adj_rho = 0.0, adj_srho = 0.0;
    for (int u = T - 1; u >= 1; u--) {
      double a = adj_e[u];
      adj_e[u - 1] += t->rho * a;
      adj_mue += (1.0 - t->rho) * a;
      adj_rho += (t->e_bias[u - 1] - t->mu_e) * a;
      grad[m->o_ze + u] += t->sigma_rho * a;
      adj_srho += ze[u] * a;
    }
    grad[m->o_ze + 0] += m->sigma_e * adj_e[0];
    adj_rho += adj_srho * m->sigma_e * (-t->rho / sqrt(1.0 - t->rho * t->rho));
    grad[m->o_mue] += 0.02 * adj_mue;
    grad[m->o_rho] += adj_rho * t->rho * (1.0 - t->rho);
  }
  free(adj_e);
}

/* log_prob + gradient, literal: forward exactly as the Stan program, reverse sweep in the
   exact opposite order (what reverse-mode AD does to stan:85-87). */
double oracle_log_prob_grad(const oracle_model *m, const double *q, double *grad) {
  const int S = m->S, T = m->T;
  tparams t = tp_alloc(m);
  transformed_parameters(m, q, &t);
  for (int i = 0; i < m->D; i++) grad[i] = 0.0;
  double lp = priors_and_jacobian(m, q, &t, grad);
  double *r_s = (double *)xmalloc(sizeof(double) * (size_t)(m->Ns + 1));
  double *r_n = (double *)xmalloc(sizeof(double) * (size_t)(m->Nn + 1));
  lp += likelihood(m, &t, r_s, r_n);
  double *adj_pb = (double *)xmalloc(sizeof(double) * (size_t)S);
  small_block_adjoints(m, q, &t, r_s, r_n, grad, adj_pb);

  /* adjoint of mu_b: polls (stan:97,103 -> also feeds adj_pb identically) and stan:87 */
  double *adj_mu_b = (double *)xmalloc(sizeof(double) * (size_t)S * T);
  double *adj_nat = (double *)xmalloc(sizeof(double) * (size_t)T);
  for (int i = 0; i < m->Ns; i++) adj_mu_b[m->state[i] + (size_t)m->day_state[i] * S] += r_s[i];
  for (int j = 0; j < m->Nn; j++) adj_nat[m->day_nat[j]] += r_n[j];
  for (int u = 0; u < T; u++)
    for (int s = 0; s < S; s++) adj_mu_b[s + (size_t)u * S] += m->w[s] * adj_nat[u];
  /* reverse of the loop at stan:86 (it ran tt = T-2 .. 0, so the sweep runs tt = 0 .. T-2) */
  double *tmp = (double *)xmalloc(sizeof(double) * (size_t)S);
  for (int tt = 0; tt <= T - 2; tt++) {
    const double *a = adj_mu_b + (size_t)tt * S;
    lower_tmv(m->LW, a, tmp, S);
    for (int s = 0; s < S; s++) {
      grad[m->o_Z + s + (size_t)tt * S] += tmp[s];
      adj_mu_b[s + (size_t)(tt + 1) * S] += a[s];
    }
  }
  lower_tmv(m->LT, adj_mu_b + (size_t)(T - 1) * S, tmp, S); /* adjoint of stan:85 */
  for (int s = 0; s < S; s++) grad[m->o_zT + s] += tmp[s];
  free(tmp); free(adj_mu_b); free(adj_nat); free(adj_pb); free(r_s); free(r_n);
  tp_free(&t);
  return lp;
}

/* Same function through suffix/prefix scans over days and the sparsity of the polls:
 *   mu_b[:,t] = (L_T z_T + prior) + L_W * sum_{u=t}^{T-2} Z[:,u]        (closed form of stan:85-86)
 *   dZ[:,u]   = L_W^T * sum_{t<=u} G[:,t] - Z[:,u]   with G the poll residuals per (state,day)
 * evaluated only at polled cells.  Checked against the literal version in tests/. */
double oracle_log_prob_grad_fast(const oracle_model *m, const double *q, double *grad) {
  const int S = m->S, T = m->T, full = m->variant == POTUS_VARIANT_FULL;
  const double *zT = q + m->o_zT, *Z = q + m->o_Z, *zb = q + m->o_zb;
  tparams t = tp_alloc(m);
  double *C = (double *)xmalloc(sizeof(double) * (size_t)S * T); /* suffix sums, C[:,T-1] = 0 */
  double *bT = (double *)xmalloc(sizeof(double) * (size_t)S);
  double *v = (double *)xmalloc(sizeof(double) * (size_t)S);     /* L_W^T w */
  lower_mv(m->LT, zT, bT, S);
  for (int s = 0; s < S; s++) bT[s] += m->prior[s];
  lower_tmv(m->LW, m->w, v, S);
  for (int u = T - 2; u >= 0; u--)
    for (int k = 0; k < S; k++) C[k + (size_t)u * S] = C[k + (size_t)(u + 1) * S] + Z[k + (size_t)u * S];
  lower_mv(m->LB, zb, t.pb, S);
  t.nat_pb = 0.0; double wbT = 0.0;
  for (int s = 0; s < S; s++) { t.nat_pb += t.pb[s] * m->w[s]; wbT += bT[s] * m->w[s]; }
  for (int p = 0; p < m->P; p++) t.mu_c[p] = q[m->o_c + p] * m->sigma_c;
  if (full) {
    for (int i = 0; i < m->M; i++) t.mu_m[i] = q[m->o_m + i] * m->sigma_m;
    for (int i = 0; i < m->Pop; i++) t.mu_pop[i] = q[m->o_pop + i] * m->sigma_pop;
    t.mu_e = 0.02 * q[m->o_mue]; t.rho = inv_logit(q[m->o_rho]);
    const double *ze = q + m->o_ze;
    t.e_bias[0] = ze[0] * m->sigma_e;
    t.sigma_rho = sqrt(1.0 - t.rho * t.rho) * m->sigma_e;
    for (int u = 1; u < T; u++) t.e_bias[u] = t.mu_e + t.rho * (t.e_bias[u - 1] - t.mu_e) + ze[u] * t.sigma_rho;
  }
  for (int i = 0; i < m->Ns; i++) {
    int s = m->state[i], d = m->day_state[i];
    double mb = bT[s];
    for (int k = 0; k <= s; k++) mb += m->LW[s + (size_t)k * S] * C[k + (size_t)d * S];
    double e = mb + t.mu_c[m->poll_state[i]];
    if (full) e += t.mu_m[m->mode_state[i]] + t.mu_pop[m->pop_state[i]] + m->unadj_state[i] * t.e_bias[d];
    t.eta_s[i] = e + q[m->o_ns + i] * m->sigma_ns + t.pb[s];
  }
  for (int j = 0; j < m->Nn; j++) {
    int d = m->day_nat[j];
    double na = wbT;
    for (int k = 0; k < S; k++) na += v[k] * C[k + (size_t)d * S];
    double e = na + t.mu_c[m->poll_nat[j]];
    if (full) e += t.mu_m[m->mode_nat[j]] + t.mu_pop[m->pop_nat[j]] + m->unadj_nat[j] * t.e_bias[d];
    t.eta_n[j] = e + q[m->o_nn + j] * m->sigma_nn + t.nat_pb;
  }
  for (int i = 0; i < m->D; i++) grad[i] = 0.0;
  double lp = priors_and_jacobian(m, q, &t, grad);
  double *r_s = (double *)xmalloc(sizeof(double) * (size_t)(m->Ns + 1));
  double *r_n = (double *)xmalloc(sizeof(double) * (size_t)(m->Nn + 1));
  lp += likelihood(m, &t, r_s, r_n);
  double *adj_pb = (double *)xmalloc(sizeof(double) * (size_t)S);
  small_block_adjoints(m, q, &t, r_s, r_n, grad, adj_pb);
  /* adj_pb[s] = sum of residuals hitting state s (+ w_s * national total) = adjoint of bT too */
  double *tmp = (double *)xmalloc(sizeof(double) * (size_t)S);
  lower_tmv(m->LT, adj_pb, tmp, S);
  for (int s = 0; s < S; s++) grad[m->o_zT + s] += tmp[s];
  /* gC[:,t] = sum_{polls on day t} r_i * L_W[s_i, :]^T + v * (national residuals on day t);
     dZ[:,u] = prefix sum over t <= u */
  double *gnat = (double *)xmalloc(sizeof(double) * (size_t)T);
  for (int j = 0; j < m->Nn; j++) gnat[m->day_nat[j]] += r_n[j];
  double *run = (double *)xmalloc(sizeof(double) * (size_t)S);
  for (int u = 0; u <= T - 2; u++) {
    for (int a = m->day_ptr[u]; a < m->day_ptr[u + 1]; a++) {
      int i = m->day_idx[a], s = m->state[i];
      double r = r_s[i];
      for (int k = 0; k <= s; k++) run[k] += r * m->LW[s + (size_t)k * S];
    }
    if (gnat[u] != 0.0) for (int k = 0; k < S; k++) run[k] += v[k] * gnat[u];
    for (int k = 0; k < S; k++) grad[m->o_Z + k + (size_t)u * S] += run[k];
  }
  free(run); free(gnat); free(tmp); free(adj_pb); free(r_s); free(r_n); free(C); free(bT); free(v);
  tp_free(&t);
  return lp;
}

/* write_array: constrained parameters (stan:56-69), transformed parameters in declaration
   order (stan:72-83) and generated quantities (stan:134-140) */
void oracle_write_array(const oracle_model *m, const double *q, double *out) {
  const int S = m->S, T = m->T, full = m->variant == POTUS_VARIANT_FULL;
  tparams t = tp_alloc(m);
  transformed_parameters(m, q, &t);
  int o = 0;
  for (int i = 0; i < m->D; i++) out[o++] = q[i];
  if (full) { out[m->o_mue] = t.mu_e; out[m->o_rho] = t.rho; }
  for (int i = 0; i < S * T; i++) out[o++] = t.mu_b[i];     /* matrix[S,T] col-major */
  for (int i = 0; i < m->P; i++) out[o++] = t.mu_c[i];
  if (full) {
    for (int i = 0; i < m->M; i++) out[o++] = t.mu_m[i];
    for (int i = 0; i < m->Pop; i++) out[o++] = t.mu_pop[i];
    for (int i = 0; i < T; i++) out[o++] = t.e_bias[i];
  }
  for (int i = 0; i < S; i++) out[o++] = t.pb[i];
  for (int i = 0; i < T; i++) out[o++] = t.nat_avg[i];
  out[o++] = t.nat_pb;
  if (full) out[o++] = t.sigma_rho;
  for (int i = 0; i < m->Ns; i++) out[o++] = t.eta_s[i];
  for (int i = 0; i < m->Nn; i++) out[o++] = t.eta_n[i];
  for (int s = 0; s < S; s++)                                /* predicted_score[T,S] col-major, stan:135-139 */
    for (int u = 0; u < T; u++) out[o + u + (size_t)s * T] = inv_logit(t.mu_b[s + (size_t)u * S]);
  tp_free(&t);
}

/* ------------------------------------------------------------------------------------ */
/* RNG: Philox4x32-10 (Salmon et al. 2011), counter-based so that the device sampler and */
/* this oracle draw identical variates independent of execution order.                   */
/* Stan itself uses boost::ecuyer1988; parity with it is statistical, not bitwise.       */
/* ------------------------------------------------------------------------------------ */
void oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static double u53(uint32_t a, uint32_t b) { /* uniform on (0,1), 53 bits */
  uint64_t x = ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
  return ((double)x + 0.5) * (1.0 / 9007199254740992.0);
}
static void rng_block(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t purpose, uint32_t aux,
                      uint32_t index, uint32_t out[4]) {
  uint32_t ctr[4] = {index, purpose | (aux << 8), iter, chain};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  oracle_philox(ctr, key, out);
}
double oracle_rng_uniform(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t purpose, uint32_t aux, uint32_t index) {
  uint32_t o[4]; rng_block(seed, chain, iter, purpose, aux, index, o);
  return u53(o[0], o[1]);
}
void oracle_rng_normal_pair(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t purpose, uint32_t aux,
                            uint32_t index, double *n0, double *n1) {
  uint32_t o[4]; rng_block(seed, chain, iter, purpose, aux, index, o);
  double u1 = u53(o[0], o[1]), u2 = u53(o[2], o[3]);
  double r = sqrt(-2.0 * log(u1)), a = 6.283185307179586476925286766559 * u2;
  *n0 = r * cos(a); *n1 = r * sin(a);
}
enum { RNG_MOMENTUM = 0, RNG_DIRECTION = 1, RNG_TOP_ACCEPT = 2, RNG_SUB_ACCEPT = 3, RNG_INIT_EPS = 4, RNG_INITS = 5 };
#define ITER_PRE 0xFFFFFFFFu /* draws made before the first transition */

/* ------------------------------------------------------------------------------------ */
/* NUTS (Stan 2.24 base_nuts.hpp) with diag_e metric, recursive exactly like upstream   */
/* ------------------------------------------------------------------------------------ */
typedef struct { double *q, *p, *g; double V; } pspoint; /* g = dV/dq = -grad lp */

typedef struct {
  const oracle_model *m;
  const oracle_opts *o;
  int D, chain;
  uint32_t iter;
  double (*lpg)(const oracle_model *, const double *, double *);
  double *minv;    /* inverse metric diagonal (dense: its diagonal, kept for reporting) */
  int dense;       /* dense_e_metric: Minv (D x D, row-major, symmetric), Lc = its lower Cholesky factor */
  double *Minv, *Lc, *wf_M2d, *tmpv;
  double eps;      /* epsilon_ used by the current transition */
  double nom_eps;
  pspoint z;
  int depth, n_leapfrog, divergent;
  double energy;
  long long total_leapfrogs;
  /* adaptation */
  double mu, s_bar, x_bar, ad_counter;            /* stepsize_adaptation.hpp */
  int win_counter, win_next, win_size;            /* windowed_adaptation.hpp */
  int nw, ib, tb, bw;
  double *wf_mean, *wf_m2; double wf_n;           /* welford_var_estimator.hpp */
  int top_depth;                                  /* depth of the doubling being built (RNG slot) */
} sampler;

static pspoint ps_alloc(int D) {
  pspoint z; z.q = (double *)xmalloc(sizeof(double) * (size_t)D); z.p = (double *)xmalloc(sizeof(double) * (size_t)D);
  z.g = (double *)xmalloc(sizeof(double) * (size_t)D); z.V = 0; return z;
}
static void ps_free(pspoint *z) { free(z->q); free(z->p); free(z->g); }
static void ps_copy(pspoint *d, const pspoint *s, int D) {
  memcpy(d->q, s->q, sizeof(double) * (size_t)D); memcpy(d->p, s->p, sizeof(double) * (size_t)D);
  memcpy(d->g, s->g, sizeof(double) * (size_t)D); d->V = s->V;
}
static double *vec(int D) { return (double *)xmalloc(sizeof(double) * (size_t)D); }

static void update_potential_gradient(sampler *sp, pspoint *z) {
  double lp = sp->lpg(sp->m, z->q, z->g);
  z->V = -lp;
  for (int i = 0; i < sp->D; i++) z->g[i] = -z->g[i];
}
/* M^-1 p: diag_e_metric / dense_e_metric::dtau_dp */
static void dtau_dp(const sampler *sp, const double *p, double *out) {
  const int D = sp->D;
  if (!sp->dense) { for (int i = 0; i < D; i++) out[i] = sp->minv[i] * p[i]; return; }
  /* (rows are independent and each is summed in index order: the threads change nothing in the result) */
#pragma omp parallel for schedule(static) if (D >= 2048)
  for (int i = 0; i < D; i++) {
    const double *row = sp->Minv + (size_t)i * D;
    double s = 0.0; for (int j = 0; j < D; j++) s += row[j] * p[j];
    out[i] = s;
  }
}
static double kinetic(const sampler *sp, const pspoint *z) { /* diag_e_metric::T / dense_e_metric::T = 0.5 p' M^-1 p */
  double s = 0.0;
  if (!sp->dense) { for (int i = 0; i < sp->D; i++) s += sp->minv[i] * z->p[i] * z->p[i]; return 0.5 * s; }
  dtau_dp(sp, z->p, sp->tmpv);
  for (int i = 0; i < sp->D; i++) s += z->p[i] * sp->tmpv[i];
  return 0.5 * s;
}
static double hamiltonian(const sampler *sp, const pspoint *z) { return kinetic(sp, z) + z->V; }
static void sample_p(sampler *sp, pspoint *z, uint32_t purpose, uint32_t aux) { /* diag_e_metric / dense_e_metric::sample_p */
  const int D = sp->D;
  for (int j = 0; 2 * j < D; j++) {
    double a, b; oracle_rng_normal_pair(sp->o->seed, (uint32_t)sp->chain, sp->iter, purpose, aux, (uint32_t)j, &a, &b);
    z->p[2 * j] = a;
    if (2 * j + 1 < D) z->p[2 * j + 1] = b;
  }
  if (!sp->dense) { for (int i = 0; i < D; i++) z->p[i] /= sqrt(sp->minv[i]); return; }
  /* p = inv_e_metric.llt().matrixU().solve(u): back substitution with U = Lc' */
  for (int i = D - 1; i >= 0; i--) {
    double s = z->p[i];
    for (int j = i + 1; j < D; j++) s -= sp->Lc[(size_t)j * D + i] * z->p[j];
    z->p[i] = s / sp->Lc[(size_t)i * D + i];
  }
}
/* expl_leapfrog::evolve (begin_update_p, update_q, end_update_p) */
static void evolve(sampler *sp, pspoint *z, double eps) {
  for (int i = 0; i < sp->D; i++) z->p[i] -= 0.5 * eps * z->g[i];
  if (!sp->dense) for (int i = 0; i < sp->D; i++) z->q[i] += eps * sp->minv[i] * z->p[i];
  else { dtau_dp(sp, z->p, sp->tmpv); for (int i = 0; i < sp->D; i++) z->q[i] += eps * sp->tmpv[i]; }
  update_potential_gradient(sp, z);
  for (int i = 0; i < sp->D; i++) z->p[i] -= 0.5 * eps * z->g[i];
}
static double log_sum_exp(double a, double b) {
  if (a == -INFINITY) return b;
  if (a == INFINITY && b == INFINITY) return INFINITY;
  return a > b ? a + log1p(exp(b - a)) : b + log1p(exp(a - b));
}
/* base_nuts::compute_criterion */
static int criterion(const sampler *sp, const double *psm, const double *psp, const double *rho) {
  double a = 0, b = 0; for (int i = 0; i < sp->D; i++) { a += psp[i] * rho[i]; b += psm[i] * rho[i]; }
  return a > 0 && b > 0;
}

/* base_nuts::build_tree.  `node` is the index of this subtree among the subtrees of its level
   inside the current doubling; it only selects the RNG slot of the multinomial draw. */
static int build_tree(sampler *sp, int depth, int node, pspoint *z_propose, double *psharp_beg, double *psharp_end,
                      double *rho, double *p_beg, double *p_end, double H0, double sign, int *n_leapfrog,
                      double *log_sum_weight, double *sum_metro_prob) {
  const int D = sp->D;
  if (depth == 0) {
    evolve(sp, &sp->z, sign * sp->eps);
    ++*n_leapfrog;
    double h = hamiltonian(sp, &sp->z);
    if (isnan(h)) h = INFINITY;
    if ((h - H0) > 1000.0) sp->divergent = 1;
    *log_sum_weight = log_sum_exp(*log_sum_weight, H0 - h);
    if (H0 - h > 0) *sum_metro_prob += 1; else *sum_metro_prob += exp(H0 - h);
    ps_copy(z_propose, &sp->z, D);
    dtau_dp(sp, sp->z.p, psharp_beg);
    memcpy(psharp_end, psharp_beg, sizeof(double) * (size_t)D);
    for (int i = 0; i < D; i++) rho[i] += sp->z.p[i];
    memcpy(p_beg, sp->z.p, sizeof(double) * (size_t)D);
    memcpy(p_end, p_beg, sizeof(double) * (size_t)D);
    return !sp->divergent;
  }
  double lsw_init = -INFINITY;
  double *p_init_end = vec(D), *psharp_init_end = vec(D), *rho_init = vec(D);
  int ok = build_tree(sp, depth - 1, 2 * node, z_propose, psharp_beg, psharp_init_end, rho_init, p_beg, p_init_end, H0,
                      sign, n_leapfrog, &lsw_init, sum_metro_prob);
  if (!ok) { free(p_init_end); free(psharp_init_end); free(rho_init); return 0; }
  pspoint z_propose_final = ps_alloc(D); ps_copy(&z_propose_final, &sp->z, D);
  double lsw_final = -INFINITY;
  double *p_final_beg = vec(D), *psharp_final_beg = vec(D), *rho_final = vec(D);
  ok = build_tree(sp, depth - 1, 2 * node + 1, &z_propose_final, psharp_final_beg, psharp_end, rho_final, p_final_beg, p_end,
                  H0, sign, n_leapfrog, &lsw_final, sum_metro_prob);
  int persist = 0;
  if (ok) {
    double lsw_subtree = log_sum_exp(lsw_init, lsw_final);
    *log_sum_weight = log_sum_exp(*log_sum_weight, lsw_subtree);
    if (lsw_final > lsw_subtree) ps_copy(z_propose, &z_propose_final, D);
    else {
      double accept = exp(lsw_final - lsw_subtree);
      uint32_t slot = ((uint32_t)sp->top_depth << 24) | ((uint32_t)depth << 16) | (uint32_t)node;
      double u = oracle_rng_uniform(sp->o->seed, (uint32_t)sp->chain, sp->iter, RNG_SUB_ACCEPT, 0, slot);
      if (u < accept) ps_copy(z_propose, &z_propose_final, D);
    }
    double *rho_subtree = vec(D), *rho_ext = vec(D);
    for (int i = 0; i < D; i++) { rho_subtree[i] = rho_init[i] + rho_final[i]; rho[i] += rho_subtree[i]; }
    persist = criterion(sp, psharp_beg, psharp_end, rho_subtree);
    for (int i = 0; i < D; i++) rho_ext[i] = rho_init[i] + p_final_beg[i];
    persist &= criterion(sp, psharp_beg, psharp_final_beg, rho_ext);
    for (int i = 0; i < D; i++) rho_ext[i] = rho_final[i] + p_init_end[i];
    persist &= criterion(sp, psharp_init_end, psharp_end, rho_ext);
    free(rho_subtree); free(rho_ext);
  }
  ps_free(&z_propose_final);
  free(p_init_end); free(psharp_init_end); free(rho_init); free(p_final_beg); free(psharp_final_beg); free(rho_final);
  return ok ? persist : 0;
}

/* base_nuts::transition; returns accept_stat, leaves the new sample in sp->z */
static double nuts_transition(sampler *sp) {
  const int D = sp->D;
  sp->eps = sp->nom_eps; /* sample_stepsize(), jitter = 0 */
  sample_p(sp, &sp->z, RNG_MOMENTUM, 0);
  update_potential_gradient(sp, &sp->z); /* hamiltonian.init */
  pspoint z_fwd = ps_alloc(D), z_bck = ps_alloc(D), z_sample = ps_alloc(D), z_propose = ps_alloc(D);
  ps_copy(&z_fwd, &sp->z, D); ps_copy(&z_bck, &sp->z, D); ps_copy(&z_sample, &sp->z, D); ps_copy(&z_propose, &sp->z, D);
  /* the trajectory is [bck subtree][fwd subtree]; each has a forward and a backward end */
  double *p_fwd_fwd = vec(D), *ps_fwd_fwd = vec(D), *p_fwd_bck = vec(D), *ps_fwd_bck = vec(D);
  double *p_bck_fwd = vec(D), *ps_bck_fwd = vec(D), *p_bck_bck = vec(D), *ps_bck_bck = vec(D);
  double *rho = vec(D), *rho_fwd = vec(D), *rho_bck = vec(D), *rho_ext = vec(D);
  const size_t nb = sizeof(double) * (size_t)D;
  memcpy(p_fwd_fwd, sp->z.p, nb); dtau_dp(sp, sp->z.p, ps_fwd_fwd);
  memcpy(p_fwd_bck, p_fwd_fwd, nb); memcpy(ps_fwd_bck, ps_fwd_fwd, nb);
  memcpy(p_bck_fwd, p_fwd_fwd, nb); memcpy(ps_bck_fwd, ps_fwd_fwd, nb);
  memcpy(p_bck_bck, p_fwd_fwd, nb); memcpy(ps_bck_bck, ps_fwd_fwd, nb);
  memcpy(rho, sp->z.p, nb);
  double log_sum_weight = 0.0, H0 = hamiltonian(sp, &sp->z);
  int n_leapfrog = 0; double sum_metro_prob = 0.0;
  sp->depth = 0; sp->divergent = 0;
  while (sp->depth < sp->o->max_depth) {
    memset(rho_fwd, 0, nb); memset(rho_bck, 0, nb);
    int valid; double lsw_subtree = -INFINITY;
    sp->top_depth = sp->depth;
    double udir = oracle_rng_uniform(sp->o->seed, (uint32_t)sp->chain, sp->iter, RNG_DIRECTION, 0, (uint32_t)sp->depth);
    if (udir > 0.5) { /* extend the current trajectory forward */
      ps_copy(&sp->z, &z_fwd, D);
      memcpy(rho_bck, rho, nb);
      memcpy(p_bck_fwd, p_fwd_fwd, nb);
      memcpy(ps_bck_fwd, ps_fwd_fwd, nb);
      valid = build_tree(sp, sp->depth, 0, &z_propose, ps_fwd_bck, ps_fwd_fwd, rho_fwd, p_fwd_bck, p_fwd_fwd, H0, 1.0,
                         &n_leapfrog, &lsw_subtree, &sum_metro_prob);
      ps_copy(&z_fwd, &sp->z, D);
    } else { /* extend the current trajectory backwards */
      ps_copy(&sp->z, &z_bck, D);
      memcpy(rho_fwd, rho, nb);
      memcpy(p_fwd_bck, p_bck_bck, nb);
      memcpy(ps_fwd_bck, ps_bck_bck, nb);
      valid = build_tree(sp, sp->depth, 0, &z_propose, ps_bck_fwd, ps_bck_bck, rho_bck, p_bck_fwd, p_bck_bck, H0, -1.0,
                         &n_leapfrog, &lsw_subtree, &sum_metro_prob);
      ps_copy(&z_bck, &sp->z, D);
    }
    if (!valid) break;
    ++sp->depth;
    if (lsw_subtree > log_sum_weight) ps_copy(&z_sample, &z_propose, D);
    else {
      double accept = exp(lsw_subtree - log_sum_weight);
      double u = oracle_rng_uniform(sp->o->seed, (uint32_t)sp->chain, sp->iter, RNG_TOP_ACCEPT, 0, (uint32_t)(sp->depth - 1));
      if (u < accept) ps_copy(&z_sample, &z_propose, D);
    }
    log_sum_weight = log_sum_exp(log_sum_weight, lsw_subtree);
    for (int i = 0; i < D; i++) rho[i] = rho_bck[i] + rho_fwd[i];
    int persist = criterion(sp, ps_bck_bck, ps_fwd_fwd, rho);            /* around the merged subtrees */
    for (int i = 0; i < D; i++) rho_ext[i] = rho_bck[i] + p_fwd_bck[i];  /* between the subtrees */
    persist &= criterion(sp, ps_bck_bck, ps_fwd_bck, rho_ext);
    for (int i = 0; i < D; i++) rho_ext[i] = rho_fwd[i] + p_bck_fwd[i];
    persist &= criterion(sp, ps_bck_fwd, ps_fwd_fwd, rho_ext);
    if (!persist) break;
  }
  sp->n_leapfrog = n_leapfrog;
  sp->total_leapfrogs += n_leapfrog;
  double accept_stat = sum_metro_prob / (double)n_leapfrog;
  ps_copy(&sp->z, &z_sample, D);
  sp->energy = hamiltonian(sp, &sp->z);
  ps_free(&z_fwd); ps_free(&z_bck); ps_free(&z_sample); ps_free(&z_propose);
  free(p_fwd_fwd); free(ps_fwd_fwd); free(p_fwd_bck); free(ps_fwd_bck); free(p_bck_fwd); free(ps_bck_fwd);
  free(p_bck_bck); free(ps_bck_bck); free(rho); free(rho_fwd); free(rho_bck); free(rho_ext);
  return accept_stat;
}

/* ------------------------------------------------------------------------------------ */
/* The same transition with pooled buffers and fused loops (oracle_opts.pooled, diagonal   */
/* metric): what bench.py's cpu_baseline times.  build_tree above allocates nine vectors    */
/* and copies three per tree node and walks every vector once per operation, as upstream    */
/* does; here every level of the recursion owns its vectors for the whole run and the       */
/* elementwise operations of a leaf / of a merge share one loop each.  Every sum runs over  */
/* the same terms in the same order, so the draws are the same bits                         */
/* (tests/test_oracle.py::test_pooled_tree_gives_the_same_draws).                           */
/* ------------------------------------------------------------------------------------ */
typedef struct { double *p_init_end, *ps_init_end, *rho_init, *p_final_beg, *ps_final_beg, *rho_final, *q_final; } tree_ws;
typedef struct {
  tree_ws lev[16];
  double *q_fwd, *p_fwd, *g_fwd, *q_bck, *p_bck, *g_bck, *q_sample, *q_propose;
  double *p_fwd_fwd, *ps_fwd_fwd, *p_fwd_bck, *ps_fwd_bck, *p_bck_fwd, *ps_bck_fwd, *p_bck_bck, *ps_bck_bck, *rho, *rho_fwd, *rho_bck;
  double V_fwd, V_bck;
  int ready;
} tree_pool;
static tree_pool *pool_make(int D, int max_depth) {
  tree_pool *tp = (tree_pool *)xmalloc(sizeof(tree_pool));
  for (int d = 0; d <= max_depth && d < 16; d++) {
    tree_ws *w = &tp->lev[d];
    w->p_init_end = vec(D); w->ps_init_end = vec(D); w->rho_init = vec(D); w->p_final_beg = vec(D); w->ps_final_beg = vec(D);
    w->rho_final = vec(D); w->q_final = vec(D);
  }
  double **all[] = {&tp->q_fwd, &tp->p_fwd, &tp->g_fwd, &tp->q_bck, &tp->p_bck, &tp->g_bck, &tp->q_sample, &tp->q_propose, &tp->p_fwd_fwd, &tp->ps_fwd_fwd,
                    &tp->p_fwd_bck, &tp->ps_fwd_bck, &tp->p_bck_fwd, &tp->ps_bck_fwd, &tp->p_bck_bck, &tp->ps_bck_bck, &tp->rho, &tp->rho_fwd, &tp->rho_bck};
  for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); k++) *all[k] = vec(D);
  tp->ready = 1;
  return tp;
}
static void pool_free_all(tree_pool *tp, int max_depth) {
  if (!tp) return;
  for (int d = 0; d <= max_depth && d < 16; d++) {
    tree_ws *w = &tp->lev[d];
    free(w->p_init_end); free(w->ps_init_end); free(w->rho_init); free(w->p_final_beg); free(w->ps_final_beg); free(w->rho_final); free(w->q_final);
  }
  double *all[] = {tp->q_fwd, tp->p_fwd, tp->g_fwd, tp->q_bck, tp->p_bck, tp->g_bck, tp->q_sample, tp->q_propose, tp->p_fwd_fwd, tp->ps_fwd_fwd,
                   tp->p_fwd_bck, tp->ps_fwd_bck, tp->p_bck_fwd, tp->ps_bck_fwd, tp->p_bck_bck, tp->ps_bck_bck, tp->rho, tp->rho_fwd, tp->rho_bck};
  for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); k++) free(all[k]);
  free(tp);
}
/* a proposal is a position and its potential: the momentum and gradient of the sample are drawn / evaluated afresh by the
   next transition (hamiltonian.sample, hamiltonian.init), so upstream's copies of them are never read */
static int build_tree_pooled(sampler *sp, tree_pool *tp, int depth, int node, double *q_propose, double *VH_propose /* potential, Hamiltonian */, double *psharp_beg,
                             double *psharp_end, double *rho, double *p_beg, double *p_end, double H0, double sign, int *n_leapfrog,
                             double *log_sum_weight, double *sum_metro_prob) {
  const int D = sp->D;
  if (depth == 0) {
    const double eps = sign * sp->eps, *minv = sp->minv;
    double *q = sp->z.q, *p = sp->z.p, *g = sp->z.g;
    for (int i = 0; i < D; i++) { p[i] -= 0.5 * eps * g[i]; q[i] += eps * minv[i] * p[i]; }   /* begin_update_p, update_q */
    update_potential_gradient(sp, &sp->z);
    double kin = 0.0;
    for (int i = 0; i < D; i++) {                                                              /* end_update_p and the leaf's bookkeeping */
      p[i] -= 0.5 * eps * g[i];
      const double pi = p[i], ps = minv[i] * pi;
      kin += minv[i] * pi * pi;
      psharp_beg[i] = ps; psharp_end[i] = ps; rho[i] += pi; p_beg[i] = pi; p_end[i] = pi; q_propose[i] = q[i];
    }
    ++*n_leapfrog;
    double h = 0.5 * kin + sp->z.V;
    if (isnan(h)) h = INFINITY;
    if ((h - H0) > 1000.0) sp->divergent = 1;
    *log_sum_weight = log_sum_exp(*log_sum_weight, H0 - h);
    if (H0 - h > 0) *sum_metro_prob += 1; else *sum_metro_prob += exp(H0 - h);
    VH_propose[0] = sp->z.V; VH_propose[1] = 0.5 * kin + sp->z.V;
    return !sp->divergent;
  }
  tree_ws *w = &tp->lev[depth];
  double lsw_init = -INFINITY;
  memset(w->rho_init, 0, sizeof(double) * (size_t)D);
  int ok = build_tree_pooled(sp, tp, depth - 1, 2 * node, q_propose, VH_propose, psharp_beg, w->ps_init_end, w->rho_init, p_beg, w->p_init_end, H0,
                             sign, n_leapfrog, &lsw_init, sum_metro_prob);
  if (!ok) return 0;
  double lsw_final = -INFINITY, VH_final[2] = {sp->z.V, 0.0};
  memset(w->rho_final, 0, sizeof(double) * (size_t)D);
  ok = build_tree_pooled(sp, tp, depth - 1, 2 * node + 1, w->q_final, VH_final, w->ps_final_beg, psharp_end, w->rho_final, w->p_final_beg, p_end,
                         H0, sign, n_leapfrog, &lsw_final, sum_metro_prob);
  if (!ok) return 0;
  double lsw_subtree = log_sum_exp(lsw_init, lsw_final);
  *log_sum_weight = log_sum_exp(*log_sum_weight, lsw_subtree);
  int take = 0;
  if (lsw_final > lsw_subtree) take = 1;
  else {
    double accept = exp(lsw_final - lsw_subtree);
    uint32_t slot = ((uint32_t)sp->top_depth << 24) | ((uint32_t)depth << 16) | (uint32_t)node;
    double u = oracle_rng_uniform(sp->o->seed, (uint32_t)sp->chain, sp->iter, RNG_SUB_ACCEPT, 0, slot);
    if (u < accept) take = 1;
  }
  if (take) { memcpy(q_propose, w->q_final, sizeof(double) * (size_t)D); VH_propose[0] = VH_final[0]; VH_propose[1] = VH_final[1]; }
  /* the three compute_criterion calls in one loop: a_k = p#_end-like . rho_k, b_k = p#_beg-like . rho_k */
  double a0 = 0, b0 = 0, a1 = 0, b1 = 0, a2 = 0, b2 = 0;
  const double *ri = w->rho_init, *rf = w->rho_final, *pfb = w->p_final_beg, *pie = w->p_init_end, *sfb = w->ps_final_beg, *sie = w->ps_init_end;
  for (int i = 0; i < D; i++) {
    const double rs = ri[i] + rf[i];
    rho[i] += rs;
    a0 += psharp_end[i] * rs; b0 += psharp_beg[i] * rs;
    const double e1 = ri[i] + pfb[i];
    a1 += sfb[i] * e1; b1 += psharp_beg[i] * e1;
    const double e2 = rf[i] + pie[i];
    a2 += psharp_end[i] * e2; b2 += sie[i] * e2;
  }
  return (a0 > 0 && b0 > 0) & (a1 > 0 && b1 > 0) & (a2 > 0 && b2 > 0);
}
static double nuts_transition_pooled(sampler *sp, tree_pool *tp) {
  const int D = sp->D;
  const size_t nb = sizeof(double) * (size_t)D;
  sp->eps = sp->nom_eps;
  sample_p(sp, &sp->z, RNG_MOMENTUM, 0);
  update_potential_gradient(sp, &sp->z);
  memcpy(tp->q_fwd, sp->z.q, nb); memcpy(tp->p_fwd, sp->z.p, nb); memcpy(tp->g_fwd, sp->z.g, nb); tp->V_fwd = sp->z.V;
  memcpy(tp->q_bck, sp->z.q, nb); memcpy(tp->p_bck, sp->z.p, nb); memcpy(tp->g_bck, sp->z.g, nb); tp->V_bck = sp->z.V;
  memcpy(tp->q_sample, sp->z.q, nb);
  memcpy(tp->q_propose, sp->z.q, nb);
  double kin0 = 0.0;
  for (int i = 0; i < D; i++) {
    const double pi = sp->z.p[i], ps = sp->minv[i] * pi;
    kin0 += sp->minv[i] * pi * pi;
    tp->p_fwd_fwd[i] = pi; tp->p_fwd_bck[i] = pi; tp->p_bck_fwd[i] = pi; tp->p_bck_bck[i] = pi; tp->rho[i] = pi;
    tp->ps_fwd_fwd[i] = ps; tp->ps_fwd_bck[i] = ps; tp->ps_bck_fwd[i] = ps; tp->ps_bck_bck[i] = ps;
  }
  double log_sum_weight = 0.0, H0 = 0.5 * kin0 + sp->z.V;
  double VH_sample[2] = {sp->z.V, H0}, VH_propose[2] = {sp->z.V, H0};
  int n_leapfrog = 0; double sum_metro_prob = 0.0;
  sp->depth = 0; sp->divergent = 0;
  while (sp->depth < sp->o->max_depth) {
    int valid; double lsw_subtree = -INFINITY;
    sp->top_depth = sp->depth;
    double udir = oracle_rng_uniform(sp->o->seed, (uint32_t)sp->chain, sp->iter, RNG_DIRECTION, 0, (uint32_t)sp->depth);
    if (udir > 0.5) {
      memcpy(sp->z.q, tp->q_fwd, nb); memcpy(sp->z.p, tp->p_fwd, nb); memcpy(sp->z.g, tp->g_fwd, nb); sp->z.V = tp->V_fwd;
      memcpy(tp->rho_bck, tp->rho, nb); memset(tp->rho_fwd, 0, nb);
      memcpy(tp->p_bck_fwd, tp->p_fwd_fwd, nb); memcpy(tp->ps_bck_fwd, tp->ps_fwd_fwd, nb);
      valid = build_tree_pooled(sp, tp, sp->depth, 0, tp->q_propose, VH_propose, tp->ps_fwd_bck, tp->ps_fwd_fwd, tp->rho_fwd, tp->p_fwd_bck, tp->p_fwd_fwd,
                                H0, 1.0, &n_leapfrog, &lsw_subtree, &sum_metro_prob);
      memcpy(tp->q_fwd, sp->z.q, nb); memcpy(tp->p_fwd, sp->z.p, nb); memcpy(tp->g_fwd, sp->z.g, nb); tp->V_fwd = sp->z.V;
    } else {
      memcpy(sp->z.q, tp->q_bck, nb); memcpy(sp->z.p, tp->p_bck, nb); memcpy(sp->z.g, tp->g_bck, nb); sp->z.V = tp->V_bck;
      memcpy(tp->rho_fwd, tp->rho, nb); memset(tp->rho_bck, 0, nb);
      memcpy(tp->p_fwd_bck, tp->p_bck_bck, nb); memcpy(tp->ps_fwd_bck, tp->ps_bck_bck, nb);
      valid = build_tree_pooled(sp, tp, sp->depth, 0, tp->q_propose, VH_propose, tp->ps_bck_fwd, tp->ps_bck_bck, tp->rho_bck, tp->p_bck_fwd, tp->p_bck_bck,
                                H0, -1.0, &n_leapfrog, &lsw_subtree, &sum_metro_prob);
      memcpy(tp->q_bck, sp->z.q, nb); memcpy(tp->p_bck, sp->z.p, nb); memcpy(tp->g_bck, sp->z.g, nb); tp->V_bck = sp->z.V;
    }
    if (!valid) break;
    ++sp->depth;
    int take = 0;
    if (lsw_subtree > log_sum_weight) take = 1;
    else {
      double accept = exp(lsw_subtree - log_sum_weight);
      double u = oracle_rng_uniform(sp->o->seed, (uint32_t)sp->chain, sp->iter, RNG_TOP_ACCEPT, 0, (uint32_t)(sp->depth - 1));
      if (u < accept) take = 1;
    }
    if (take) { memcpy(tp->q_sample, tp->q_propose, nb); VH_sample[0] = VH_propose[0]; VH_sample[1] = VH_propose[1]; }
    log_sum_weight = log_sum_exp(log_sum_weight, lsw_subtree);
    double a0 = 0, b0 = 0, a1 = 0, b1 = 0, a2 = 0, b2 = 0;
    for (int i = 0; i < D; i++) {
      const double r = tp->rho_bck[i] + tp->rho_fwd[i];
      tp->rho[i] = r;
      a0 += tp->ps_fwd_fwd[i] * r; b0 += tp->ps_bck_bck[i] * r;
      const double e1 = tp->rho_bck[i] + tp->p_fwd_bck[i];
      a1 += tp->ps_fwd_bck[i] * e1; b1 += tp->ps_bck_bck[i] * e1;
      const double e2 = tp->rho_fwd[i] + tp->p_bck_fwd[i];
      a2 += tp->ps_fwd_fwd[i] * e2; b2 += tp->ps_bck_fwd[i] * e2;
    }
    if (!((a0 > 0 && b0 > 0) && (a1 > 0 && b1 > 0) && (a2 > 0 && b2 > 0))) break;
  }
  sp->n_leapfrog = n_leapfrog;
  sp->total_leapfrogs += n_leapfrog;
  double accept_stat = sum_metro_prob / (double)n_leapfrog;
  /* the new sample: position, potential and Hamiltonian (energy__: upstream evaluates H at the copied sample point, the same
     expression the leaf evaluated) */
  memcpy(sp->z.q, tp->q_sample, nb); sp->z.V = VH_sample[0]; sp->energy = VH_sample[1];
  return accept_stat;
}

/* base_hmc::init_stepsize */
static void init_stepsize(sampler *sp) {
  const int D = sp->D;
  if (sp->nom_eps == 0 || sp->nom_eps > 1e7 || isnan(sp->nom_eps)) return;
  pspoint z_init = ps_alloc(D); ps_copy(&z_init, &sp->z, D);
  uint32_t attempt = 0;
  sample_p(sp, &sp->z, RNG_INIT_EPS, attempt++);
  update_potential_gradient(sp, &sp->z);
  double H0 = hamiltonian(sp, &sp->z);
  evolve(sp, &sp->z, sp->nom_eps);
  double h = hamiltonian(sp, &sp->z); if (isnan(h)) h = INFINITY;
  double delta_H = H0 - h;
  int direction = delta_H > log(0.8) ? 1 : -1;
  while (1) {
    ps_copy(&sp->z, &z_init, D);
    sample_p(sp, &sp->z, RNG_INIT_EPS, attempt++);
    update_potential_gradient(sp, &sp->z);
    H0 = hamiltonian(sp, &sp->z);
    evolve(sp, &sp->z, sp->nom_eps);
    h = hamiltonian(sp, &sp->z); if (isnan(h)) h = INFINITY;
    delta_H = H0 - h;
    if (direction == 1 && !(delta_H > log(0.8))) break;
    else if (direction == -1 && !(delta_H < log(0.8))) break;
    else sp->nom_eps = direction == 1 ? 2 * sp->nom_eps : 0.5 * sp->nom_eps;
    if (sp->nom_eps > 1e7 || sp->nom_eps == 0) break; /* upstream throws */
  }
  ps_copy(&sp->z, &z_init, D);
  ps_free(&z_init);
}

/* stepsize_adaptation::learn_stepsize */
static void learn_stepsize(sampler *sp, double adapt_stat) {
  sp->ad_counter += 1;
  adapt_stat = adapt_stat > 1 ? 1 : adapt_stat;
  double eta = 1.0 / (sp->ad_counter + sp->o->t0);
  sp->s_bar = (1.0 - eta) * sp->s_bar + eta * (sp->o->delta - adapt_stat);
  double x = sp->mu - sp->s_bar * sqrt(sp->ad_counter) / sp->o->gamma;
  double x_eta = pow(sp->ad_counter, -sp->o->kappa);
  sp->x_bar = (1.0 - x_eta) * sp->x_bar + x_eta * x;
  sp->nom_eps = exp(x);
}
/* windowed_adaptation::compute_next_window */
static void compute_next_window(sampler *sp) {
  if (sp->win_next == sp->nw - sp->tb - 1) return;
  sp->win_size *= 2;
  sp->win_next = sp->win_counter + sp->win_size;
  if (sp->win_next == sp->nw - sp->tb - 1) return;
  int next_boundary = sp->win_next + 2 * sp->win_size;
  if (next_boundary >= sp->nw - sp->tb) sp->win_next = sp->nw - sp->tb - 1;
}
/* var_adaptation::learn_variance */
static int learn_variance(sampler *sp, const double *q) {
  const int D = sp->D;
  int in_window = sp->win_counter >= sp->ib && sp->win_counter < sp->nw - sp->tb && sp->win_counter != sp->nw;
  if (in_window && !sp->dense) { /* welford_var_estimator::add_sample */
    sp->wf_n += 1;
    for (int i = 0; i < D; i++) {
      double delta = q[i] - sp->wf_mean[i];
      sp->wf_mean[i] += delta / sp->wf_n;
      sp->wf_m2[i] += (q[i] - sp->wf_mean[i]) * delta;
    }
  }
  if (in_window && sp->dense) { /* welford_covar_estimator::add_sample: m2 += (q - m_new) * delta' */
    sp->wf_n += 1;
    double *delta = sp->tmpv;
    for (int i = 0; i < D; i++) { delta[i] = q[i] - sp->wf_mean[i]; sp->wf_mean[i] += delta[i] / sp->wf_n; }
    for (int i = 0; i < D; i++) {
      const double di = q[i] - sp->wf_mean[i];
      double *row = sp->wf_M2d + (size_t)i * D;
      for (int j = 0; j < D; j++) row[j] += di * delta[j];
    }
  }
  int end_window = sp->win_counter == sp->win_next && sp->win_counter != sp->nw;
  if (end_window) {
    compute_next_window(sp);
    double n = sp->wf_n;
    if (!sp->dense) {
      for (int i = 0; i < D; i++) {
        double var = sp->wf_m2[i] / (n - 1.0);
        sp->minv[i] = (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0));
      }
    } else { /* covar_adaptation::learn_covariance: covar = n/(n+5) * m2/(n-1) + 1e-3 * 5/(n+5) * I */
      for (int i = 0; i < D; i++)
        for (int j = 0; j < D; j++)
          sp->Minv[(size_t)i * D + j] = (n / (n + 5.0)) * (sp->wf_M2d[(size_t)i * D + j] / (n - 1.0)) + (i == j ? 1e-3 * (5.0 / (n + 5.0)) : 0.0);
      for (int i = 0; i < D; i++)      /* the estimator's m2 is symmetric up to rounding; the metric must be exactly so */
        for (int j = 0; j < i; j++) { const double a2 = 0.5 * (sp->Minv[(size_t)i * D + j] + sp->Minv[(size_t)j * D + i]); sp->Minv[(size_t)i * D + j] = a2; sp->Minv[(size_t)j * D + i] = a2; }
      for (int i = 0; i < D; i++) sp->minv[i] = sp->Minv[(size_t)i * D + i];
      /* column-major view of a symmetric matrix = the matrix: reuse cholesky_lower (column-major in / out), then transpose to row-major */
      double *Lcol = (double *)xmalloc(sizeof(double) * (size_t)D * D);
      if (cholesky_lower(sp->Minv, Lcol, D) != 0) { free(Lcol); return -1; }
      for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) sp->Lc[(size_t)i * D + j] = Lcol[i + (size_t)j * D];
      free(Lcol);
      memset(sp->wf_M2d, 0, sizeof(double) * (size_t)D * D);
    }
    sp->wf_n = 0; memset(sp->wf_mean, 0, sizeof(double) * (size_t)D); memset(sp->wf_m2, 0, sizeof(double) * (size_t)D);
    ++sp->win_counter;
    return 1;
  }
  ++sp->win_counter;
  return 0;
}

void oracle_default_opts(oracle_opts *o) {
  memset(o, 0, sizeof(*o));
  o->num_warmup = 1000; o->num_samples = 1000; o->max_depth = 10;
  o->init_buffer = 75; o->term_buffer = 50; o->window = 25;
  o->delta = 0.8; o->gamma = 0.05; o->kappa = 0.75; o->t0 = 10; o->stepsize = 1.0; o->init_radius = 2.0;
  o->seed = 1843; o->fast_grad = 0; o->save_warmup = 0;
}

static int sample_chain_impl(const oracle_model *m, const oracle_opts *o, int chain_id, const double *q0, double *draws,
                             double *adapt_out, long long *total_leapfrogs, double *metric_out, double *timing, double budget_s);

int oracle_sample_chain(const oracle_model *m, const oracle_opts *o, int chain_id, const double *q0, double *draws,
                        double *adapt_out, long long *total_leapfrogs) {
  return sample_chain_impl(m, o, chain_id, q0, draws, adapt_out, total_leapfrogs, NULL, NULL, 0.0);
}
int oracle_sample_chain_metric(const oracle_model *m, const oracle_opts *o, int chain_id, const double *q0, double *draws,
                               double *adapt_out, long long *total_leapfrogs, double *metric_out) {
  return sample_chain_impl(m, o, chain_id, q0, draws, adapt_out, total_leapfrogs, metric_out, NULL, 0.0);
}
/* bench.py's cpu_baseline: the same run with wall-clock seconds and leapfrogs of the two phases,
 * timing = {warm-up seconds, sampling seconds, warm-up leapfrogs, sampling leapfrogs, iterations done} (initialisation
 * and the first init_stepsize are counted as warm-up, as CmdStan's "Elapsed Time" does).  budget_s > 0: the run is cut
 * at the first iteration boundary after that many seconds -- a bounded PREFIX of the configured run (same seed, same
 * adaptation schedule), for timing only (draws / adapt_out cover the iterations done). */
int oracle_sample_chain_timed(const oracle_model *m, const oracle_opts *o, int chain_id, double *draws, double *adapt_out,
                              double *timing, double budget_s) {
  long long nl = 0;
  return sample_chain_impl(m, o, chain_id, NULL, draws, adapt_out, &nl, NULL, timing, budget_s);
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static int sample_chain_impl(const oracle_model *m, const oracle_opts *o, int chain_id, const double *q0, double *draws,
                             double *adapt_out, long long *total_leapfrogs, double *metric_out, double *timing, double budget_s) {
  const int D = m->D;
  int iters_done = 0;
  const double t_start = now_s();
  double t_warm_end = t_start;
  long long lf_warm = 0;
  sampler sp; memset(&sp, 0, sizeof(sp));
  sp.m = m; sp.o = o; sp.D = D; sp.chain = chain_id; sp.iter = ITER_PRE;
  sp.lpg = o->fast_grad ? oracle_log_prob_grad_fast : oracle_log_prob_grad;
  sp.minv = vec(D); for (int i = 0; i < D; i++) sp.minv[i] = 1.0;
  sp.z = ps_alloc(D);
  sp.wf_mean = vec(D); sp.wf_m2 = vec(D); sp.tmpv = vec(D);
  memset(sp.wf_mean, 0, sizeof(double) * (size_t)D); memset(sp.wf_m2, 0, sizeof(double) * (size_t)D);
  sp.dense = o->dense_metric != 0;
  if (sp.dense) { /* dense_e_point: inv_e_metric_ = identity */
    sp.Minv = (double *)xmalloc(sizeof(double) * (size_t)D * D); sp.Lc = (double *)xmalloc(sizeof(double) * (size_t)D * D);
    sp.wf_M2d = (double *)xmalloc(sizeof(double) * (size_t)D * D);
    memset(sp.Minv, 0, sizeof(double) * (size_t)D * D); memset(sp.Lc, 0, sizeof(double) * (size_t)D * D);
    memset(sp.wf_M2d, 0, sizeof(double) * (size_t)D * D);
    for (int i = 0; i < D; i++) { sp.Minv[(size_t)i * D + i] = 1.0; sp.Lc[(size_t)i * D + i] = 1.0; }
  }
  tree_pool *pool = (o->pooled && !sp.dense && o->max_depth < 16) ? pool_make(D, o->max_depth) : NULL;
  int rc = 0;
  /* stan::services::util::initialize: U(-R,R) on the unconstrained scale, up to 100 attempts */
  if (q0) memcpy(sp.z.q, q0, sizeof(double) * (size_t)D);
  else {
    int ok = 0;
    for (uint32_t attempt = 0; attempt < 100 && !ok; attempt++) {
      for (int i = 0; i < D; i++) {
        double u = oracle_rng_uniform(o->seed, (uint32_t)chain_id, ITER_PRE, RNG_INITS, attempt, (uint32_t)i);
        sp.z.q[i] = o->init_radius * (2.0 * u - 1.0);
      }
      double lp = sp.lpg(m, sp.z.q, sp.z.g);
      ok = isfinite(lp);
      for (int i = 0; i < D && ok; i++) ok = isfinite(sp.z.g[i]);
    }
    if (!ok) { rc = POTUS_ERR_INIT; goto done; }
  }
  /* hmc_nuts_diag_e_adapt: windows, mu = log(10*stepsize), then run_adaptive_sampler's init_stepsize */
  sp.nw = o->num_warmup; sp.ib = o->init_buffer; sp.tb = o->term_buffer; sp.bw = o->window;
  if (sp.nw < 20) { /* windowed_adaptation::restart: no adaptation windows */ }
  else if (sp.ib + sp.bw + sp.tb > sp.nw) {
    sp.ib = (int)(0.15 * sp.nw); sp.tb = (int)(0.1 * sp.nw); sp.bw = sp.nw - (sp.ib + sp.tb);
  }
  sp.win_counter = 0; sp.win_size = sp.bw; sp.win_next = sp.ib + sp.win_size - 1;
  sp.nom_eps = o->stepsize;
  sp.mu = log(10.0 * o->stepsize); sp.s_bar = 0; sp.x_bar = 0; sp.ad_counter = 0;
  update_potential_gradient(&sp, &sp.z);
  init_stepsize(&sp);
  if (o->num_warmup == 0) sp.nom_eps = exp(sp.x_bar); /* engage + disengage with no transitions: complete_adaptation */
  int saved = 0;
  for (int it = 0; it < o->num_warmup + o->num_samples; it++) {
    sp.iter = (uint32_t)it;
    int warm = it < o->num_warmup;
    double accept_stat = pool ? nuts_transition_pooled(&sp, pool) : nuts_transition(&sp);
    double eps_used = sp.eps;
    if (warm) { /* adapt_diag_e_nuts::transition */
      learn_stepsize(&sp, accept_stat);
      int update = sp.nw >= 20 ? learn_variance(&sp, sp.z.q) : 0;
      if (update < 0) { rc = POTUS_ERR_STATE; goto done; }   /* adapted covariance not positive definite */
      if (update) {
        init_stepsize(&sp);
        sp.mu = log(10.0 * sp.nom_eps);
        sp.s_bar = 0; sp.x_bar = 0; sp.ad_counter = 0;
      }
      if (it == o->num_warmup - 1) sp.nom_eps = exp(sp.x_bar); /* complete_adaptation */
      if (it == o->num_warmup - 1) { t_warm_end = now_s(); lf_warm = sp.total_leapfrogs; }
    }
    iters_done = it + 1;
    if (!warm || o->save_warmup) {
      double *row = draws + (size_t)saved * (POTUS_N_SAMPLER_COLS + D);
      row[0] = -sp.z.V; row[1] = accept_stat; row[2] = eps_used; row[3] = sp.depth; row[4] = sp.n_leapfrog;
      row[5] = sp.divergent; row[6] = sp.energy;
      memcpy(row + POTUS_N_SAMPLER_COLS, sp.z.q, sizeof(double) * (size_t)D);
      saved++;
    }
    if (budget_s > 0 && now_s() - t_start > budget_s) break;
  }
  if (adapt_out) { adapt_out[0] = sp.nom_eps; memcpy(adapt_out + 1, sp.minv, sizeof(double) * (size_t)D); }
  if (metric_out) {
    if (sp.dense) memcpy(metric_out, sp.Minv, sizeof(double) * (size_t)D * D);
    else { memset(metric_out, 0, sizeof(double) * (size_t)D * D); for (int i = 0; i < D; i++) metric_out[(size_t)i * D + i] = sp.minv[i]; }
  }
  if (timing) {
    const double t_end = now_s();
    if (o->num_warmup == 0) t_warm_end = t_start;
    if (iters_done < o->num_warmup) { t_warm_end = t_end; lf_warm = sp.total_leapfrogs; }   /* cut inside the warm-up */
    timing[0] = t_warm_end - t_start; timing[1] = t_end - t_warm_end;
    timing[2] = (double)lf_warm; timing[3] = (double)(sp.total_leapfrogs - lf_warm); timing[4] = (double)iters_done;
  }
done:
  if (total_leapfrogs) *total_leapfrogs = sp.total_leapfrogs;
  pool_free_all(pool, o->max_depth);
  ps_free(&sp.z); free(sp.minv); free(sp.wf_mean); free(sp.wf_m2); free(sp.tmpv); free(sp.Minv); free(sp.Lc); free(sp.wf_M2d);
  return rc;
}

/* Single transitions from given states (tests of the dense sampler at sizes where a whole oracle run is out of reach): transition t
 * starts at qs[t] with step size eps[t] and RNG iteration iter0 + t, under the metric handed in -- dense: Minv and its lower
 * Cholesky factor Lc, both D x D row-major (used in place, not copied); diagonal: Minv holds the D diagonal elements, Lc is
 * ignored.  rows: [n][7 + D] as oracle_sample_chain's draws. */
int oracle_transitions_from(const oracle_model *m, const oracle_opts *o, int chain_id, int iter0, int n, const double *qs, const double *eps,
                            const double *Minv, const double *Lc, double *rows) {
  const int D = m->D;
  sampler sp; memset(&sp, 0, sizeof(sp));
  sp.m = m; sp.o = o; sp.D = D; sp.chain = chain_id;
  sp.lpg = o->fast_grad ? oracle_log_prob_grad_fast : oracle_log_prob_grad;
  sp.dense = o->dense_metric != 0;
  sp.minv = vec(D); sp.tmpv = vec(D);
  if (sp.dense) { sp.Minv = (double *)Minv; sp.Lc = (double *)Lc; for (int i = 0; i < D; i++) sp.minv[i] = Minv[(size_t)i * D + i]; }
  else memcpy(sp.minv, Minv, sizeof(double) * (size_t)D);
  sp.z = ps_alloc(D);
  for (int t = 0; t < n; t++) {
    sp.iter = (uint32_t)(iter0 + t);
    sp.nom_eps = eps[t];
    memcpy(sp.z.q, qs + (size_t)t * D, sizeof(double) * (size_t)D);
    const double accept_stat = nuts_transition(&sp);
    double *row = rows + (size_t)t * (POTUS_N_SAMPLER_COLS + D);
    row[0] = -sp.z.V; row[1] = accept_stat; row[2] = sp.eps; row[3] = sp.depth; row[4] = sp.n_leapfrog; row[5] = sp.divergent; row[6] = sp.energy;
    memcpy(row + POTUS_N_SAMPLER_COLS, sp.z.q, sizeof(double) * (size_t)D);
  }
  ps_free(&sp.z); free(sp.minv); free(sp.tmpv);
  return 0;
}

/* bench.py's cpu_baseline for the dense metric (BASELINE configs[4]): n leapfrogs of one chain under a dense D x D inverse metric
 * (the unit matrix plus a small symmetric perturbation, so that nothing about it can be skipped), the matrix-vector product with its
 * rows over the OpenMP threads of the box.  Returns the seconds of the n leapfrogs; *threads = OpenMP threads used, *matrix_bytes = 8 D^2. */
double oracle_time_leapfrogs_dense(const oracle_model *m, int n, double eps, uint64_t seed, int *threads, long long *matrix_bytes) {
  const int D = m->D;
  oracle_opts o; oracle_default_opts(&o); o.seed = seed; o.dense_metric = 1;
  sampler sp; memset(&sp, 0, sizeof(sp));
  sp.m = m; sp.o = &o; sp.D = D; sp.chain = 1; sp.iter = 0; sp.dense = 1;
  sp.lpg = oracle_log_prob_grad_fast;
  sp.minv = vec(D); sp.tmpv = vec(D);
  sp.Minv = (double *)xmalloc(sizeof(double) * (size_t)D * D);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < D; i++) {
    double *row = sp.Minv + (size_t)i * D;
    for (int j = 0; j < D; j++) row[j] = i == j ? 1.0 : 1e-9 / (1.0 + (double)(i > j ? i - j : j - i));
  }
  for (int i = 0; i < D; i++) sp.minv[i] = 1.0;
  sp.z = ps_alloc(D);
  for (int i = 0; i < D; i++) sp.z.q[i] = 0.1 * (2.0 * oracle_rng_uniform(seed, 1, 0, RNG_INITS, 0, (uint32_t)i) - 1.0);
  for (int j = 0; 2 * j < D; j++) {
    double a, b; oracle_rng_normal_pair(seed, 1, 0, RNG_MOMENTUM, 0, (uint32_t)j, &a, &b);
    sp.z.p[2 * j] = a; if (2 * j + 1 < D) sp.z.p[2 * j + 1] = b;
  }
  update_potential_gradient(&sp, &sp.z);
  struct timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
  for (int i = 0; i < n; i++) evolve(&sp, &sp.z, eps);
  clock_gettime(CLOCK_MONOTONIC, &b);
  volatile double sink = sp.z.V; (void)sink;
  int nt = 1;
#ifdef _OPENMP
#pragma omp parallel
  {
#pragma omp single
    nt = omp_get_num_threads();
  }
#endif
  if (threads) *threads = nt;
  if (matrix_bytes) *matrix_bytes = (long long)sizeof(double) * D * D;
  ps_free(&sp.z); free(sp.minv); free(sp.tmpv); free(sp.Minv);
  return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

/* base_hmc::init_stepsize from a given state (tests of the device samplers' window ends): the heuristic as adapt_diag_e_nuts /
 * adapt_dense_e_nuts run it after a metric update -- at point q, starting from step size eps0, under the metric handed in (as for
 * oracle_transitions_from), with the momentum draws of RNG iteration `iter` (the iteration whose transition ended the window;
 * ITER_PRE = 0xFFFFFFFF for the search before the first transition).  Returns the step size found. */
double oracle_init_stepsize_from(const oracle_model *m, const oracle_opts *o, int chain_id, uint32_t iter, const double *q, double eps0,
                                 const double *Minv, const double *Lc) {
  const int D = m->D;
  sampler sp; memset(&sp, 0, sizeof(sp));
  sp.m = m; sp.o = o; sp.D = D; sp.chain = chain_id; sp.iter = iter;
  sp.lpg = o->fast_grad ? oracle_log_prob_grad_fast : oracle_log_prob_grad;
  sp.dense = o->dense_metric != 0;
  sp.minv = vec(D); sp.tmpv = vec(D);
  if (sp.dense) { sp.Minv = (double *)Minv; sp.Lc = (double *)Lc; for (int i = 0; i < D; i++) sp.minv[i] = Minv[(size_t)i * D + i]; }
  else memcpy(sp.minv, Minv, sizeof(double) * (size_t)D);
  sp.z = ps_alloc(D);
  memcpy(sp.z.q, q, sizeof(double) * (size_t)D);
  update_potential_gradient(&sp, &sp.z);
  sp.nom_eps = eps0;
  init_stepsize(&sp);
  const double eps = sp.nom_eps;
  ps_free(&sp.z); free(sp.minv); free(sp.tmpv);
  return eps;
}

double oracle_time_leapfrogs(const oracle_model *m, int n, double eps, int fast_grad, uint64_t seed) {
  const int D = m->D;
  oracle_opts o; oracle_default_opts(&o); o.seed = seed;
  sampler sp; memset(&sp, 0, sizeof(sp));
  sp.m = m; sp.o = &o; sp.D = D; sp.chain = 1; sp.iter = 0;
  sp.lpg = fast_grad ? oracle_log_prob_grad_fast : oracle_log_prob_grad;
  sp.minv = vec(D); for (int i = 0; i < D; i++) sp.minv[i] = 1.0;
  sp.z = ps_alloc(D);
  for (int i = 0; i < D; i++) sp.z.q[i] = 0.1 * (2.0 * oracle_rng_uniform(seed, 1, 0, RNG_INITS, 0, (uint32_t)i) - 1.0);
  sample_p(&sp, &sp.z, RNG_MOMENTUM, 0);
  update_potential_gradient(&sp, &sp.z);
  struct timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
  for (int i = 0; i < n; i++) evolve(&sp, &sp.z, eps);
  clock_gettime(CLOCK_MONOTONIC, &b);
  volatile double sink = sp.z.V; (void)sink;
  ps_free(&sp.z); free(sp.minv);
  return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
