/*
 * potus_oracle.h -- CPU fp64 restatement of the reference hot path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (libpotus_hmc.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (TheEconomist/us-potus-model) contains no tests,
 * golden vectors or runnable implementation of this path -- the arithmetic lives in
 * the third-party CmdStan 2.24.1 / Stan Math 3.3 / Eigen 3.3.7 toolchain, which is
 * neither vendored nor installable here.  This oracle restates
 *   - scripts/model/poll_model_2020.stan:42-140 (and the no_mode_adjustment variant)
 *     line by line, and
 *   - the published Stan 2.24 NUTS algorithm (stan/mcmc/hmc/nuts/base_nuts.hpp,
 *     stan/mcmc/hmc/integrators/expl_leapfrog.hpp, stan/mcmc/hmc/hamiltonians/
 *     diag_e_metric.hpp, stan/mcmc/stepsize_adaptation.hpp, stan/mcmc/
 *     windowed_adaptation.hpp, stan/mcmc/var_adaptation.hpp, stan/mcmc/hmc/base_hmc.hpp,
 *     stan/services/sample/hmc_nuts_diag_e_adapt.hpp) invoked at
 *     scripts/model/final_2016.R:533-541,
 * and is itself cross-checked by finite differences and by torch-fp64 autograd of an
 * independent literal transcription of the Stan program (tests/test_oracle.py).
 */
#ifndef POTUS_ORACLE_H
#define POTUS_ORACLE_H

#include <stdint.h>
#include "../include/potus_hmc.h" /* potus_data: the Stan data block as a C struct */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_model oracle_model;

typedef struct oracle_opts {
  int32_t num_warmup, num_samples, max_depth;
  int32_t init_buffer, term_buffer, window;
  double delta, gamma, kappa, t0, stepsize, init_radius;
  uint64_t seed;
  int32_t fast_grad;   /* 0: literal dense recursion (stan:86); 1: scan/sparse reformulation */
  int32_t save_warmup;
  int32_t dense_metric; /* 0: diag_e (what the reference runs); 1: dense_e (stan::mcmc::dense_e_metric + covar_adaptation) */
  int32_t pooled;      /* 1 (diagonal metric): pooled buffers and fused loops -- the same draws bit for bit, faster; what
                          bench.py's cpu_baseline times.  0: the recursion written as upstream writes it */
} oracle_opts;

void oracle_default_opts(oracle_opts *o);

/* returns NULL and fills err on a data-block violation */
oracle_model *oracle_model_create(const potus_data *d, char *err, int errlen);
void oracle_model_free(oracle_model *m);
int oracle_num_params(const oracle_model *m);
int oracle_num_columns(const oracle_model *m); /* 7 + D + TP + GQ */
/* the three scaled Cholesky factors of transformed data (stan:42-55), col-major SxS */
void oracle_cholesky_factors(const oracle_model *m, double *L_B, double *L_T, double *L_W);

/* log_prob<propto=true, jacobian=true> and gradient; literal restatement */
double oracle_log_prob_grad(const oracle_model *m, const double *q, double *grad);
/* same value via suffix/prefix scans and the sparse poll structure */
double oracle_log_prob_grad_fast(const oracle_model *m, const double *q, double *grad);

/* write_array: out[0 .. n_cols-7): constrained params, transformed params, GQ */
void oracle_write_array(const oracle_model *m, const double *q, double *out);

/* Philox4x32-10 block and the derived draws shared with the device sampler */
void oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double oracle_rng_uniform(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t purpose,
                          uint32_t aux, uint32_t index);
void oracle_rng_normal_pair(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t purpose,
                            uint32_t aux, uint32_t index, double *n0, double *n1);

/* One chain of adaptive NUTS.  draws: [n_saved][7 + D] row-major (unconstrained q);
 * adapt_out: [1 + D] = final step size, inverse metric.  Returns 0 on success.
 * q0: optional initial point (NULL -> U(-r,r) inits with retry). */
int oracle_sample_chain(const oracle_model *m, const oracle_opts *o, int chain_id,
                        const double *q0, double *draws, double *adapt_out,
                        long long *total_leapfrogs);

/* the same with the adapted dense inverse metric returned as well (metric_out: D*D row-major, or NULL);
 * with o->dense_metric = 0 it is filled with diag(minv) */
int oracle_sample_chain_metric(const oracle_model *m, const oracle_opts *o, int chain_id,
                               const double *q0, double *draws, double *adapt_out,
                               long long *total_leapfrogs, double *metric_out);

/* bench.py's cpu_baseline: oracle_sample_chain with the wall-clock seconds and leapfrogs of the two phases,
 * timing[5] = {warm-up s, sampling s, warm-up leapfrogs, sampling leapfrogs, iterations done}; budget_s > 0 cuts the
 * run at the first iteration boundary after that many seconds (a bounded prefix of the configured run) */
int oracle_sample_chain_timed(const oracle_model *m, const oracle_opts *o, int chain_id, double *draws,
                              double *adapt_out, double *timing, double budget_s);

/* n single transitions, each from its own given state qs[t] ([n][D]) with step size eps[t] and RNG iteration iter0 + t, under a
 * given metric (dense: Minv and its lower Cholesky factor Lc, D x D row-major, used in place; diagonal: Minv = the D diagonal
 * elements).  rows: [n][7 + D]. */
int oracle_transitions_from(const oracle_model *m, const oracle_opts *o, int chain_id, int iter0, int n, const double *qs,
                            const double *eps, const double *Minv, const double *Lc, double *rows);

/* the same under a dense D x D inverse metric (BASELINE configs[4]); the product's rows run on the OpenMP threads of the box:
 * *threads = how many, *matrix_bytes = 8 D^2 */
double oracle_time_leapfrogs_dense(const oracle_model *m, int n, double eps, uint64_t seed, int *threads, long long *matrix_bytes);

/* base_hmc::init_stepsize at point q from step size eps0 under the given metric (Minv / Lc as for oracle_transitions_from), with the
 * momentum draws of RNG iteration iter: what the adaptive samplers run after every metric update.  Returns the step size. */
double oracle_init_stepsize_from(const oracle_model *m, const oracle_opts *o, int chain_id, uint32_t iter, const double *q, double eps0,
                                 const double *Minv, const double *Lc);

/* leapfrog micro-benchmark for bench.py's cpu_baseline: n steps from q0 with unit metric */
double oracle_time_leapfrogs(const oracle_model *m, int n, double eps, int fast_grad, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
