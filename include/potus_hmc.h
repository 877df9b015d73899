/*
 * potus_hmc.h -- C ABI of libpotus_hmc.so, the MI355X-native HMC/NUTS sampler for the
 * posterior defined by scripts/model/poll_model_2020.stan (and the
 * poll_model_2020_no_mode_adjustment.stan variant) of TheEconomist/us-potus-model.
 *
 * What each entry point replaces in the reference
 * -----------------------------------------------
 * The reference has no FFI: the R scripts hand the `data` list to CmdStan through
 * files and a process boundary:
 *
 *   scripts/model/final_2016.R:475-514   data <- list(...)            -> potus_data
 *   scripts/model/final_2016.R:532       cmdstan_model(..., compile)  -> potus_create
 *   scripts/model/final_2016.R:533-541   model$sample(data, seed, chains, iter_warmup,
 *                                        iter_sampling, refresh)      -> potus_run (chunked)
 *   scripts/model/final_2016.R:543       rstan::read_stan_csv(files)  -> potus_get_draws /
 *                                                                        potus_write_array /
 *                                                                        potus_write_stan_csv
 *   scripts/model/final_2016.R:556,...   rstan::extract(out, pars=)   -> potus_write_array
 *   (same call sites: final_2012.R:558-569, final_2008.R:562-573; the commented rstan
 *    surface at final_2016.R:525-529 and scripts/deprecated/R/Refactored/poll_run_v9.R:387-390)
 *
 * Conventions
 * -----------
 *  - plain C, no C++/torch types; every function returns an int status (0 = ok) and
 *    the message of the last failure is available from potus_last_error().
 *  - the caller owns every buffer it passes; inputs are copied at potus_create and no
 *    caller pointer is retained.  Device memory, streams and kernels live behind the
 *    integer handle.
 *  - indices inside potus_data are 1-based int32 exactly as in the Stan `data{}` block
 *    (poll_model_2020.stan:1-41); matrices are column-major.
 *  - all arithmetic is fp64.
 *  - the `_R` entry points take only int* / double* / char** so that R's .C() can call
 *    them (R passes every argument by pointer and ignores return values, so they also
 *    write *status).
 */
#ifndef POTUS_HMC_H
#define POTUS_HMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POTUS_VARIANT_FULL 0    /* scripts/model/poll_model_2020.stan */
#define POTUS_VARIANT_NO_MODE 1 /* scripts/model/poll_model_2020_no_mode_adjustment.stan */

/* status codes */
#define POTUS_OK 0
#define POTUS_ERR_ARG 1      /* bad argument / Stan data-block constraint violated */
#define POTUS_ERR_DEVICE 2   /* HIP runtime failure or no gfx950 device */
#define POTUS_ERR_INIT 3     /* no finite initial point after 100 attempts */
#define POTUS_ERR_STATE 4    /* call order / handle state */
#define POTUS_ERR_IO 5
#define POTUS_ERR_UNSUPPORTED 6
#define POTUS_ERR_STEPSIZE 7 /* base_hmc::init_stepsize ran the step size to 0 or beyond 1e7 (Stan throws there) */
#define POTUS_ERR_WATCHDOG 8 /* the workgroups of a chain's cluster were not resident together (GPU shared with
                                another process?): the launch gave up instead of hanging; the handle is dead */

/* The Stan data block (poll_model_2020.stan:1-41) as a C struct.  For the no-mode
 * variant the four poll_mode_* / poll_pop_* pointers, M, Pop, sigma_m, sigma_pop and
 * sigma_e_bias and the unadjusted_* vectors are ignored (as Stan ignores extra list
 * entries) and may be NULL / 0. */
typedef struct potus_data {
  int32_t N_national_polls, N_state_polls, T, S, P, M, Pop;
  const int32_t *state;               /* [N_state_polls]    1..S   (stan:9)  */
  const int32_t *day_state;           /* [N_state_polls]    1..T   (stan:10) */
  const int32_t *day_national;        /* [N_national_polls] 1..T   (stan:11) */
  const int32_t *poll_state;          /* [N_state_polls]    1..P   (stan:12) */
  const int32_t *poll_national;       /* [N_national_polls] 1..P   (stan:13) */
  const int32_t *poll_mode_state;     /* 1..M    (stan:14) */
  const int32_t *poll_mode_national;  /* 1..M    (stan:15) */
  const int32_t *poll_pop_state;      /* 1..Pop  (stan:16) */
  const int32_t *poll_pop_national;   /* 1..Pop  (stan:17) */
  const int32_t *n_democrat_national; /* (stan:18) */
  const int32_t *n_two_share_national;
  const int32_t *n_democrat_state;
  const int32_t *n_two_share_state;
  const double *unadjusted_national;  /* in [0,1] (stan:22) */
  const double *unadjusted_state;     /* in [0,1] (stan:23) */
  const double *mu_b_prior;           /* [S] (stan:28) */
  const double *state_weights;        /* [S] (stan:29) */
  double sigma_c, sigma_m, sigma_pop;
  double sigma_measure_noise_national, sigma_measure_noise_state, sigma_e_bias;
  const double *state_covariance_0;   /* [S*S] column-major, symmetric PD (stan:37) */
  double random_walk_scale, mu_b_T_scale, polling_bias_scale;
  int32_t variant;                    /* POTUS_VARIANT_* */
} potus_data;

/* Sampler options: the argument surface of cmdstanr's $sample() as used at
 * final_2016.R:533-541 plus the CmdStan 2.24 defaults it implies.
 * ALWAYS start from potus_default_opts(): fields are appended over time (metric_storage came with library version 0.4 in what used
 * to be tail padding, pooled_metric with 0.5), and a struct filled field by field by a caller built against an older header leaves them undefined;
 * potus_version() names the version the header must match. */
typedef struct potus_opts {
  int32_t chains;          /* chains run by THIS handle                                  */
  int32_t chain_id_offset; /* global id of this handle's first chain minus 1; chain c of
                              the handle uses RNG stream chain_id_offset + c + 1, so the
                              draws do not depend on how chains are split over GPUs       */
  int32_t num_warmup;      /* iter_warmup   (final_2016.R:538) */
  int32_t num_samples;     /* iter_sampling (final_2016.R:539) */
  int32_t max_depth;       /* 10 */
  int32_t init_buffer, term_buffer, window; /* 75, 50, 25 */
  double delta, gamma, kappa, t0;           /* 0.8, 0.05, 0.75, 10 */
  double stepsize;         /* 1.0 */
  double init_radius;      /* 2.0 : inits ~ U(-2,2) on the unconstrained scale */
  uint64_t seed;           /* 1843 (final_2016.R:535) */
  int32_t device;          /* HIP device ordinal */
  int32_t save_warmup;     /* 0 */
  int32_t cus_per_chain;   /* compute units (workgroups) cooperating on one chain: 1 = one workgroup per
                              chain (best throughput with >= 64 chains), 2..32 = a cluster per chain
                              (lowest latency with few chains; chains * cus_per_chain <= CUs of the device),
                              0 = choose from {16, 8, 4, 1} by what fits (10-14 for
                              9-12 chains, as two clusters each, and 1 with two workgroups per chain for 65-128 chains: see twin).  Draws are reproducible bit for bit
                              for a given value; different values differ in floating-point summation order. */
  int32_t metric;          /* POTUS_METRIC_DIAG (CmdStan's default, what final_2016.R:533-541 runs) or
                              POTUS_METRIC_DENSE (metric = "dense_e": stan::mcmc::dense_e_metric + covar_adaptation,
                              BASELINE configs[4]) */
  int32_t twin;            /* cluster mode, diagonal metric: 1 = TWO clusters of cus_per_chain workgroups per chain, one per
                              end of the NUTS trajectory -- the doublings that go forward and those that go backward are
                              integrated at the same time (2 * chains * cus_per_chain <= CUs of the device); 0 = one
                              cluster; -1 = the library decides when it also chooses the cluster size (cus_per_chain = 0):
                              two clusters if they fit.  Same algorithm, RNG streams and arithmetic: the draws are the
                              same bytes as with one cluster of the same size.  With cus_per_chain = 1 the "cluster" is one
                              workgroup: two workgroups per chain (2 * chains <= resident workgroups of the device; what
                              the library picks for 65-128 chains on 256 compute units).  Every form with more than one
                              workgroup per chain needs ALL its workgroups resident together, i.e. the GPU to itself: beside
                              another process the launch ends with POTUS_ERR_WATCHDOG after ~1-3 s and the handle is dead (ask
                              for twin = 0, cus_per_chain = 1 on a shared GPU). */
  int32_t metric_storage;  /* dense metric only: POTUS_STORAGE_F64 (Stan's) or POTUS_STORAGE_F32 -- the adapted covariance is
                              rounded to fp32 and THAT matrix is the metric: its Cholesky factor (fp64) draws the momenta, the
                              leapfrog multiplies with it (fp64 accumulation), so the sampler stays exact while the matrix pass
                              of every leapfrog streams 2 D^2 bytes instead of 4 D^2.  A declared deviation from Stan, which
                              keeps the covariance in fp64; same memory per chain.  With pooled_metric
                              the handle's ONE matrix is kept in fp32 beside its fp64 factor and the pooled pass streams 4 D^2
                              bytes per round instead of 8 D^2. */
  int32_t pooled_metric;   /* dense metric only, 0 (Stan's: every chain adapts its own covariance), 1 or 2 (as 1, but every window end is finished by
                              the host, which may pool over several handles and GPUs first: potus_dense_pool_window).  1: at every window end the draws of ALL
                              chains of the handle form ONE regularised covariance -- covar_adaptation::learn_covariance applied to the
                              pooled sample of chains x n draws -- and ONE Cholesky factor (library version 0.5).  A leaf round then streams
                              one D x D matrix once for every chain (M^-1 times a D x (chains x right-hand sides) block on the fp64 matrix
                              cores) instead of one triangle per chain, and the handle keeps two matrices instead of one per chain.  A
                              declared deviation from Stan / CmdStan, whose chains are separate processes and cannot pool (SURVEY.md
                              section 7.3-5, section 8 f4); the sampler stays exact for the metric it uses.  fp64 storage only. */
  int32_t reserved_;       /* (keeps the struct a multiple of 8 bytes; set by potus_default_opts) */
} potus_opts;
#define POTUS_METRIC_DIAG 0
#define POTUS_METRIC_DENSE 1
#define POTUS_STORAGE_F64 0
#define POTUS_STORAGE_F32 1

/* per-draw sampler columns, in CmdStan order */
#define POTUS_N_SAMPLER_COLS 7 /* lp__,accept_stat__,stepsize__,treedepth__,n_leapfrog__,divergent__,energy__ */

const char *potus_version(void);
int potus_last_error(char *buf, int len);
void potus_default_opts(potus_opts *o);

/* Number of unconstrained parameters D and of columns of one full output row
 * (7 sampler + D constrained parameters + transformed parameters + generated
 * quantities; 43 360 for the 2016 data). Pure host arithmetic, no device needed. */
int potus_num_params(const potus_data *d, int *D);
int potus_num_columns(const potus_data *d, int *n_cols);
/* Column names in CmdStan CSV order ("raw_mu_b.3.17" style, column-major).  Writes at
 * most n_names pointers into a caller array of char[name_len] rows. */
int potus_column_name(const potus_data *d, int col, char *buf, int len);

/* Validate data (Stan's declared bounds), build the three scaled Cholesky factors
 * (transformed data, stan:42-55), upload everything, allocate chain state. */
int potus_create(const potus_data *d, const potus_opts *o, int *handle);
int potus_destroy(int handle);
/* Compute units (workgroups) per chain the handle actually runs with (cus_per_chain = 0 resolved). */
int potus_cus_per_chain(int handle, int *k);
/* clusters per chain of the handle: 2 when it runs the two ends of the trajectory on a cluster each (potus_opts.twin),
 * 1 otherwise */
int potus_clusters_per_chain(int handle, int *n);
/* How potus_create resolves cus_per_chain = 0 and twin = -1, as pure functions of the sizes (no device needed; the reference has
 * no counterpart: CmdStan runs one process per chain, scripts/model/final_2016.R:533-541 `parallel_chains`).
 * potus_plan_cus_per_chain: the first plan for the workgroups per chain (potus_create may still double it when a member's polls
 * do not fit its LDS); one_workgroup_ok = 0 for models beyond the one-workgroup kernels (T > 256 or > 2 048 polls);
 * *lowered_for_twin = 1 when 10-14 was chosen instead of 16 so that a second cluster per chain fits.
 * potus_plan_sides: 1 or 2 clusters (K > 1) / workgroups (K = 1) per chain; resident_per_cu = workgroups of the kernel a compute
 * unit holds (1 for these kernels on gfx950). */
int potus_plan_cus_per_chain(int chains, int T, int n_cus, int cus_per_chain, int twin, int metric, int one_workgroup_ok, int *K, int *lowered_for_twin);
int potus_plan_sides(int chains, int K, int n_cus, int resident_per_cu, int cus_per_chain, int twin, int metric, int *sides);
/* twin mode: leapfrogs counted in the trajectories (n_leapfrog__ summed, = potus_total_leapfrogs) and leaves each side has
 * integrated, those of speculative subtrees that were dropped included -- what the second cluster costs and buys */
int potus_twin_stats(int handle, long long *counted, long long *run_backward, long long *run_forward);

/* Parity hook: log-density (with Jacobians, constants dropped as `~` does) and its
 * gradient for n points of the unconstrained space, evaluated by the same device
 * code the leapfrog uses.  q: [n][D], lp: [n], grad: [n][D] (host pointers). */
int potus_log_prob_grad(int handle, const double *q, int n, double *lp, double *grad);

/* Initial values ~ U(-r,r) with retry (CmdStan semantics), then the initial
 * step-size search.  Must be called once before potus_run. Optional user inits:
 * q0 [chains][D] or NULL. */
int potus_init(int handle, const double *q0);

/* Advance every chain by n_iter NUTS transitions (warmup transitions adapt).
 * Blocks until done.  R calls this in chunks of `refresh` iterations. */
int potus_run(int handle, int n_iter);

/* The same for several handles at once: all launches are issued before any is waited for, so the
 * handles of different GPUs (shards of one posterior, BASELINE configs[2]) or of different posteriors on
 * one GPU (the 2008 / 2012 / 2016 backtests, configs[3]) run concurrently under a single host thread --
 * R has only one.  No reference counterpart: cmdstanr gets its concurrency from one OS process per chain
 * (final_2016.R:536 parallel_chains).  Handles that cannot be co-resident on their device run in turn. */
int potus_run_many(const int *handles, int n_handles, int n_iter);

/* Progress / accounting. */
int potus_iterations_done(int handle, int *n);
int potus_total_leapfrogs(int handle, long long *n); /* sum over chains and iterations so far */
int potus_chain_status(int handle, int *status /*[chains]*/, int *n_divergent /*[chains]*/);

/* Adaptation result per chain: step size and diagonal inverse metric (dense metric: its diagonal). */
int potus_get_adaptation(int handle, double *stepsize /*[chains]*/, double *inv_metric /*[chains][D]*/);
/* Dense metric only: the D x D inverse metric of one chain of the handle (row-major = column-major, it is symmetric). */
int potus_get_dense_metric(int handle, int chain, double *inv_metric /*[D][D]*/);

/* Saved draws on the unconstrained scale: out[chain][iter][7 + D] (host pointer).
 * n_saved = num_samples (+ num_warmup when save_warmup). */
int potus_get_draws(int handle, double *out, int *n_saved);

/* Device pointer of the same array (for RCCL all-gather through torch); the buffer
 * stays owned by the handle. */
int potus_draws_device_ptr(int handle, void **dptr, long long *n_doubles);

/* write_array: constrained parameters, transformed parameters (stan:70-113) and
 * generated quantities (stan:134-140) for every saved draw, restricted to the columns
 * [col_begin, col_end) of the full CmdStan row (0-based, including the 7 sampler
 * columns).  out[iter][chain][col_end-col_begin] -- the as.array(stanfit) layout. */
int potus_write_array(int handle, int col_begin, int col_end, double *out);
/* The same rows written into DEVICE memory of the handle's GPU (e.g. the data_ptr() of a torch tensor that an RCCL
 * all-gather then sends: the draws-of-interest never visit the host). */
int potus_write_array_device(int handle, int col_begin, int col_end, void *out_device);
/* rstan::extract(out, pars)[[1]] (final_2016.R:556, :708) as R stores it, filled in place in ONE pass: `out` = a column-major matrix
 * [rows, col_end - col_begin] whose row chain_global * n_saved + iteration holds a draw -- chains merged chain after chain, the handles' chains in the order
 * listed.  out = NULL: only *rows_out (the draws the handles hold) is set, to size the result; otherwise rows must equal it.  This is what the .Call()
 * wrapper R/src/potus_call.c hands an allocMatrix'ed result to (long vectors welcome); the .C() path returns [iteration][chain][column] rows that the R shim
 * must permute -- two more copies of the block. */
int potus_extract_matrix(const int *handles, int n_handles, int col_begin, int col_end, double *out, long long rows, long long *rows_out);

/* One CmdStan-format CSV per chain (<dir>/<basename>-<chain>.csv) readable by
 * rstan::read_stan_csv (final_2016.R:543). */
int potus_write_stan_csv(int handle, const char *dir, const char *basename);

/* Posterior summaries the run scripts build from extract(out, "predicted_score") (final_2016.R:708-762 state and
 * national intervals, :799-823 electoral-college simulation), computed on the device from the saved draws of
 * all chains of the handle (pooled; any number of draws).  Cell order of state_out: t + T*s (CmdStan's column-major
 * predicted_score[T,S]); quantiles are R's default (type 7); the national vote is weighted.mean(score, state_weights).
 *   state_out [T*S][4] = low (2.5 %), high (97.5 %), mean, P(score > 0.5)
 *   natl_out  [T][4]   = the same for the state_weights-weighted national vote of each draw
 *   ev_out    [T][5]   = mean, median, high, low, P(>= 270) of sum_s ev[s] 1[score > 0.5]      (ev: [S]) */
int potus_posterior_summary(int handle, const double *ev, double *state_out, double *natl_out, double *ev_out);
/* The same over the pooled draws of several handles of one posterior (its chains spread over several samplers or
 * GPUs: potus_run_many); runs on the first handle's GPU. */
int potus_posterior_summary_many(const int *handles, int n_handles, const double *ev, double *state_out, double *natl_out,
                                 double *ev_out);
/* Backtest scores of final_2016.R:925-945 (final_2012.R:918-931, final_2008.R:922-935) from state_out: with p_s =
 * P(score > 0.5) of state s on `day` (1-based; 0 = last day) and won[s] in {0,1} the outcome,
 * out[3] = EV-weighted Brier score, unweighted Brier score, states called correctly (round(p) == won). */
int potus_backtest_scores(const double *state_out, int T, int S, int day, const double *ev, const int *won, double *out);

/* Cross-chain diagnostics on the device: rank-normalised split R-hat and bulk ESS (Vehtari et al. 2021; the definitions bench.py's
 * ESS/s uses) of the columns [col_begin, col_end) of the output row, over the pooled chains of several handles of ONE posterior
 * (equal numbers of saved draws).  The reference has no counterpart (final_2016.R:543-556 never looks at a diagnostic); the
 * all-gather of BASELINE.json's north_star exists "to pool draws for R-hat / ESS".  rhat_out, ess_bulk_out: [col_end - col_begin].
 * Warm-up rows saved with save_warmup = 1 are left out, as rstan::monitor / extract() leave them out; at most 512 chains pooled; a
 * column that holds a NaN or an infinite draw gets NaN for both, a constant column NaN as in `posterior`. */
int potus_diagnostics(const int *handles, int n_handles, int col_begin, int col_end, double *rhat_out, double *ess_bulk_out);
/* The same for a block that already sits in DEVICE memory of GPU `device`: block[draw][chain][column] -- the layout
 * potus_write_array_device produces and an RCCL all-gather of it keeps.  rhat_out, ess_bulk_out: host arrays [n_cols]. */
int potus_diagnostics_device(int device, const void *block, long long n_draws, int n_chains, int n_cols, double *rhat_out, double *ess_bulk_out);

/* Online convergence check for a host loop that advances the sampler in chunks ("online R-hat-based early
 * stop"): rank-normalised split R-hat and bulk ESS of lp__ and mu_b[:, T] (what predicted_score[T, :], the quantity the scripts report, is a
 * monotone map of: final_2016.R:708-762) over the post-warm-up draws saved so far by the pooled chains of the handles.  *converged = 1 when
 * every R-hat is below rhat_below and every bulk ESS is at least ess_at_least (fewer than four draws: 0, no error).  The sampler itself never
 * looks at the flag: the draws up to that point are those of an uninterrupted run.  Stopping on it is a DEVIATION from Stan / the reference,
 * which always run iter_sampling iterations (final_2016.R:539): the host has to ask (argument rhat_stop of the R shim's and the
 * Python host's sample functions). */
int potus_check_convergence(const int *handles, int n_handles, double rhat_below, double ess_at_least, int *converged, double *rhat_max, double *ess_bulk_min);

/* Kernel timing of the most recent potus_run, measured with HIP events on the
 * sampler's own stream: elapsed milliseconds and leapfrogs executed in it. */
int potus_last_run_timing(int handle, double *ms, long long *leapfrogs);

/* Dense metric only: milliseconds spent in the matrix passes (k_dn_matvec: M^-1 times the momenta of a leaf, HIP events on
 * the sampler's stream), their number, the bytes of matrix they streamed (active chains x D x LD x 8 each) and the
 * number of leaf rounds, since potus_create. */
int potus_dense_timing(int handle, double *matvec_ms, long long *passes, long long *bytes, long long *rounds);
/* Dense metric only: what the window ends of the warm-up (covar_adaptation::learn_covariance, then base_hmc::init_stepsize) have
 * cost since potus_create: milliseconds in the covariance, in the blocked Cholesky factorisation and in the step-size search,
 * and the number of window ends. */
int potus_dense_adapt_timing(int handle, double *cov_ms, double *chol_ms, double *init_stepsize_ms, int *window_ends);
/* potus_opts.pooled_metric = 2: the pooled window end in two halves, so that the HOST can pool further -- over the handles of a process, and through an
 * all-reduce (RCCL) over the GPUs of a node (us_potus_model_amd/parallel.py: pool_window_moments, sampler.run_pooled).
 * potus_run / potus_run_many stop after the transition that ends a window (potus_iterations_done says where; a further potus_run before the finish is
 * POTUS_ERR_STATE).  potus_dense_pool_window: *pending = 1 then; *count = the draws behind the handle's moments (chains x window length); *mean_dev and
 * *m2_dev = DEVICE pointers to the handle's mean [D] and M2 = sum of the centred outer products [D rows of *ld doubles, both triangles].  The host
 * replaces M2 by the pooled one (Chan's update: M2 += count (mean - pooled mean)(mean - pooled mean)', then the sum over all handles and ranks) and
 * calls potus_dense_pool_finish with the pooled count N: M^-1 = N/(N+5) M2/(N-1) + 1e-3 5/(N+5) I, its Cholesky factor, base_hmc::init_stepsize.
 * With one handle and nothing done in between, finish(count) is exactly pooled_metric = 1. */
int potus_dense_pool_window(int handle, int *pending, double *count, void **mean_dev, void **m2_dev, long long *ld);
int potus_dense_pool_finish(int handle, double n_total);
/* Dense metric only, verification hook (as potus_log_prob_grad is for the gradient): is the factor L the momentum draw solves
 * with the Cholesky factor of the inverse metric the leapfrog multiplies with?  For n_probe standard-normal vectors x,
 * M^-1 x by the sampler's own matrix pass against L (L' x) by plain kernels over the factor, and the momentum draw's blocked
 * back substitution L' p = u multiplied back:
 *   out[0] = max ||L L' x - M^-1 x|| / ||M^-1 x||,   out[1] = ||L' p - u|| / ||u||.
 * Only BETWEEN runs of an initialised handle whose last window end succeeded (POTUS_ERR_STATE otherwise): the check uses the
 * sampler's own scratch vectors (momentum, temporaries) and round descriptors of every chain, which the next potus_run re-arms;
 * its passes are not part of potus_dense_timing's counts. */
int potus_dense_check(int handle, int chain, int n_probe, double *out /*[2]*/);

/* ---- .C()-callable wrappers (int* / double* / char** only) ---- */
void potus_R_create(int *dims /*[8]: N_nat,N_state,T,S,P,M,Pop,variant*/,
                    int *state, int *day_state, int *day_national, int *poll_state,
                    int *poll_national, int *poll_mode_state, int *poll_mode_national,
                    int *poll_pop_state, int *poll_pop_national, int *n_democrat_national,
                    int *n_two_share_national, int *n_democrat_state, int *n_two_share_state,
                    double *unadjusted_national, double *unadjusted_state, double *mu_b_prior,
                    double *state_weights, double *scalars /*[9]: sigma_c,sigma_m,sigma_pop,
                    sigma_noise_nat,sigma_noise_state,sigma_e_bias,random_walk_scale,
                    mu_b_T_scale,polling_bias_scale*/,
                    double *state_covariance_0,
                    int *iopts /*[12]: chains,chain_id_offset,num_warmup,num_samples,max_depth,
                    device,save_warmup,cus_per_chain,metric,twin,metric_storage,pooled_metric*/,
                    double *dopts /*[7]: delta,gamma,kappa,t0,stepsize,init_radius,seed (an integer < 2^53:
                    R's own integers have 32 bits)*/,
                    int *handle, int *status);
void potus_R_init(int *handle, int *status);
void potus_R_run(int *handle, int *n_iter, int *status);
void potus_R_run_many(int *handles, int *n_handles, int *n_iter, int *status);
void potus_R_num_columns(int *handle, int *D, int *n_cols, int *status);
void potus_R_write_array(int *handle, int *col_begin, int *col_end, double *out, int *status);
void potus_R_write_stan_csv(int *handle, char **dir, char **basename, int *status);
void potus_R_saved_count(int *handle, int *n_saved, int *status);
void potus_R_posterior_summary(int *handles, int *n_handles, double *ev, double *state_out, double *natl_out, double *ev_out, int *status);
void potus_R_diagnostics(int *handles, int *n_handles, int *cols /*[2]: col_begin, col_end*/, double *rhat_out, double *ess_bulk_out, int *status);
void potus_R_check_convergence(int *handles, int *n_handles, double *limits /*[2]: rhat_below, ess_at_least*/, int *converged, double *out /*[2]: rhat_max, ess_bulk_min*/,
                                int *status);
void potus_R_backtest_scores(double *state_out, int *dims /*[3]: T, S, day*/, double *ev, int *won, double *out /*[3]*/, int *status);
void potus_R_last_error(char **buf, int *len);
void potus_R_destroy(int *handle, int *status);

#ifdef __cplusplus
}
#endif
#endif /* POTUS_HMC_H */
