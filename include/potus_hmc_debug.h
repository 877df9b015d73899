/*
 * potus_hmc_debug.h -- development and verification exports of libpotus_hmc.so that are NOT part of the drop-in boundary
 * (include/potus_hmc.h): nothing in the reference corresponds to them and no R / Python product code calls them.  They exist for
 * tests/ (device state behind a handle, kernels of the dense metric on matrices a test hands in) and scripts/ (in-kernel cycle
 * counters of -DPOTUS_PROF builds).  Declared here so that every symbol the library exports is declared somewhere
 * (tests/test_abi.py checks both directions).
 */
#ifndef POTUS_HMC_DEBUG_H
#define POTUS_HMC_DEBUG_H

#include "potus_hmc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The whole state block of a handle, [chains (x 2 in twin mode)][V_COUNT][Dpad] doubles in the kernels' internal element order, and
 * every replica of the chain scalars as raw bytes.  which = 0: sizes only (out[0] = V_COUNT, out[1] = Dpad, out[2] = bytes of
 * scalars); which = 1: the data.  Returns 1 on success (a development hook, not a status code). */
int potus_debug_state(int handle, int which, double *out, unsigned char *scal);
/* -DPOTUS_PROF builds: the in-kernel cycle counters, [chains x members (x 2)][64] doubles (scripts/gpu_probe.py).  Returns 1 when
 * the build carries them, 0 otherwise. */
int potus_debug_profile(int handle, double *out);
/* Which build of the cluster pass the handle runs: 4 / 8 = days per wave with every size read from the model descriptor, 12 = 4 days per
 * wave with the adjoint product on the fp64 matrix cores, 16 / 17 = the fixed-layout builds of poll_model_2020.stan / its
 * no_mode_adjustment variant; 0 = one workgroup per chain; -1 = bad handle (tests: which posterior gets which kernel). */
int potus_debug_build_tag(int handle);
/* Cluster mode: what the last launch found about its placement (potus_cluster.hpp, cl_find_local), one int per chain and side ([side][chain]): 1 = every
 * member of that cluster ran on one XCD and published its exchange words with plain stores, 0 = write-through.  Returns the number of ints written, -1 for a
 * bad handle or a one-workgroup sampler.  (POTUS_DEBUG_DROP_MEMBER=-1 at potus_create forces the write-through path: the two must give the same bytes.) */
int potus_debug_xcd_local(int handle, int *out);
/* Dense metric, without a sampler: y = M^-1 x for `chains` matrices (D x D, row-major, upper triangle read) and nrhs <= 3 vectors
 * each by the sampler's matrix pass (k_dn_symv + k_dn_symv_finish), repeated `reps` times; dot_host: x_0 . y_0 per chain; ms: time
 * of the passes; pass_bytes: bytes of matrix one pass loads (tests/test_gpu_dense.py, scripts/micro/dense_probe.py). */
int potus_dense_matvec_probe(int device, int chains, int D, int nrhs, const double *Minv_host, const double *x_host, double *y_host, double *dot_host, int reps,
                             double *ms, long long *pass_bytes);
/* Dense metric, without a sampler: the window end's kernels on n draws per chain handed in (covariance -> regularised inverse
 * metric -> blocked Cholesky -> back substitution L' p = u); any output pointer may be NULL.  ms[3]: covariance, factorisation,
 * solve. */
int potus_dense_factor_probe(int device, int chains, int D, int n, const double *draws_host, const double *u_host, double *Minv_host, double *L_host,
                             double *p_host, double *ms);
/* The same two probes for the POOLED dense metric (potus_opts.pooled_metric, csrc/potus_dense_pool.hpp): ONE D x D matrix (full symmetric storage)
 * for all chains -- Minv_host [D][D] (NULL: generated on the device) -- every right-hand side of every chain in one pass on the fp64 matrix cores
 * (k_dn_pool_mm + k_dn_pool_finish), pass_bytes = the whole matrix; and the pooled window end: ONE regularised covariance of chains x n draws ->
 * Minv_host [D][D], ONE factor L_host [D][D], p = L^-T u for every chain's u [chains][D]. */
int potus_dense_pool_matvec_probe(int device, int chains, int D, int nrhs, const double *Minv_host, const double *x_host, double *y_host, double *dot_host, int reps,
                                  double *ms, long long *pass_bytes);
int potus_dense_pool_factor_probe(int device, int chains, int D, int n, const double *draws_host, const double *u_host, double *Minv_host, double *L_host,
                                  double *p_host, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* POTUS_HMC_DEBUG_H */
