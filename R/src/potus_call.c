/* potus_call.c -- .Call() entry points of the R host shim (R/potus_sampling.R) over the C ABI of libpotus_hmc.so.
 *
 * The .C() surface (potus_R_*, include/potus_hmc.h) needs no R headers, but .C() duplicates every argument on the way in and
 * on the way out and refuses long vectors: rstan::extract(out, pars = "predicted_score")[[1]] at final_2016.R:708 is 0.83 GB
 * for 8 x 1000 draws -- copied twice by .C() and permuted once more by the shim -- and the 64 000 pooled draws x 43 360 columns
 * of BASELINE configs[2] (2.8e9 elements) cannot pass at all.  Here the result is allocated ONCE (allocMatrix: long vectors
 * welcome) and potus_extract_matrix fills it in place, already in R's column-major [draws, columns] order.
 *
 * Build (where R is installed; it is not in the build image of this repository -- tests/test_abi.py compiles this very file
 * against the stub headers under tests/r_stub/ and drives it through them):
 *     R CMD SHLIB -o R/src/potus_call.so R/src/potus_call.c -I include -L us_potus_model_amd -lpotus_hmc
 * Load:  dyn.load("us_potus_model_amd/libpotus_hmc.so", local = FALSE); dyn.load("R/src/potus_call.so")
 * The shim prefers these entry points when is.loaded("potus_call_extract") and falls back to .C().
 * Errors become R errors (Rf_error with the library's message); nothing here keeps a pointer beyond the call. */
#include <R.h>
#include <Rinternals.h>
#include "potus_hmc.h"

static void potus_call_fail(int status) {
  char buf[512];
  potus_last_error(buf, (int)sizeof buf);
  Rf_error("libpotus_hmc error %d: %s", status, buf);
}

SEXP potus_call_version(void) { return Rf_mkString(potus_version()); }

/* rstan::extract(fit, pars)[[1]] before its dims are set: a numeric matrix [draws, col_end - col_begin], chains merged chain after chain
 * (handles: integer vector, the fit's handles in chain order; col_begin / col_end: 0-based positions in the CmdStan row). */
SEXP potus_call_extract(SEXP handles, SEXP col_begin, SEXP col_end) {
  if (!Rf_isInteger(handles) || LENGTH(handles) < 1) Rf_error("potus_call_extract: handles must be a non-empty integer vector");
  const int cb = Rf_asInteger(col_begin), ce = Rf_asInteger(col_end);
  long long rows = 0;
  int status = potus_extract_matrix(INTEGER(handles), LENGTH(handles), cb, ce, NULL, 0, &rows);
  if (status) potus_call_fail(status);
  if (rows > 2147483647LL) Rf_error("potus_call_extract: %lld draws exceed the rows of an R matrix", rows);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)rows, ce - cb));       /* the ONE allocation; its length may exceed 2^31 - 1 */
  status = potus_extract_matrix(INTEGER(handles), LENGTH(handles), cb, ce, REAL(out), rows, NULL);
  UNPROTECT(1);
  if (status) potus_call_fail(status);
  return out;
}

/* Rank-normalised split R-hat and bulk ESS of columns [col_begin, col_end) over the pooled chains of the handles, computed on the device:
 * a numeric matrix [col_end - col_begin, 2] (rhat, ess_bulk). */
SEXP potus_call_diagnostics(SEXP handles, SEXP col_begin, SEXP col_end) {
  if (!Rf_isInteger(handles) || LENGTH(handles) < 1) Rf_error("potus_call_diagnostics: handles must be a non-empty integer vector");
  const int cb = Rf_asInteger(col_begin), ce = Rf_asInteger(col_end);
  if (ce <= cb) Rf_error("potus_call_diagnostics: empty column range");
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, ce - cb, 2));
  const int status = potus_diagnostics(INTEGER(handles), LENGTH(handles), cb, ce, REAL(out), REAL(out) + (ce - cb));
  UNPROTECT(1);
  if (status) potus_call_fail(status);
  return out;
}

/* the seven sampler columns of every saved draw (lp__, accept_stat__, stepsize__, treedepth__, n_leapfrog__, divergent__, energy__): [draws, 7] */
SEXP potus_call_sampler_params(SEXP handles) {
  SEXP b = PROTECT(Rf_ScalarInteger(0)), e = PROTECT(Rf_ScalarInteger(POTUS_N_SAMPLER_COLS));
  SEXP out = potus_call_extract(handles, b, e);
  UNPROTECT(2);
  return out;
}
