/* TEST INFRASTRUCTURE -- a 60-line stand-in for R's C API, just enough to COMPILE R/src/potus_call.c without R and to drive its entry points from
 * ctypes (tests/test_abi.py, tests/test_gpu_boundary.py): R is not installed in the image this repository is built in.  It is not a product file and
 * nothing under us_potus_model_amd/ or R/ includes it; with a real R the wrapper is built against R's own headers (R CMD SHLIB). */
#ifndef POTUS_R_STUB_RINTERNALS_H
#define POTUS_R_STUB_RINTERNALS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct stub_sexp { int type; long long length; int nrow, ncol; void *data; } *SEXP;
typedef long long R_xlen_t;
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
extern SEXP R_NilValue;
SEXP Rf_allocVector(int type, R_xlen_t n);
SEXP Rf_allocMatrix(int type, int nrow, int ncol);
SEXP Rf_ScalarInteger(int v);
SEXP Rf_mkString(const char *s);
int Rf_isInteger(SEXP x);
int Rf_asInteger(SEXP x);
int *INTEGER(SEXP x);
double *REAL(SEXP x);
int LENGTH(SEXP x);
R_xlen_t XLENGTH(SEXP x);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
void Rf_error(const char *fmt, ...) __attribute__((noreturn));
/* the harness side (what an R session would do around a .Call): build arguments, call under a handler that turns Rf_error into a message, inspect, free */
SEXP stub_int_vector(const int *v, int n);
SEXP stub_call0(SEXP (*fn)(void));
SEXP stub_call1(SEXP (*fn)(SEXP), SEXP a);
SEXP stub_call3(SEXP (*fn)(SEXP, SEXP, SEXP), SEXP a, SEXP b, SEXP c);
const char *stub_last_error(void);          /* "" when the last stub_call succeeded */
int stub_protect_depth(void);               /* PROTECTs minus UNPROTECTs so far: a wrapper must return with it unchanged */
long long stub_live_bytes(void);            /* bytes of vectors allocated and not yet released */
const char *stub_string(SEXP x);
void stub_release_all(void);
#ifdef __cplusplus
}
#endif
#endif
