/* TEST INFRASTRUCTURE: the implementation behind tests/r_stub/Rinternals.h (not a product file). */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "Rinternals.h"

static struct stub_sexp nil_ = {0, 0, 0, 0, NULL};
SEXP R_NilValue = &nil_;
static SEXP all_[4096];
static int n_all_ = 0, depth_ = 0, armed_ = 0;
static long long live_ = 0;
static char err_[1024];
static jmp_buf jb_;

static SEXP make(int type, long long n, size_t elem) {
  SEXP s = (SEXP)calloc(1, sizeof *s);
  s->type = type; s->length = n; s->data = calloc((size_t)(n > 0 ? n : 1), elem);
  if (!s->data) { fprintf(stderr, "r_stub: out of memory for %lld elements\n", n); abort(); }
  live_ += n * (long long)elem;
  if (n_all_ < 4096) all_[n_all_++] = s;
  return s;
}
SEXP Rf_allocVector(int type, R_xlen_t n) { return make(type, n, type == INTSXP ? sizeof(int) : type == REALSXP ? sizeof(double) : sizeof(char *)); }
SEXP Rf_allocMatrix(int type, int nrow, int ncol) { SEXP s = Rf_allocVector(type, (R_xlen_t)nrow * ncol); s->nrow = nrow; s->ncol = ncol; return s; }
SEXP Rf_ScalarInteger(int v) { SEXP s = Rf_allocVector(INTSXP, 1); ((int *)s->data)[0] = v; return s; }
SEXP Rf_mkString(const char *str) { SEXP s = make(STRSXP, 1, sizeof(char *)); ((char **)s->data)[0] = strdup(str); return s; }
int Rf_isInteger(SEXP x) { return x && x->type == INTSXP; }
int Rf_asInteger(SEXP x) { return x->type == INTSXP ? ((int *)x->data)[0] : (int)((double *)x->data)[0]; }
int *INTEGER(SEXP x) { return (int *)x->data; }
double *REAL(SEXP x) { return (double *)x->data; }
int LENGTH(SEXP x) { return (int)x->length; }
R_xlen_t XLENGTH(SEXP x) { return x->length; }
SEXP Rf_protect(SEXP x) { depth_++; return x; }
void Rf_unprotect(int n) { depth_ -= n; }
void Rf_error(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(err_, sizeof err_, fmt, ap); va_end(ap);
  if (!armed_) { fprintf(stderr, "r_stub: Rf_error outside stub_call: %s\n", err_); abort(); }
  longjmp(jb_, 1);
}
SEXP stub_int_vector(const int *v, int n) { SEXP s = Rf_allocVector(INTSXP, n); memcpy(s->data, v, (size_t)n * sizeof(int)); return s; }
#define GUARDED(call) do { err_[0] = 0; armed_ = 1; if (setjmp(jb_)) { armed_ = 0; depth_ = d0; return R_NilValue; } SEXP r_ = (call); armed_ = 0; return r_; } while (0)
SEXP stub_call0(SEXP (*fn)(void)) { const int d0 = depth_; GUARDED(fn()); }
SEXP stub_call1(SEXP (*fn)(SEXP), SEXP a) { const int d0 = depth_; GUARDED(fn(a)); }
SEXP stub_call3(SEXP (*fn)(SEXP, SEXP, SEXP), SEXP a, SEXP b, SEXP c) { const int d0 = depth_; GUARDED(fn(a, b, c)); }
const char *stub_last_error(void) { return err_; }
int stub_protect_depth(void) { return depth_; }
long long stub_live_bytes(void) { return live_; }
const char *stub_string(SEXP x) { return x && x->type == STRSXP ? ((char **)x->data)[0] : ""; }
void stub_release_all(void) {
  for (int i = 0; i < n_all_; i++) { if (all_[i]->type == STRSXP) free(((char **)all_[i]->data)[0]); free(all_[i]->data); free(all_[i]); }
  n_all_ = 0; live_ = 0;
}
