#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static double rcp_approx(double d) { return (double)(1.0f / (float)d); }   /* stands for v_rcp_f64 (at least single precision) */
static inline double nr_recip(double d) {   /* 1/d to full precision, d in [1, 3] */
  double y = rcp_approx(d);
  double e = fma(-d, y, 1.0); y = fma(y, e, y);
  e = fma(-d, y, 1.0); y = fma(y, e, y);
  e = fma(-d, y, 1.0); y = fma(y, e, y);
  return y;
}
static inline double fx_div(double a, double d) { const double y = nr_recip(d); double q = a * y; const double r = fma(-d, q, a); return fma(r, y, q); }
/* exp(-a), a >= 0 */
static inline double fx_exp_neg(double a) {
  const double x = -fmin(a, 745.2);
  const double n = nearbyint(x * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, x);   /* ln2 hi (fdlibm) */
  r = fma(n, -1.90821492927058770002e-10, r);          /* ln2 lo */
  const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
  /* exp(r) = 1 + r + r^2 q(r), q = sum_{k=2}^{13} r^(k-2) / k! */
  const double c2 = 1.0/2, c3 = 1.0/6, c4 = 1.0/24, c5 = 1.0/120, c6 = 1.0/720, c7 = 1.0/5040, c8 = 1.0/40320, c9 = 1.0/362880, c10 = 1.0/3628800,
               c11 = 1.0/39916800, c12 = 1.0/479001600, c13 = 1.0/6227020800.0;
  const double q01 = fma(c3, r, c2), q23 = fma(c5, r, c4), q45 = fma(c7, r, c6), q67 = fma(c9, r, c8), q89 = fma(c11, r, c10), qab = fma(c13, r, c12);
  const double q03 = fma(q23, r2, q01), q47 = fma(q67, r2, q45), q8b = fma(qab, r2, q89);
  const double q = fma(q8b, r8, fma(q47, r4, q03));
  const double p = 1.0 + fma(r2, q, r);
  return ldexp(p, (int)n);
}
/* log1p(e), 0 <= e <= 1, and 1 / (1 + e) */
static inline void fx_log1p_recip(double e, double *l1, double *inv) {
  const double u = 1.0 + e, c = e - (u - 1.0);
  const double yu = nr_recip(u);
  *inv = yu;
  const int k = u > 1.4142135623730951;
  const double up = k ? 0.5 * u : u;
  const double f = up - 1.0, d = 2.0 + f;
  const double s = fx_div(f, d), z = s * s, z2 = z * z, z4 = z2 * z2;
  /* 2 atanh(s) = 2 s (1 + z/3 + z^2/5 + ... + z^11/23) */
  const double a1 = 2.0/3, a2 = 2.0/5, a3 = 2.0/7, a4 = 2.0/9, a5 = 2.0/11, a6 = 2.0/13, a7 = 2.0/15, a8 = 2.0/17, a9 = 2.0/19, a10 = 2.0/21, a11 = 2.0/23, a12 = 2.0/25;
  const double t01 = fma(a2, z, a1), t23 = fma(a4, z, a3), t45 = fma(a6, z, a5), t67 = fma(a8, z, a7), t89 = fma(a10, z, a9), tab = fma(a12, z, a11);
  const double t03 = fma(t23, z2, t01), t47 = fma(t67, z2, t45), t8b = fma(tab, z2, t89);
  const double Q = fma(t8b, z4 * z4, fma(t47, z4, t03));       /* sum_{i>=1} 2 z^(i-1) / (2i+1) */
  const double tail = fma(s * z, Q, c * yu);                   /* 2 s z Q/2... see below */
  /* log(u') = 2 s + s z Q ; log1p = k ln2 + log(u') + c/u */
  const double lo = tail + (k ? 1.90821492927058770002e-10 : 0.0);
  *l1 = (k ? 6.93147180369123816490e-01 : 0.0) + (2.0 * s + lo);
}
int main(void) {
  double worst_e = 0, worst_l = 0, worst_p = 0;
  srand(1);
  for (long it = 0; it < 20000000; it++) {
    double a;
    if (it % 3 == 0) a = 50.0 * rand() / RAND_MAX; else if (it % 3 == 1) a = 2.0 * rand() / RAND_MAX; else a = ldexp(1.0 + (double)rand() / RAND_MAX, -(rand() % 40));
    const double ex = fx_exp_neg(a), ref = exp(-a);
    const double ee = fabs(ex - ref) / ref / 2.220446049250313e-16;
    if (ee > worst_e) worst_e = ee;
    double l1, inv; fx_log1p_recip(ref, &l1, &inv);
    const double lr = log1p(ref), le = fabs(l1 - lr) / lr / 2.220446049250313e-16;
    if (le > worst_l) { worst_l = le; }
    const double pr = fabs(inv - 1.0 / (1.0 + ref)) * (1.0 + ref) / 2.220446049250313e-16;
    if (pr > worst_p) worst_p = pr;
  }
  printf("worst error in units of eps (relative): exp %.2f, log1p %.2f, 1/(1+e) %.2f\n", worst_e, worst_l, worst_p);
  return 0;
}
