// potus_diag.hpp -- rank-normalised split R-hat and bulk ESS of pooled chains, on the device.
//
// The reference never looks at a sampler diagnostic (scripts/model/final_2016.R:543-556 goes from read_stan_csv straight to
// extract); BASELINE.json's ESS/s metric and the all-gather "to pool draws for R-hat / ESS" (SURVEY.md sections 7.1-6, 8d, 8e, 8f-4) make
// the cross-chain diagnostics ours.  Definitions: Vehtari, Gelman, Simpson, Carpenter, Buerkner (2021), exactly as
// us_potus_model_amd/diagnostics.py states them in numpy (the CPU restatement the tests compare with):
//   split every chain in halves; rank-normalise the pooled draws (ties in the order of the split array, z = Phi^-1((r - 3/8) / (N + 1/4)));
//   R-hat = the larger of the classic R-hat of z and of the rank-normalised folded draws |x - median|;
//   bulk ESS = N / tau of z, tau from Geyer's initial monotone sequence over the chain-averaged autocorrelations.
// One workgroup per column (a column = one quantity of the output row: lp__, mu_b[s, t], ...):
//   1. k_dg_transpose turns the gathered block [draw][chain][column] into columns [column][chain][draw];
//   2. k_dg_column: the split draws of the column are bitonic-sorted as (value, position) pairs in LDS in runs of DG_RUN (one run
//      holds the reference's 6 x 500 and BASELINE configs[1]'s 8 x 1000; the 64 x 1000 of configs[2] take eight, which then go to a
//      scratch in global memory); a draw's rank = the number of smaller pairs, by binary search in the sorted run(s); the normal scores go to a
//      scratch row in the split order; per-chain moments and the autocovariances are summed in a fixed order (lane-strided,
//      then a DPP tree): same draws, same bytes.  The autocovariances are formed lag by lag only as far as Geyer's
//      sequence reads them (64 lags at a time): a few hundred thousand products per column, not an FFT.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "potus_dpp.hpp"
#include "potus_summary.hpp"

#define DG_RUN 8192           // (key, split index) pairs sorted in LDS at a time: 96 KB of dynamic LDS -- beyond the 64 KB of other parts: gfx950 (160 KB per
                              // workgroup) only, like the rest of this library; the host's hipFuncSetAttribute call fails with a message elsewhere
#define DG_THREADS 512
#define DG_MAXCH 1024         // split chains (2 x pooled chains): up to 512 pooled chains -- two full GPUs of one-workgroup chains, or the 64 / 96 / 128 of BASELINE configs[2..4]
#define DG_LAGS 64            // autocorrelations formed per round of Geyer's sequence

// block [nd][C][NC] (row = one draw of one chain) -> cols [NC][Ctot][nd], the block's chains at offset coff (several handles pooled)
__global__ __launch_bounds__(256) void k_dg_transpose(const double *in, double *out, long long nd, int C, int NC, int Ctot, int coff) {
  __shared__ double tile[64][65];
  const int c = blockIdx.z;
  const long long d0 = (long long)blockIdx.y * 64;
  const int c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  for (int r = ty; r < 64; r += 4) {
    const long long d = d0 + r;
    const int col = c0 + tx;
    tile[r][tx] = (d < nd && col < NC) ? in[((size_t)d * C + c) * NC + col] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int col = c0 + r;
    const long long d = d0 + tx;
    if (col < NC && d < nd) out[((size_t)col * Ctot + coff + c) * nd + d] = tile[tx][r];
  }
}

struct DgParams {
  const double *cols;       // [NC][C][n]
  double *zbuf;             // [grid][2][N] normal scores of the split draws (bulk, folded), split order
  unsigned long long *rkey; // [grid][N] sorted runs: keys ...
  unsigned *ridx;           // [grid][N] ... and the split index each key came from (only when N > DG_RUN)
  double *rhat, *ess;       // [NC]
  long long n;              // draws per chain
  int C, NC;
};

// value j of the split array (j = chain' * h + i', chain' < C: first halves, else second halves); -0.0 counts as 0.0, as it does for numpy's sort
__device__ __forceinline__ double dg_split_value(const double *x, long long n, int C, long long h, long long j) {
  const long long cp = j / h, i = j - cp * h;
  const int c = (int)(cp < C ? cp : cp - C);
  return x[(size_t)c * n + (cp < C ? i : n - h + i)] + 0.0;
}

// normcdfinv out of line: inlined into the two ranking loops its table of coefficients is hoisted out of them and held in registers across the
// whole column loop (217 spilled vector registers in round 4's build)
__device__ __noinline__ double dg_normal_score(double p) { return normcdfinv(p); }

// A draw's rank among the pooled split draws, ties in the order of the split array (numpy: argsort(argsort(x, stable), stable)) = 1 + the
// number of (key, split index) pairs that are lexicographically smaller.  The pairs are bitonic-sorted in LDS in runs of DG_RUN;
// one run stays there, several go to global memory; the count is a binary search per run.  No scan over tied draws: a sampler
// that keeps its point for a while, or a rounded quantity, costs nothing extra.
__device__ __forceinline__ bool dg_less(unsigned long long ka, unsigned ia, unsigned long long kb, unsigned ib) { return ka < kb || (ka == kb && ia < ib); }
template <class K, class I>
__device__ __forceinline__ int dg_count_less(const K *keys, const I *idx, int n, unsigned long long key, unsigned id) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (dg_less(keys[mid], idx[mid], key, id)) lo = mid + 1; else hi = mid; }
  return lo;
}
template <class F>
__device__ __forceinline__ void dg_sort_runs(F val, long long N, unsigned long long *xk, unsigned *xi, unsigned long long *rkey, unsigned *ridx) {
  const int tid = threadIdx.x;
  const bool multi = N > DG_RUN;
  for (long long r0 = 0; r0 < N; r0 += DG_RUN) {
    const int nn = (int)(N - r0 < DG_RUN ? N - r0 : DG_RUN);
    int npad = 1;
    while (npad < nn) npad <<= 1;
    for (int d = tid; d < npad; d += DG_THREADS) { xk[d] = d < nn ? ps_key(val(r0 + d)) : ~0ull; xi[d] = d < nn ? (unsigned)(r0 + d) : 0xffffffffu; }   // padding sorts to the end
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < npad; i += DG_THREADS) {
          const int l = i ^ j;
          if (l > i) {
            const unsigned long long a = xk[i], b = xk[l];
            const unsigned ia = xi[i], ib = xi[l];
            const bool up = (i & k) == 0;
            if (dg_less(b, ib, a, ia) == up) { xk[i] = b; xk[l] = a; xi[i] = ib; xi[l] = ia; }
          }
        }
        __syncthreads();
      }
    if (multi) {
      for (int d = tid; d < nn; d += DG_THREADS) { rkey[r0 + d] = xk[d]; ridx[r0 + d] = xi[d]; }
      __syncthreads();
    }
  }
  if (multi) __threadfence_block();
}
__device__ __forceinline__ long long dg_rank(double v, long long j, long long N, const unsigned long long *xk, const unsigned *xi, const unsigned long long *rkey, const unsigned *ridx) {
  const unsigned long long key = ps_key(v);
  if (N <= DG_RUN) return 1 + dg_count_less(xk, xi, (int)N, key, (unsigned)j);
  long long r = 1;
  for (long long r0 = 0; r0 < N; r0 += DG_RUN) r += dg_count_less(rkey + r0, ridx + r0, (int)(N - r0 < DG_RUN ? N - r0 : DG_RUN), key, (unsigned)j);
  return r;
}
// k-th smallest key (0-based) of the union of the sorted runs in global memory
__device__ __forceinline__ unsigned long long dg_select(const unsigned long long *rkey, long long N, long long k) {
  unsigned long long lo = 0ull, hi = ~0ull;
  while (lo < hi) {
    const unsigned long long mid = lo + ((hi - lo) >> 1);
    long long cnt = 0;
    for (long long r0 = 0; r0 < N; r0 += DG_RUN) {
      const int nn = (int)(N - r0 < DG_RUN ? N - r0 : DG_RUN);
      int a = 0, b = nn;
      while (a < b) { const int m = (a + b) >> 1; if (rkey[r0 + m] <= mid) a = m + 1; else b = m; }
      cnt += a;
    }
    if (cnt >= k + 1) hi = mid; else lo = mid + 1;
  }
  return lo;
}

__global__ __launch_bounds__(DG_THREADS) void k_dg_column(DgParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long xk[];   // min(npad, DG_RUN) keys, then as many split indices
  __shared__ double cmean[DG_MAXCH], cvar[DG_MAXCH], rho[DG_LAGS];
  __shared__ double sc[8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long n = P.n, h = n / 2;
  const int C = P.C, C2 = 2 * C;
  const long long N = (long long)C2 * h;
  double *z = P.zbuf + (size_t)blockIdx.x * 2 * (size_t)N, *zf = z + N;
  unsigned long long *rkey = P.rkey ? P.rkey + (size_t)blockIdx.x * (size_t)N : nullptr;
  unsigned *ridx = P.ridx ? P.ridx + (size_t)blockIdx.x * (size_t)N : nullptr;
  int npad_all = 1;
  while (npad_all < N && npad_all < DG_RUN) npad_all <<= 1;
  unsigned *xi = (unsigned *)(xk + npad_all);
  for (int col = blockIdx.x; col < P.NC; col += gridDim.x) {
    const double *x = P.cols + (size_t)col * C * n;
    if (h < 2) { if (tid == 0) { P.rhat[col] = NAN; P.ess[col] = NAN; } continue; }
    auto val = [&](long long j) { return dg_split_value(x, n, C, h, j); };
    {
      // a NaN or infinite draw makes median, folded values and ranks meaningless (and a NaN with an all-ones payload has the key of the
      // sort's padding): the column's diagnostics are NaN, as numpy's propagate it
      int bad = 0;
      for (long long j = tid; j < N; j += DG_THREADS) bad |= !isfinite(val(j));
      if (__syncthreads_or(bad)) { if (tid == 0) { P.rhat[col] = NAN; P.ess[col] = NAN; } continue; }
    }
    // ---- bulk: ranks of the split draws
    dg_sort_runs(val, N, xk, xi, rkey, ridx);
    if (tid == 0) {                                    // the median of the split draws (numpy: mean of the two middle ones)
      const long long k1 = (N - 1) / 2, k2 = N / 2;
      const double a = ps_unkey(N <= DG_RUN ? xk[k1] : dg_select(rkey, N, k1)), b = ps_unkey(N <= DG_RUN ? xk[k2] : dg_select(rkey, N, k2));
      sc[0] = 0.5 * (a + b);
    }
    for (long long j = tid; j < N; j += DG_THREADS) {
      const double v = val(j);
      const long long r = dg_rank(v, j, N, xk, xi, rkey, ridx);
      z[j] = dg_normal_score(((double)r - 0.375) / ((double)N + 0.25));
    }
    __syncthreads();
    const double med = sc[0];
    __syncthreads();
    // ---- folded: ranks of |x - median|
    auto fval = [&](long long j) { return fabs(dg_split_value(x, n, C, h, j) - med) + 0.0; };
    dg_sort_runs(fval, N, xk, xi, rkey, ridx);
    for (long long j = tid; j < N; j += DG_THREADS) {
      const double v = fval(j);
      const long long r = dg_rank(v, j, N, xk, xi, rkey, ridx);
      zf[j] = dg_normal_score(((double)r - 0.375) / ((double)N + 0.25));
    }
    __threadfence_block();
    __syncthreads();
    // ---- classic R-hat of both (one wave per split chain, two passes: mean, then variance)
    double rh[2];
    for (int pass = 0; pass < 2; pass++) {
      const double *zz = pass == 0 ? zf : z;          // bulk last: its chain means and variances stay for the ESS
      for (int cp = w; cp < C2; cp += DG_THREADS / 64) {
        double s = 0.0;
        for (long long i = lane; i < h; i += 64) s += zz[(size_t)cp * h + i];
        const double m = dpp_wave_sum(s) / (double)h;
        double q = 0.0;
        for (long long i = lane; i < h; i += 64) { const double d = zz[(size_t)cp * h + i] - m; q += d * d; }
        const double v = dpp_wave_sum(q) / (double)(h - 1);
        if (lane == 0) { cmean[cp] = m; cvar[cp] = v; }
      }
      __syncthreads();
      if (tid == 0) {
        double wv = 0.0, mm = 0.0;
        for (int cp = 0; cp < C2; cp++) { wv += cvar[cp]; mm += cmean[cp]; }
        wv /= C2; mm /= C2;
        double b = 0.0;
        for (int cp = 0; cp < C2; cp++) b += (cmean[cp] - mm) * (cmean[cp] - mm);
        const double bv = C2 > 1 ? b / (C2 - 1) : 0.0;                       // variance of the chain means (ddof = 1)
        sc[1 + pass] = wv == 0.0 ? NAN : sqrt(((double)(h - 1) / (double)h * wv + bv) / wv);   // rhat_basic: B / n = var of the means
        sc[3] = wv; sc[4] = bv;
      }
      __syncthreads();
      rh[pass] = sc[1 + pass];
    }
    // ---- bulk ESS (ess_basic of z): centre the chains, autocovariances lag by lag as far as Geyer's sequence needs them
    for (long long j = tid; j < N; j += DG_THREADS) z[j] -= cmean[j / h];
    __threadfence_block();
    __syncthreads();
    // chain_var = acov[:, 0] * h / (h - 1) = cvar; mean_var = mean(cvar); var_plus = mean_var (h - 1) / h + var(means)
    const double mean_var = sc[3], var_plus = mean_var * (double)(h - 1) / (double)h + (C2 > 1 ? sc[4] : 0.0);
    double ess = NAN;
    if (h >= 4 && isfinite(var_plus) && var_plus > 0.0) {
      if (tid == 0) { sc[5] = 0.0; sc[6] = INFINITY; sc[7] = 0.0; }   // tau so far, previous pair, done flag
      __syncthreads();
      for (long long t0 = 0; t0 < h; t0 += DG_LAGS) {
        for (int lg = w; lg < DG_LAGS; lg += DG_THREADS / 64) {
          const long long t = t0 + lg;
          double s = 0.0;
          if (t < h)
            for (int cp = 0; cp < C2; cp++) {
              const double *zc = z + (size_t)cp * h;
              double a = 0.0;
              for (long long i = lane; i + t < h; i += 64) a += zc[i] * zc[i + t];
              s += dpp_wave_sum(a) / (double)h;                                  // biased autocovariance of the chain
            }
          if (lane == 0) rho[lg] = t == 0 ? 1.0 : 1.0 - (mean_var - s / C2) / var_plus;
        }
        __syncthreads();
        if (tid == 0) {
          double tau = sc[5], prev = sc[6];
          bool done = false;
          for (int lg = 0; lg + 1 < DG_LAGS; lg += 2) {
            if (t0 + lg + 1 >= h) { done = true; break; }
            double pair = rho[lg] + rho[lg + 1];
            if (pair < 0) { done = true; break; }
            pair = fmin(pair, prev);
            tau += 2.0 * pair;
            prev = pair;
          }
          sc[5] = tau; sc[6] = prev; sc[7] = done ? 1.0 : 0.0;
        }
        __syncthreads();
        if (sc[7] != 0.0) break;
      }
      double tau = sc[5] - 1.0;
      tau = fmax(tau, 1.0 / log10((double)N));
      ess = (double)N / tau;
    }
    if (tid == 0) {
      P.rhat[col] = (isnan(rh[0]) && isnan(rh[1])) ? NAN : fmax(isnan(rh[0]) ? -INFINITY : rh[0], isnan(rh[1]) ? -INFINITY : rh[1]);   // nanmax(bulk, folded)
      P.ess[col] = ess;
    }
    __syncthreads();
  }
}
