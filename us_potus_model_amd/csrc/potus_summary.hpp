// potus_summary.hpp -- posterior summaries of predicted_score on the device.
//
// What the reference scripts compute from rstan::extract(out, "predicted_score") (final_2016.R:708-762 state and
// national vote intervals, :799-823 electoral-college simulation), without shipping draws x 12 954 doubles to R.
// Any number of pooled draws (BASELINE configs[2]: 64 chains x 1000 = 64 000) and any number of samplers
// (the chains of one posterior spread over several handles / GPUs):
//
//   1. k_write_array (potus_hmc.hip) produces predicted_score of every saved draw, [draw][T*S], cell = t + T*s;
//   2. k_ps_derived adds the two per-draw series the scripts build -- the state_weights-weighted national vote and the
//      Democratic electoral votes -- and k_ps_transpose turns everything into COLUMNS, [column][draw], so that
//      the draws of a cell are contiguous;
//   3. k_col_summary: one workgroup per column.  The column is cut into runs of at most PS_RUN = 16 384 draws; a run
//      is bitonic-sorted in LDS (128 KB).  One run (<= 16 384 draws, the reference's 6 x 500 and BASELINE configs[1]'s
//      8 x 1000): order statistics straight out of LDS.  Several runs: the sorted runs go to a per-workgroup scratch in
//      global memory and each needed order statistic is found by bisection on the 64-bit order-preserving key of a
//      double, counting `<= key` with a binary search per run -- exact, no second sort, no atomics.
//      Means and exceedance frequencies are summed in a fixed order (thread-strided over the sorted runs, then a
//      tree): same draws, same bytes.
// Quantiles are R's default (type 7).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PS_RUN 16384
#define PS_THREADS 512

// national[d][t] = sum_s w[s] ps[d][t + T s] (w normalised on the host: weighted.mean, final_2016.R:745);
// ev[d][t] = sum_s ev[s] 1[ps > 0.5] (final_2016.R:799-823).  Output rows are appended to the draw's row:
// full[d][T*S + t] and full[d][T*S + T + t], row length NC = T*S + 2T.
__global__ __launch_bounds__(256) void k_ps_derived(double *full /*[nd][NC]*/, long long nd, int T, int S, const double *w, const double *ev) {
  const int NC = T * S + 2 * T;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < nd * T; idx += (long long)gridDim.x * 256) {
    const long long d = idx / T;
    const int t = (int)(idx - d * T);
    double *row = full + d * NC;
    double a = 0.0, b = 0.0;
    for (int s = 0; s < S; s++) { const double x = row[t + T * s]; a += w[s] * x; b += x > 0.5 ? ev[s] : 0.0; }
    row[T * S + t] = a;
    row[T * S + T + t] = b;
  }
}

// [nd][NC] -> [NC][nd] through 64 x 64 LDS tiles
__global__ __launch_bounds__(256) void k_ps_transpose(const double *in, double *out, long long nd, int NC) {
  __shared__ double tile[64][65];
  const long long d0 = (long long)blockIdx.y * 64;
  const int c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  for (int r = ty; r < 64; r += 4) {
    const long long d = d0 + r;
    const int c = c0 + tx;
    tile[r][tx] = (d < nd && c < NC) ? in[d * NC + c] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r;
    const long long d = d0 + tx;
    if (c < NC && d < nd) out[(long long)c * nd + d] = tile[tx][r];
  }
}

__device__ __forceinline__ unsigned long long ps_key(double x) {   // order-preserving: a < b  <=>  key(a) < key(b)
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double ps_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __builtin_bit_cast(double, b);
}
// number of elements <= key in a sorted run
__device__ __forceinline__ int ps_upper(const double *run, int n, unsigned long long key) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (ps_key(run[mid]) <= key) lo = mid + 1; else hi = mid; }
  return lo;
}
// k-th smallest (0-based) of the union of the sorted runs of a column
__device__ __forceinline__ double ps_select(const double *runs, long long nd, long long k) {
  unsigned long long lo = 0ull, hi = ~0ull;
  while (lo < hi) {
    const unsigned long long mid = lo + ((hi - lo) >> 1);
    long long cnt = 0;
    for (long long r0 = 0; r0 < nd; r0 += PS_RUN) cnt += ps_upper(runs + r0, (int)(nd - r0 < PS_RUN ? nd - r0 : PS_RUN), mid);
    if (cnt >= k + 1) hi = mid; else lo = mid + 1;
  }
  return ps_unkey(lo);
}

// cols [NC][nd]; scratch [gridDim.x][nd] (only touched when nd > PS_RUN).
// out_state [T*S][4] = low, high, mean, P(> 0.5); out_natl [T][4] the same; out_ev [T][5] = mean, median, high, low, P(>= 270)
__global__ __launch_bounds__(PS_THREADS) void k_col_summary(const double *cols, double *scratch, long long nd, int T, int S,
                                                            double *out_state, double *out_natl, double *out_ev) {
  extern __shared__ __attribute__((aligned(16))) double xs[];   // min(npad, PS_RUN) doubles
  __shared__ double red[2][PS_THREADS];
  __shared__ double qv[6];
  const int tid = threadIdx.x, TS = T * S, NC = TS + 2 * T;
  const bool multi = nd > PS_RUN;
  double *runs = scratch + (size_t)blockIdx.x * (size_t)nd;
  for (int col = blockIdx.x; col < NC; col += gridDim.x) {
    const int kind = col < TS ? 0 : col < TS + T ? 1 : 2;
    const double thr = kind == 2 ? 270.0 : 0.5;
    const double *src = cols + (size_t)col * (size_t)nd;
    double sm = 0.0, ex = 0.0;
    for (long long r0 = 0; r0 < nd; r0 += PS_RUN) {
      const int n = (int)(nd - r0 < PS_RUN ? nd - r0 : PS_RUN);
      int npad = 1;
      while (npad < n) npad <<= 1;
      for (int d = tid; d < npad; d += PS_THREADS) xs[d] = d < n ? src[r0 + d] : INFINITY;   // padding sorts to the end
      __syncthreads();
      for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = tid; i < npad; i += PS_THREADS) {
            const int l = i ^ j;
            if (l > i) {
              const double a = xs[i], b = xs[l];
              const bool up = (i & k) == 0;
              if ((a > b) == up) { xs[i] = b; xs[l] = a; }
            }
          }
          __syncthreads();
        }
      for (int d = tid; d < n; d += PS_THREADS) {
        const double x = xs[d];
        sm += x;
        ex += kind == 2 ? (x >= thr ? 1.0 : 0.0) : (x > thr ? 1.0 : 0.0);
        if (multi) runs[r0 + d] = x;
      }
      __syncthreads();
    }
    red[0][tid] = sm; red[1][tid] = ex;
    __syncthreads();
    for (int off = PS_THREADS / 2; off > 0; off >>= 1) {
      if (tid < off) { red[0][tid] += red[0][tid + off]; red[1][tid] += red[1][tid + off]; }
      __syncthreads();
    }
    // order statistics x_(lo), x_(hi) of the three type-7 quantiles 0.025, 0.975, 0.5: threads 0..5
    if (tid < 6) {
      const double p = tid < 2 ? 0.025 : tid < 4 ? 0.975 : 0.5;
      const double h = (double)(nd - 1) * p;
      const long long lo = (long long)floor(h);
      const long long k = (tid & 1) ? (lo + 1 < nd ? lo + 1 : nd - 1) : lo;
      qv[tid] = multi ? ps_select(runs, nd, k) : xs[k];
    }
    __syncthreads();
    if (tid == 0) {
      const double mean = red[0][0] / (double)nd, prob = red[1][0] / (double)nd;
      double q3[3];
      for (int j = 0; j < 3; j++) {
        const double p = j == 0 ? 0.025 : j == 1 ? 0.975 : 0.5;
        const double h = (double)(nd - 1) * p;
        q3[j] = qv[2 * j] + (h - floor(h)) * (qv[2 * j + 1] - qv[2 * j]);
      }
      if (kind == 0) { double *o = out_state + (size_t)col * 4; o[0] = q3[0]; o[1] = q3[1]; o[2] = mean; o[3] = prob; }
      else if (kind == 1) { double *o = out_natl + (size_t)(col - TS) * 4; o[0] = q3[0]; o[1] = q3[1]; o[2] = mean; o[3] = prob; }
      else { double *o = out_ev + (size_t)(col - TS - T) * 5; o[0] = mean; o[1] = q3[2]; o[2] = q3[1]; o[3] = q3[0]; o[4] = prob; }
    }
    __syncthreads();
  }
}
