// potus_dense.hpp -- first building block of the dense metric (stan::mcmc::dense_e_metric, BASELINE configs[4]):
// y = M^-1 p for a batch of chains, each with its own D x D inverse metric (row-major, symmetric, in HBM).
//
// This is the one operation of the path that is bound by HBM bandwidth: 8 D^2 bytes per chain and leapfrog (position
// update, kinetic energy and the p-sharp vectors of the U-turn checks all need M^-1 p).  Not wired into the sampler
// yet; exported as a development entry point (potus_dense_matvec_probe, not part of include/potus_hmc.h) so that its
// results and its rate can be checked on their own.
//
// Layout of the work: one workgroup = 64 consecutive rows of one chain (8 waves x 8 rows); a wave streams a row with
// all 64 lanes (512 contiguous bytes per load instruction, eight loads in flight per lane), p sits in LDS in column
// tiles of 16 384 doubles (128 KB) so that D = 41 610 needs three tiles.  Sums run in a fixed order (per lane over the
// columns, DPP across the lanes): same bytes every time.  The full matrix is read although it is symmetric: using
// a_ij for both y_i and y_j needs cross-workgroup accumulation (atomics or a second pass) and would halve the traffic
// -- left for when the kernel is part of the sampler.
#pragma once
#include "potus_dpp.hpp"

#define PD_THREADS 512
#define PD_ROWS_PER_WAVE 8
#define PD_ROWS (PD_ROWS_PER_WAVE * (PD_THREADS / 64))
#define PD_TILE 16384
#define PD_UNR 8

__global__ __launch_bounds__(PD_THREADS) void k_dense_matvec(const double *__restrict__ Minv /*[chains][D][D]*/,
                                                             const double *__restrict__ p /*[chains][D]*/,
                                                             double *__restrict__ y /*[chains][D]*/, int D) {
  extern __shared__ __attribute__((aligned(16))) double pd_lds[];   // PD_TILE doubles
  const int chain = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * PD_ROWS + w * PD_ROWS_PER_WAVE;
  const double *A = Minv + (size_t)chain * D * D;
  const double *pc = p + (size_t)chain * D;
  double acc[PD_ROWS_PER_WAVE];
#pragma unroll
  for (int r = 0; r < PD_ROWS_PER_WAVE; r++) acc[r] = 0.0;
  for (int t0 = 0; t0 < D; t0 += PD_TILE) {
    const int tl = min(PD_TILE, D - t0);
    __syncthreads();
    for (int j = tid; j < tl; j += PD_THREADS) pd_lds[j] = pc[t0 + j];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PD_ROWS_PER_WAVE; r++) {
      const int row = row0 + r;
      if (row >= D) break;                         // wave-uniform
      const double *Ar = A + (size_t)row * D + t0;
      double s = 0.0;
      int j = lane;
      for (; j + 64 * (PD_UNR - 1) < tl; j += 64 * PD_UNR) {
        double a[PD_UNR];
#pragma unroll
        for (int u = 0; u < PD_UNR; u++) a[u] = __builtin_nontemporal_load(Ar + j + 64 * u);
#pragma unroll
        for (int u = 0; u < PD_UNR; u++) s += a[u] * pd_lds[j + 64 * u];
      }
      for (; j < tl; j += 64) s += __builtin_nontemporal_load(Ar + j) * pd_lds[j];
      acc[r] += s;
    }
  }
#pragma unroll
  for (int r = 0; r < PD_ROWS_PER_WAVE; r++) {
    const double tot = dpp_wave_sum(acc[r]);
    const int row = row0 + r;
    if (lane == 0 && row < D) y[(size_t)chain * D + row] = tot;
  }
}

// welford_covar_estimator::add_sample for a batch of chains: m2 += (q - mean_new) delta', delta = q - mean_old (both
// D-vectors prepared by the caller: a = q - mean_new, delta).  16 D^2 bytes per chain and warm-up iteration inside the
// adaptation windows; one workgroup per 8 rows, a thread keeps its column of delta in a register across the rows.
__global__ __launch_bounds__(256) void k_dense_welford(double *__restrict__ M2 /*[chains][D][D]*/, const double *__restrict__ a /*[chains][D]*/,
                                                       const double *__restrict__ delta /*[chains][D]*/, int D) {
  const int chain = blockIdx.y, row0 = blockIdx.x * 8;
  double *M = M2 + (size_t)chain * D * D;
  const double *ac = a + (size_t)chain * D, *dc = delta + (size_t)chain * D;
  double ar[8];
#pragma unroll
  for (int r = 0; r < 8; r++) ar[r] = row0 + r < D ? ac[row0 + r] : 0.0;
  for (int j = threadIdx.x; j < D; j += 256) {
    const double dj = dc[j];
#pragma unroll
    for (int r = 0; r < 8; r++)
      if (row0 + r < D) { double *e = M + (size_t)(row0 + r) * D + j; *e = *e + ar[r] * dj; }
  }
}

// fills a symmetric positive definite test matrix on the device: A[i][j] = exp(-|i-j|/50) * (1 + 0.1 c) + (i == j ? 1 : 0)
__global__ void k_dense_fill(double *Minv, int D, int chains) {
  const size_t n = (size_t)D * D;
  for (int c = 0; c < chains; c++)
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
      const int i = (int)(e / D), j = (int)(e % D);
      const int d = i > j ? i - j : j - i;
      Minv[(size_t)c * n + e] = exp(-(double)d / 50.0) * (1.0 + 0.1 * c) + (i == j ? 1.0 : 0.0);
    }
}
