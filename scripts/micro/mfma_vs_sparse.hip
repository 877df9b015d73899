// mfma_vs_sparse.hip -- the measurement behind DESIGN.md's "fp64 MFMA for L_W x C: not built".
//
// north_star names "MFMA only for the dense Cholesky-factor x latent-matrix product": mu_b = L_W (51 x 51, padded to
// 64 x 64) times the suffix sums C (64 x T).  The sampler never forms that product: mu_b is needed only at the polled
// (state, day) cells, one 51-term dot each.  This program times both forms of the FORWARD product on one workgroup of
// 512 threads (every compute unit busy with a copy, as in the one-workgroup-per-chain kernels), operands in LDS:
//   dense   64 x 64 x Tc on v_mfma_f64_16x16x4_f64 (Tc = a 240-day chunk: what fits LDS beside the factor; T = 600 is 2.5 chunks)
//   sparse  one thread per poll: dot of L_W[s, :] with C[:, t] (the inner loop of phase C)
// for the 2016 shape (1 619 polls over 254 days -> 1 530 per chunk) and the configs[4] shape (10 000 polls over 600 days -> 4 000 per chunk).
//
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_vs_sparse.hip -o /tmp/mfma_vs_sparse && /tmp/mfma_vs_sparse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define S 51
#define SP 65          // LDS row stride of the 64 x 64 factor
#define TC 240         // days per chunk: 64 x 65 + 64 x 241 doubles = 157 KB of LDS
#define TP 241         // LDS row stride of C[k][t]

__global__ __launch_bounds__(512) void k_dense(const double *Lw, const double *Cin, double *sink, long long *cycles, int reps) {
  extern __shared__ double lds[];
  double *L = lds, *C = lds + 64 * SP;                       // L[m][k], C[k][t]
  for (int i = threadIdx.x; i < 64 * 64; i += 512) L[(i >> 6) * SP + (i & 63)] = Lw[i];
  for (int i = threadIdx.x; i < 64 * TC; i += 512) C[(i / TC) * TP + (i % TC)] = Cin[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, m = lane & 15, kk = lane >> 4;
  double acc_sink = 0.0;
  const long long t0 = clock64();
  for (int r = 0; r < reps; r++) {
    // output 64 x 240 = 4 x 15 tiles of 16 x 16; wave w takes day blocks w and w + 8 for all four state blocks
    for (int db = w; db < TC / 16; db += 8) {
      d4 acc[4];
      for (int sb = 0; sb < 4; sb++) acc[sb] = d4{0, 0, 0, 0};
      for (int k0 = 0; k0 < 64; k0 += 4) {
        const double b = C[(k0 + kk) * TP + 16 * db + m];     // B[k][n] = C[k][day]
#pragma unroll
        for (int sb = 0; sb < 4; sb++) acc[sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(L[(16 * sb + m) * SP + k0 + kk], b, acc[sb], 0, 0, 0);
      }
      for (int sb = 0; sb < 4; sb++) acc_sink += acc[sb][0] + acc[sb][1] + acc[sb][2] + acc[sb][3];
    }
    __syncthreads();
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 512 + threadIdx.x] = acc_sink;
}

__global__ __launch_bounds__(512) void k_sparse(const double *Lw, const double *Cin, const int *ps, const int *pt, int npoll, double *sink, long long *cycles,
                                                int reps) {
  extern __shared__ double lds[];
  double *L = lds, *C = lds + 64 * SP;
  for (int i = threadIdx.x; i < 64 * 64; i += 512) L[(i >> 6) * SP + (i & 63)] = Lw[i];
  for (int i = threadIdx.x; i < 64 * TC; i += 512) C[(i / TC) * TP + (i % TC)] = Cin[i];
  __syncthreads();
  double acc_sink = 0.0;
  const long long t0 = clock64();
  for (int r = 0; r < reps; r++) {
    for (int i = threadIdx.x; i < npoll; i += 512) {
      const double *L0 = L + ps[i] * SP, *C0 = C + pt[i];
      double a0 = 0.0, a1 = 0.0;
      for (int k0 = 0; k0 < 48; k0 += 16) {
        double l[16], c[16];
#pragma unroll
        for (int j = 0; j < 16; j++) { l[j] = L0[k0 + j]; c[j] = C0[(k0 + j) * TP]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; j += 2) { a0 += l[j] * c[j]; a1 += l[j + 1] * c[j + 1]; }
      }
      for (int k = 48; k < S; k++) a0 += L0[k] * C0[k * TP];
      acc_sink += a0 + a1;
    }
    __syncthreads();
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 512 + threadIdx.x] = acc_sink;
}

int main() {
  const int grid = 256, reps = 200;
  std::vector<double> Lw(64 * 64, 0.0), C(64 * TC);
  for (int i = 0; i < S; i++) for (int k = 0; k <= i; k++) Lw[i * 64 + k] = 0.01 * (1 + (i * 7 + k) % 5);
  for (size_t i = 0; i < C.size(); i++) C[i] = 0.001 * (double)(i % 97);
  double *dL, *dC, *dsink; long long *dcyc; int *dps, *dpt;
  hipMalloc(&dL, Lw.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dsink, grid * 512 * 8); hipMalloc(&dcyc, grid * 8);
  hipMalloc(&dps, 16384 * 4); hipMalloc(&dpt, 16384 * 4);
  hipMemcpy(dL, Lw.data(), Lw.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
  const size_t lds = (64 * SP + 64 * TP) * 8;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_sparse), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<long long> cyc(grid);
  auto avg = [&]() { hipMemcpy(cyc.data(), dcyc, grid * 8, hipMemcpyDeviceToHost); double s = 0; for (auto c : cyc) s += (double)c; return s / grid / reps; };
  hipLaunchKernelGGL(k_dense, dim3(grid), dim3(512), lds, 0, dL, dC, dsink, dcyc, reps);
  hipDeviceSynchronize();
  const double dense = avg();
  printf("dense  64 x 64 x %d days on v_mfma_f64_16x16x4_f64: %.0f cycles per workgroup pass (%.1f GFLOP/s per CU at 2.4 GHz)\n", TC, dense,
         2.0 * 64 * 64 * TC / dense * 2.4);
  for (int npoll : {1530, 4000}) {
    std::vector<int> ps(npoll), pt(npoll);
    for (int i = 0; i < npoll; i++) { ps[i] = (i * 37) % (S + 1) % 64; pt[i] = (i * 101) % TC; }
    hipMemcpy(dps, ps.data(), npoll * 4, hipMemcpyHostToDevice); hipMemcpy(dpt, pt.data(), npoll * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sparse, dim3(grid), dim3(512), lds, 0, dL, dC, dps, dpt, npoll, dsink, dcyc, reps);
    hipDeviceSynchronize();
    const double sp = avg();
    printf("sparse %5d polls x 51-term dots out of LDS:            %.0f cycles per workgroup pass  -> dense / sparse = %.1f x (flops %.1f x)\n", npoll, sp,
           dense / sp, 64.0 * 64 * TC / (npoll * 51.0));
  }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
