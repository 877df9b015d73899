// Micro-benchmark (development aid): cost of one all-to-all exchange among the K workgroups of a
// "cluster" that cooperate on one chain.  Blocks b with the same (b % NCL) form a cluster, so with
// the observed round-robin placement (block b -> XCD b % 8) a cluster of NCL = 8 shares one L2.
// Protocol: sc1 (write-through) stores of the payload, per-wave drain, one relaxed agent-scope
// fetch_add on the cluster counter, one lane polls relaxed, payload read back with sc1 loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__global__ __launch_bounds__(512) void k_exchange(double *xbuf, unsigned *cnt, unsigned *tmo, int ncl, int K, int nval, int iters, double *out,
                                                  long long *cycles) {
  const int cl = blockIdx.x % ncl, m = blockIdx.x / ncl, tid = threadIdx.x;
  double *xb = xbuf + (size_t)cl * 2 * K * nval;
  unsigned *c = cnt + cl * 64;   // one counter per cluster, own cache line
  double acc = 0.0;
  __shared__ int fail;
  if (tid == 0) fail = 0;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 1; it <= iters; it++) {
    double *slot = xb + (size_t)(it & 1) * K * nval;
    for (int i = tid; i < nval; i += blockDim.x)
      __hip_atomic_store((u64 *)(slot + (size_t)m * nval + i), (u64)__double_as_longlong((double)(it + m) + 1e-3 * i + acc * 1e-9), RLX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(c, 1u, RLX);
      const unsigned want = (unsigned)it * (unsigned)K;
      unsigned spins = 0;
      while (__hip_atomic_load(c, RLX) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 4000000u) { fail = 1; __hip_atomic_store(tmo, 1u, RLX); break; }
      }
    }
    __syncthreads();
    if (fail) break;
    for (int i = tid; i < K * nval; i += blockDim.x)
      acc += __longlong_as_double((long long)__hip_atomic_load((u64 *)(slot + i), RLX));
  }
  const long long t1 = clock64();
  if (tid == 0 && m == 0) cycles[cl] = t1 - t0;
  out[blockIdx.x * blockDim.x + tid] = acc;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  for (int ncl : {8, 1}) for (int K : {2, 4, 8, 16, 32}) for (int nval : {64, 512}) {
    if (ncl * K > 256) continue;
    double *xbuf, *out; unsigned *cnt, *tmo; long long *cyc;
    hipMalloc(&xbuf, sizeof(double) * ncl * 2 * K * nval); hipMalloc(&out, sizeof(double) * ncl * K * 512);
    hipMalloc(&cnt, sizeof(unsigned) * ncl * 64); hipMalloc(&tmo, 4); hipMalloc(&cyc, sizeof(long long) * ncl);
    hipMemset(cnt, 0, sizeof(unsigned) * ncl * 64); hipMemset(tmo, 0, 4); hipMemset(xbuf, 0, sizeof(double) * ncl * 2 * K * nval);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_exchange, dim3(ncl * K), dim3(512), 0, 0, xbuf, cnt, tmo, ncl, K, nval, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned t; hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost);
    std::vector<double> h(ncl * K * 512); hipMemcpy(h.data(), out, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    // check: every block of a cluster must have the same sum
    bool same = true; for (int cl = 0; cl < ncl; cl++) for (int m = 1; m < K; m++) for (int i = 0; i < 512; i++)
      if (h[(size_t)(m * ncl + cl) * 512 + i] != h[(size_t)cl * 512 + i]) same = false;
    printf("clusters=%d K=%2d nval=%3d: %.3f us per exchange (%s%s)\n", ncl, K, nval, 1e3 * ms / iters, t ? "TIMEOUT " : "", same ? "consistent" : "MISMATCH");
    hipFree(xbuf); hipFree(out); hipFree(cnt); hipFree(tmo); hipFree(cyc);
  }
  return 0;
}
