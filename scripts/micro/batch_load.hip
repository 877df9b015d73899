// Micro-benchmark (development aid): latency of a batch of N outstanding 16-byte-per-lane buffer loads (one
// wave, 51 active lanes, rows 5.5 KB apart, written earlier by another workgroup) for several cache policies.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define ROWB 5488u

template <int N, int AUX>
__global__ __launch_bounds__(64) void k_batch(double *buf, unsigned *flag, long long *cyc, int iters) {
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1u << 20, 0x00020000);
  const int lane = threadIdx.x;
  if (blockIdx.x == 8) {          // writer (same XCD as block 0)
    for (int it = 1; it <= iters; it++) {
      for (int u = 0; u < 16; u++) {
        u32x4 w = {(unsigned)it, (unsigned)u, (unsigned)it, 7u};
        __builtin_amdgcn_raw_buffer_store_b128(w, r, lane < 51 ? 16u * lane : 0xFFFFFF00u, u * ROWB, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(flag, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(flag + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(2);
    }
  } else if (blockIdx.x == 0) {   // reader
    long long tot = 0;
    unsigned acc = 0;
    for (int it = 1; it <= iters; it++) {
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_s_sleep(20);
      const long long t0 = clock64();
      u32x4 w[N];
#pragma unroll
      for (int u = 0; u < N; u++) w[u] = __builtin_amdgcn_raw_buffer_load_b128(r, lane < 51 ? 16u * lane : 0xFFFFFF00u, u * ROWB, AUX | (int)0x80000000 * 0);
#pragma unroll
      for (int u = 0; u < N; u++) acc += w[u][2];
      asm volatile("" : "+v"(acc));
      const long long t1 = clock64();
      tot += t1 - t0;
      if (lane == 0) __hip_atomic_store(flag + 64, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) { cyc[0] = tot / iters; cyc[1] = acc; }
  }
}
template <int N, int AUX> void run(const char *nm) {
  double *buf; unsigned *flag; long long *cyc;
  hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20); hipMalloc(&flag, 1024); hipMemset(flag, 0, 1024); hipMalloc(&cyc, 16);
  hipLaunchKernelGGL((k_batch<N, AUX>), dim3(16), dim3(64), 0, 0, buf, flag, cyc, 300);
  hipDeviceSynchronize();
  long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-10s N=%2d: %lld cycles per batch\n", nm, N, c[0]);
  hipFree(buf); hipFree(flag); hipFree(cyc);
}
int main() {
  run<1, 16>("sc1"); run<2, 16>("sc1"); run<4, 16>("sc1"); run<8, 16>("sc1"); run<15, 16>("sc1");
  run<1, 1>("sc0"); run<15, 1>("sc0"); run<1, 0>("plain"); run<15, 0>("plain"); run<15, 17>("sc0 sc1"); run<15, 2>("nt");
  return 0;
}
