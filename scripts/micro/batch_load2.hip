// Micro-benchmark (development aid): one reader wave fetches 16 rows (51 lanes x 16 bytes each) written by 16 different
// workgroups of the same XCD (as the cluster's exchange does) or by a single one; sc1 loads, all in flight.
// prestore = 1 / 2: the reader issues one sc1 / plain store right before the loads (the wait for the loads then also
// waits for the store's acknowledgement: loads and stores share vmcnt on gfx9-class hardware).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define ROWB 3712u

template <int WRITERS, int AUX, int NLANES, int PRESTORE>
__global__ __launch_bounds__(64) void k_batch(double *buf, unsigned *flag, long long *cyc, int iters) {
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1u << 20, 0x00020000);
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  if (b % 8 != 0) return;                      // blocks 0, 8, 16, ... share an XCD
  const int id = b / 8;                        // 0 = reader, 1..16 writers
  if (id >= 1 && id <= WRITERS) {
    for (int it = 1; it <= iters; it++) {
      for (int u = id - 1; u < 16; u += WRITERS) {
        u32x4 w = {(unsigned)it, (unsigned)u, (unsigned)it, 7u};
        __builtin_amdgcn_raw_buffer_store_b128(w, r, lane < NLANES ? 16u * lane : 0xFFFFFF00u, u * ROWB, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(flag + 32 * id, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(2);
    }
  } else if (id == 0) {
    long long tot = 0;
    unsigned acc = 0;
    for (int it = 1; it <= iters; it++) {
      for (int wr = 1; wr <= WRITERS; wr++)
        while (__hip_atomic_load(flag + 32 * wr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_s_sleep(40);
      const long long t0 = clock64();
      if (PRESTORE) { u32x4 z = {(unsigned)it, 1u, 2u, 3u}; __builtin_amdgcn_raw_buffer_store_b128(z, r, lane < NLANES ? 16u * lane : 0xFFFFFF00u, 20 * ROWB, PRESTORE == 1 ? 16 : 0); }
      u32x4 w[16];
#pragma unroll
      for (int u = 0; u < 16; u++) w[u] = __builtin_amdgcn_raw_buffer_load_b128(r, lane < NLANES ? 16u * lane : 0xFFFFFF00u, u * ROWB, AUX);
#pragma unroll
      for (int u = 0; u < 16; u++) acc += w[u][2] == (unsigned)it;
      asm volatile("" : "+v"(acc));
      const long long t1 = clock64();
      tot += t1 - t0;
      if (lane == 0) __hip_atomic_store(flag, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) { cyc[0] = tot / iters; cyc[1] = acc; }
  }
}
template <int WRITERS, int AUX, int NLANES, int PRESTORE = 0> void run(const char *nm) {
  double *buf; unsigned *flag; long long *cyc;
  hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20); hipMalloc(&flag, 8192); hipMemset(flag, 0, 8192); hipMalloc(&cyc, 16);
  hipLaunchKernelGGL((k_batch<WRITERS, AUX, NLANES, PRESTORE>), dim3(8 * 17), dim3(64), 0, 0, buf, flag, cyc, 300);
  hipDeviceSynchronize();
  long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-8s prestore=%d writers=%2d lanes=%2d: %lld cycles per batch of 16 (fresh words seen: %lld of %d)\n", nm, PRESTORE, WRITERS, NLANES, c[0], c[1], 300 * 16);
  hipFree(buf); hipFree(flag); hipFree(cyc);
}
int main() {
  run<16, 16, 51, 1>("sc1"); run<16, 16, 51, 2>("sc1");
  run<1, 16, 51>("sc1"); run<16, 16, 51>("sc1"); run<16, 16, 20>("sc1"); run<16, 16, 4>("sc1"); run<16, 17, 51>("sc0 sc1"); run<1, 0, 51>("plain"); run<16, 0, 51>("plain");
  return 0;
}
