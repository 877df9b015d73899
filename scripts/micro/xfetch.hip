// Micro-benchmark (development aid, round 4): what an exchange fetch costs as a function of its shape.  One reader workgroup of NW waves;
// wave w fetches NI rows (NLANES lanes x 16 or 8 bytes each, row stride ROWB) that 16 writer workgroups of the same XCD have just
// rewritten with sc1 stores (as the cluster's exchange does); all loads of a wave in flight at once.  Reported: cycles per batch on wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int NI, int NW, int WIDE, int AUX>
__global__ __launch_bounds__(64 * NW) void k_fetch(double *buf, unsigned *flag, long long *cyc, int iters, int nlanes, unsigned rowb) {
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1u << 22, 0x00020000);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x;
  if (b % 8 != 0) return;                      // blocks 0, 8, 16, ... share an XCD
  const int id = b / 8;                        // 0 = reader, 1..16 writers
  const unsigned vo = lane < nlanes ? (WIDE ? 16u : 8u) * lane : 0xFFFFFF00u;
  if (id >= 1 && id <= 16) {
    if (w > 0) return;
    for (int it = 1; it <= iters; it++) {
      for (int ww = 0; ww < NW; ww++)
        for (int u = id - 1; u < NI; u += 16) {      // writer id-1 owns rows u = id-1, id+15, ... of every reader wave
          const unsigned so = (unsigned)(ww * NI + u) * rowb;
          if (WIDE) { u32x4 v = {(unsigned)it, (unsigned)u, (unsigned)it, 7u}; __builtin_amdgcn_raw_buffer_store_b128(v, r, vo, so, 16); asm volatile("s_nop 1" :: "v"(v)); }
          else { u32x2 v = {(unsigned)it, (unsigned)it}; __builtin_amdgcn_raw_buffer_store_b64(v, r, vo, so, 16); }
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(flag + 32 * id, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(2);
    }
  } else if (id == 0) {
    long long tot = 0;
    unsigned acc = 0;
    for (int it = 1; it <= iters; it++) {
      for (int wr = 1; wr <= 16; wr++)
        while (__hip_atomic_load(flag + 32 * wr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(2);
      __syncthreads();
      __builtin_amdgcn_s_sleep(40);
      const long long t0 = clock64();
      if (WIDE) {
        u32x4 d[NI];
#pragma unroll
        for (int u = 0; u < NI; u++) d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, vo, (unsigned)(w * NI + u) * rowb, AUX);
#pragma unroll
        for (int u = 0; u < NI; u++) acc += d[u][2] == (unsigned)it;
      } else {
        u32x2 d[NI];
#pragma unroll
        for (int u = 0; u < NI; u++) d[u] = __builtin_amdgcn_raw_buffer_load_b64(r, vo, (unsigned)(w * NI + u) * rowb, AUX);
#pragma unroll
        for (int u = 0; u < NI; u++) acc += d[u][1] == (unsigned)it;
      }
      asm volatile("" : "+v"(acc));
      const long long t1 = clock64();
      if (w == 0) tot += t1 - t0;
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(flag, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) { cyc[0] = tot / iters; cyc[1] = acc; }
  }
}
template <int NI, int NW, int WIDE, int AUX = 16> void run(int nlanes, unsigned rowb) {
  const int aux_plain = AUX == 0;
  double *buf; unsigned *flag; long long *cyc;
  hipMalloc(&buf, 1 << 22); hipMemset(buf, 0, 1 << 22); hipMalloc(&flag, 8192); hipMemset(flag, 0, 8192); hipMalloc(&cyc, 16);
  const int iters = 300;
  hipLaunchKernelGGL((k_fetch<NI, NW, WIDE, AUX>), dim3(8 * 17), dim3(64 * NW), 0, 0, buf, flag, cyc, iters, nlanes, rowb);
  hipDeviceSynchronize();
  long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  printf("loads/wave %2d x %2d B, lanes %2d, row stride %5u, reader waves %d%s: %5lld cycles per batch (fresh words seen on wave 0: %lld of %d)\n", NI, WIDE ? 16 : 8, nlanes, rowb, NW,
         aux_plain ? ", PLAIN loads" : "", c[0], c[1], iters * NI);
  hipFree(buf); hipFree(flag); hipFree(cyc);
}
int main() {
  run<16, 1, 1>(51, 13312); run<8, 1, 1>(51, 13312); run<4, 1, 1>(51, 13312); run<2, 1, 1>(51, 13312); run<1, 1, 1>(51, 13312);
  run<16, 1, 1>(13, 13312); run<16, 1, 1>(64, 13312); run<16, 1, 1>(51, 832); run<16, 1, 1>(51, 1024);
  run<16, 1, 0>(51, 13312); run<16, 1, 0>(51, 512); run<8, 1, 0>(64, 512);
  run<16, 2, 1>(51, 13312); run<16, 4, 1>(51, 13312); run<8, 2, 1>(51, 13312); run<8, 4, 1>(51, 13312); run<4, 4, 1>(51, 13312); run<2, 8, 1>(51, 13312);
  run<16, 1, 1, 0>(51, 13312);
  return 0;
}
