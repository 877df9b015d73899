// Micro-benchmark (development aid, round 4): the tagged-word all-to-all of a cluster in the regime the sampler runs it in -- every
// member stores its 51 words {value, tag} and at once fetches the words of the fifteen others, iteration after iteration, with
// NCL clusters busy at the same time (8 = one per XCD, 16 = two per XCD as in twin mode) -- for every combination of store and load
// cache policy.  Reported: cycles per exchange on member 0 of cluster 0 beyond the fixed delay between exchanges, re-fetch rounds,
// and whether every word arrived (a protocol without cross-CU visibility times out).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define K 16
#define ROWB 13312u

template <int ST_AUX, int LD_AUX>
__global__ __launch_bounds__(512) void k_race(unsigned *buf, long long *res, int ncl, int iters, int delay) {
  const int cl = blockIdx.x % ncl, m = blockIdx.x / ncl;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf + (size_t)cl * 4 * K * (ROWB / 4), 0, 4u * K * ROWB, 0x00020000);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long tot = 0, spins = 0;
  int dead = 0;
  unsigned acc = 0;
  for (int it = 1; it <= iters && !dead; it++) {
    const unsigned slot = (unsigned)(it & 3) * K * ROWB;
    const long long t0 = clock64();
    if (w == 0) {
      u32x4 v = {(unsigned)(it * 131 + m * 7 + lane), (unsigned)m, (unsigned)it, 0x5ea1u};
      __builtin_amdgcn_raw_buffer_store_b128(v, r, lane < 51 ? 16u * lane : 0xFFFFFF00u, slot + (unsigned)m * ROWB, ST_AUX);
      asm volatile("s_nop 1" :: "v"(v));
      u32x4 d[K];
      bool done[K], all = false;
#pragma unroll
      for (int u = 0; u < K; u++) done[u] = false;
      for (unsigned sp = 0; !all; sp++) {
        if (sp > 2000000u) { dead = 1; break; }
        if (sp) { spins++; __builtin_amdgcn_s_sleep(1); asm volatile("" ::: "memory"); }
#pragma unroll
        for (int u = 0; u < K; u++)
          if (!done[u]) { unsigned o = lane < 51 ? 16u * lane : 0xFFFFFF00u; asm volatile("" : "+v"(o)); d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, o, slot + (unsigned)u * ROWB, LD_AUX); }
        all = true;
#pragma unroll
        for (int u = 0; u < K; u++) { if (!done[u]) done[u] = __all(lane >= 51 || d[u][2] == (unsigned)it); all = all && done[u]; }
      }
#pragma unroll
      for (int u = 0; u < K; u++) acc += (lane < 51 && d[u][0] != (unsigned)(it * 131 + u * 7 + lane)) ? 1u : 0u;   // wrong value under a right tag
    }
    const long long t1 = clock64();
    tot += t1 - t0;
    __syncthreads();
    for (int s = 0; s < delay; s += 64) __builtin_amdgcn_s_sleep(1);       // the rest of the pass
    __syncthreads();
  }
  if (m == 0 && cl == 0 && tid == 0) { res[0] = tot / iters; res[1] = spins; res[2] = dead; res[3] = xcc & 15; }
  if (tid == 0) atomicAdd((unsigned long long *)&res[4], (unsigned long long)acc);
  if (tid == 0 && m == K - 1 && cl == 0) res[5] = xcc & 15;
}
template <int ST_AUX, int LD_AUX> void run(int ncl, int delay) {
  unsigned *buf; long long *res;
  const size_t bytes = (size_t)ncl * 4 * K * ROWB;
  hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes); hipMalloc(&res, 64); hipMemset(res, 0, 64);
  const int iters = 2000;
  hipLaunchKernelGGL((k_race<ST_AUX, LD_AUX>), dim3(ncl * K), dim3(512), 0, 0, buf, res, ncl, iters, delay);
  hipDeviceSynchronize();
  long long c[8]; hipMemcpy(c, res, 64, hipMemcpyDeviceToHost);
  printf("store aux %2d, load aux %2d, %2d clusters, delay %5d: %6lld cycles per exchange, %.2f re-fetch rounds, %s, wrong values %lld (XCC of members 0 / 15: %lld / %lld)\n",
         ST_AUX, LD_AUX, ncl, delay, c[0], (double)c[1] / iters, c[2] ? "TIMEOUT (not visible)" : "all words arrived", c[4], c[3], c[5]);
  hipFree(buf); hipFree(res);
}
int main() {
  // aux bits: 1 = sc0, 2 = nt, 16 = sc1
  for (int ncl : {8, 16}) for (int delay : {0, 4000, 16000}) {
    run<16, 16>(ncl, delay);      // the sampler's protocol: sc1 stores, sc1 loads
    run<17, 17>(ncl, delay);      // sc0 + sc1 both ways
    run<1, 16>(ncl, delay);       // sc0 stores (L2 of the XCD, no write-through to memory), sc1 loads
    run<1, 1>(ncl, delay);        // sc0 stores, sc0 loads
    run<0, 16>(ncl, delay);       // plain stores, sc1 loads
    run<1, 17>(ncl, delay);
    run<0, 1>(ncl, delay);
  }
  return 0;
}
