// Micro-benchmark (development aid, round 6): what an exchange fetch costs as a function of its VOLUME, in the regime the sampler runs it in
// (16 members per cluster, NCL clusters busy at once: 16 = two per XCD as in twin mode).  Every member stores one row of `nl` tagged words and
// fetches `nr` rows of the others; words of 16 bytes {value, tag} (the sampler's) or of 8 bytes {value32, tag32} fetched as b64.
// Reported: cycles per exchange on member 0 of cluster 0.   hipcc --offload-arch=gfx950 -O3 scripts/micro/xvol.hip -o scripts/micro/xvol
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define K 16
#define ROWB 13312u
#define OOB 0xFFFFFF00u

template <int WB, int NWAVES, int OLD>   // WB: bytes per word (16 or 8); NWAVES: waves that fetch side by side (each its own nr rows, same rows)
__global__ __launch_bounds__(512) void k_vol(unsigned *buf, long long *res, int ncl, int iters, int delay, int nl, int nr) {
  const int cl = blockIdx.x % ncl, m = blockIdx.x / ncl;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf + (size_t)cl * 4 * K * (ROWB / 4), 0, 4u * K * ROWB, 0x00020000);
  long long tot = 0, spins = 0, first = 0;
  int dead = 0;
  unsigned acc = 0;
  for (int it = 1; it <= iters && !dead; it++) {
    const unsigned slot = (unsigned)(it & 3) * K * ROWB;
    const long long t0 = clock64();
    if (w < NWAVES) {
      const unsigned off = lane < nl ? (unsigned)WB * lane : OOB;
      if (w == 0) {
        if (WB == 16) {
          u32x4 v = {(unsigned)(it * 131 + m * 7 + lane), (unsigned)m, (unsigned)it, 0x5ea1u};
          __builtin_amdgcn_raw_buffer_store_b128(v, r, off, slot + (unsigned)m * ROWB, 16);
          asm volatile("s_nop 1" :: "v"(v));
        } else {
          u32x2 v = {(unsigned)(it * 131 + m * 7 + lane), (unsigned)it};
          __builtin_amdgcn_raw_buffer_store_b64(v, r, off, slot + (unsigned)m * ROWB, 16);
        }
      }
      u32x4 d[K];
      bool done[K], all = true;
      // first round as in the sampler's xld: sixteen loads in a row, no branch between them (rows beyond nr: every lane out of range);
      // OLD = 1: the rows of the PREVIOUS exchange (certainly there: a fetch that never waits)
      const unsigned want = OLD ? (unsigned)(it - 1) : (unsigned)it;
      const unsigned rslot = OLD ? (unsigned)((it - 1) & 3) * K * ROWB : slot;
#pragma unroll
      for (int u = 0; u < K; u++) {
        const unsigned o = u < nr ? off : OOB;
        const unsigned so = rslot + (unsigned)((m + 1 + u) & 15) * ROWB;
        if (WB == 16) d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, o, so, 16);
        else { u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, o, so, 16); d[u][0] = t[0]; d[u][2] = t[1]; }
      }
#pragma unroll
      for (int u = 0; u < K; u++) { done[u] = __all(u >= nr || lane >= nl || d[u][2] == want) || (OLD && it == 1); all = all && done[u]; }
      const long long tf = clock64();
      first += tf - t0;
      for (unsigned sp = 1; !all; sp++) {
        if (sp > 2000000u) { dead = 1; break; }
        spins++; __builtin_amdgcn_s_sleep(1); asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < K; u++)
          if (!done[u]) {
            unsigned o = off; asm volatile("" : "+v"(o));
            const unsigned so = rslot + (unsigned)((m + 1 + u) & 15) * ROWB;
            if (WB == 16) d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, o, so, 16);
            else { u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, o, so, 16); d[u][0] = t[0]; d[u][2] = t[1]; }
          }
        all = true;
#pragma unroll
        for (int u = 0; u < K; u++) { if (!done[u]) done[u] = __all(lane >= nl || d[u][2] == want); all = all && done[u]; }
      }
#pragma unroll
      for (int u = 0; u < K; u++) acc += (u < nr && lane < nl && !(OLD && it == 1) && d[u][0] != (unsigned)(want * 131 + ((m + 1 + u) & 15) * 7 + lane)) ? 1u : 0u;
    }
    const long long t1 = clock64();
    tot += t1 - t0;
    __syncthreads();
    for (int s = 0; s < delay; s += 64) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
  }
  if (m == 0 && cl == 0 && tid == 0) { res[0] = tot / iters; res[1] = spins; res[2] = dead; res[3] = first / iters; }
  if (tid == 0) atomicAdd((unsigned long long *)&res[4], (unsigned long long)acc);
}
template <int WB, int NWAVES, int OLD> void run(int ncl, int delay, int nl, int nr) {
  unsigned *buf; long long *res;
  const size_t bytes = (size_t)ncl * 4 * K * ROWB;
  hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes); hipMalloc(&res, 64); hipMemset(res, 0, 64);
  const int iters = 2000;
  hipLaunchKernelGGL((k_vol<WB, NWAVES, OLD>), dim3(ncl * K), dim3(512), 0, 0, buf, res, ncl, iters, delay, nl, nr);
  hipDeviceSynchronize();
  long long c[8]; hipMemcpy(c, res, 64, hipMemcpyDeviceToHost);
  printf("word %2d B, %d fetching wave(s), %s rows, %2d clusters, delay %5d, %2d lanes x %2d rows: %6lld cycles per exchange (first round %5lld), %.2f re-fetch rounds%s, wrong %lld\n",
         WB, NWAVES, OLD ? "OLD" : "new", ncl, delay, nl, nr, c[0], c[3], (double)c[1] / iters, c[2] ? " TIMEOUT" : "", c[4]);
  hipFree(buf); hipFree(res);
}
int main() {
  for (int ncl : {8, 16}) for (int delay : {16000}) {
    for (int nr : {16, 8, 1}) for (int nl : {51, 13}) run<16, 1, 0>(ncl, delay, nl, nr);
    for (int nr : {16, 8, 1}) for (int nl : {51, 13}) run<16, 1, 1>(ncl, delay, nl, nr);
    for (int nr : {16}) for (int nl : {51}) { run<8, 1, 0>(ncl, delay, nl, nr); run<8, 1, 1>(ncl, delay, nl, nr); }
    for (int nr : {16, 8}) for (int nl : {51}) { run<16, 2, 0>(ncl, delay, nl, nr); run<16, 2, 1>(ncl, delay, nl, nr); run<16, 3, 1>(ncl, delay, nl, nr); }
  }
  return 0;
}
