// Micro-benchmark (development aid): one-way latency of "store a tagged 16-byte word, the other
// workgroup spins on loading it", for several cache-policy bits on the buffer store / load, between
// two workgroups on the same XCD (blocks 0 and 8) and on different XCDs (blocks 0 and 1).
// Loads are inline asm so that the compiler can neither hoist them nor change their policy bits.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define DEF_KERNEL(NAME, STPOL, LDPOL)                                                                   \
  __global__ __launch_bounds__(64) void NAME(unsigned *buf, int other_block, int iters, long long *cycles, unsigned *xcc) { \
    const int me = blockIdx.x == 0 ? 0 : (int)blockIdx.x == other_block ? 1 : -1;                        \
    if (me < 0) return;                                                                                  \
    unsigned id;                                                                                         \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));                                    \
    if (threadIdx.x == 0) xcc[me] = id;                                                                  \
    const unsigned long long p = (unsigned long long)buf;                                                \
    u32x4 r = {(unsigned)p, (unsigned)(p >> 32) & 0xffffu, 4096u, 0x00020000u};                          \
    r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);            \
    const unsigned mine = me * 1024u, theirs = (1 - me) * 1024u;                                         \
    const long long t0 = clock64();                                                                      \
    int fail = 0;                                                                                        \
    for (int i = 1; i <= iters && !fail; i++) {                                                          \
      u32x4 w = {(unsigned)i, 0u, (unsigned)i, 7u};                                                      \
      if (me == 0) asm volatile("buffer_store_dwordx4 %0, off, %1, %2 " STPOL :: "v"(w), "s"(r), "s"(mine) : "memory"); \
      int spin = 0;                                                                                      \
      for (;; spin++) {                                                                                  \
        u32x4 v;                                                                                         \
        asm volatile("buffer_load_dwordx4 %0, off, %1, %2 " LDPOL "\n s_waitcnt vmcnt(0)" : "=v"(v) : "s"(r), "s"(theirs) : "memory"); \
        if (v[2] == (unsigned)i) break;                                                                  \
        if (spin > 3000000) { fail = 1; break; }                                                         \
      }                                                                                                  \
      if (me == 1) asm volatile("buffer_store_dwordx4 %0, off, %1, %2 " STPOL :: "v"(w), "s"(r), "s"(mine) : "memory"); \
    }                                                                                                    \
    const long long t1 = clock64();                                                                      \
    if (threadIdx.x == 0 && me == 0) { cycles[0] = t1 - t0; cycles[1] = fail; }                          \
  }

DEF_KERNEL(k_sc1_sc1, "sc1", "sc1")
DEF_KERNEL(k_sys_sys, "sc0 sc1", "sc0 sc1")
DEF_KERNEL(k_plain_sc0, "", "sc0")
DEF_KERNEL(k_sc0_sc0, "sc0", "sc0")
DEF_KERNEL(k_plain_sc1, "", "sc1")
DEF_KERNEL(k_sc1_sc0, "sc1", "sc0")
DEF_KERNEL(k_plain_plain, "", "")
DEF_KERNEL(k_nt_nt, "nt", "nt")
DEF_KERNEL(k_sc1_nt, "sc1", "nt")
DEF_KERNEL(k_plain_nt, "", "nt")
DEF_KERNEL(k_sc1_sc0nt, "sc1", "sc0 nt")
DEF_KERNEL(k_sc1_sc1nt, "sc1 nt", "sc1 nt")

template <class F> void run(F kern, const char *name, int other) {
  unsigned *buf, *xcc; long long *cyc;
  hipMalloc(&buf, 4096); hipMemset(buf, 0, 4096); hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16); hipMalloc(&xcc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(16), dim3(64), 0, 0, buf, other, iters, cyc, xcc);
  hipEventRecord(e1);
  if (hipEventSynchronize(e1) != hipSuccess) { printf("%s: FAILED\n", name); return; }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c[2]; unsigned x[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
  printf("%-30s blocks 0,%d (XCC %u,%u): %s %.3f us one-way, %lld cycles one-way\n", name, other, x[0] & 15, x[1] & 15, c[1] ? "NOT VISIBLE (timeout)" : "", 1e3 * ms / iters / 2, c[0] / iters / 2);
  hipFree(buf); hipFree(cyc); hipFree(xcc);
}
int main() {
  for (int other : {8, 1}) {
    run(k_sc1_sc1, "store sc1, load sc1", other);
    run(k_sys_sys, "store sc0 sc1, load sc0 sc1", other);
    run(k_plain_sc0, "store plain, load sc0", other);
    run(k_sc0_sc0, "store sc0, load sc0", other);
    run(k_plain_sc1, "store plain, load sc1", other);
    run(k_sc1_sc0, "store sc1, load sc0", other);
    run(k_nt_nt, "store nt, load nt", other);
    run(k_sc1_nt, "store sc1, load nt", other);
    run(k_plain_nt, "store plain, load nt", other);
    run(k_sc1_sc0nt, "store sc1, load sc0 nt", other);
    run(k_sc1_sc1nt, "store sc1 nt, load sc1 nt", other);
    run(k_plain_plain, "store plain, load plain", other);
  }
  return 0;
}
