// FETCH_SIZE calibration for the sampler's own load forms.  MI355X_MICROARCH.md calibrates the counter only for
// 16 B/lane streaming loads ("reports exactly 1/2 of the bytes ... other access widths are uncalibrated").  The sampler reads its
// state with 8-byte buffer loads (default policy and sc1) and its exchange words with 16-byte sc1 buffer loads, mostly out of
// lines that stay in the XCD's L2.  Kernels, each loading a known number of bytes:
//   k_ld64 / k_ld64_sc1 / k_ld128 / k_ld128_sc1            streaming over 1 GiB (past L2 and the 256 MiB Infinity Cache)
//   k_ld64_reread / k_ld64_sc1_reread / k_ld128_sc1_reread  64 KiB per workgroup read 1024 times (16 MiB in all: L2-resident)
// Run under   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -o r -- ./fetch_calib   and compare per kernel with
// scripts/micro/pmc_calib.py <dir> FETCH_SIZE.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t mk(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000); }

template <int AUX> __global__ void k_ld64_t(const double *p, size_t per_block, unsigned *sink) {
  const rsrc_t r = mk(p + (size_t)blockIdx.x * per_block, (unsigned)(per_block * 8));
  unsigned acc = 0;
  for (size_t i = threadIdx.x; i < per_block; i += 4 * blockDim.x) {
    u32x2 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b64(r, (unsigned)(8 * (i + u * blockDim.x)), 0, AUX);
#pragma unroll
    for (int u = 0; u < 4; u++) acc += v[u][0] ^ v[u][1];
  }
  if (acc == 0x12345u) sink[0] = acc;
}
template <int AUX> __global__ void k_ld128_t(const double *p, size_t per_block, unsigned *sink) {
  const rsrc_t r = mk(p + (size_t)blockIdx.x * per_block, (unsigned)(per_block * 8));
  unsigned acc = 0;
  for (size_t i = threadIdx.x; 2 * i < per_block; i += 4 * blockDim.x) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(16 * (i + u * blockDim.x)), 0, AUX);
#pragma unroll
    for (int u = 0; u < 4; u++) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x12345u) sink[0] = acc;
}
template <int AUX> __global__ void k_ld64_reread_t(const double *p, size_t n, int reps, unsigned *sink) {
  const rsrc_t r = mk(p + (size_t)blockIdx.x * n, (unsigned)(n * 8));
  unsigned acc = 0;
  for (int k = 0; k < reps; k++) {
    asm volatile("" ::: "memory");
    for (size_t i = threadIdx.x; i < n; i += 4 * blockDim.x) {
      u32x2 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { unsigned o = (unsigned)(8 * (i + u * blockDim.x)); asm volatile("" : "+v"(o)); v[u] = __builtin_amdgcn_raw_buffer_load_b64(r, o, 0, AUX); }
#pragma unroll
      for (int u = 0; u < 4; u++) acc += v[u][0] ^ v[u][1];
    }
  }
  if (acc == 0x12345u) sink[0] = acc;
}
template <int AUX> __global__ void k_ld128_reread_t(const double *p, size_t n, int reps, unsigned *sink) {
  const rsrc_t r = mk(p + (size_t)blockIdx.x * n, (unsigned)(n * 8));
  unsigned acc = 0;
  for (int k = 0; k < reps; k++) {
    asm volatile("" ::: "memory");
    for (size_t i = threadIdx.x; 2 * i < n; i += 4 * blockDim.x) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { unsigned o = (unsigned)(16 * (i + u * blockDim.x)); asm volatile("" : "+v"(o)); v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, AUX); }
#pragma unroll
      for (int u = 0; u < 4; u++) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
  }
  if (acc == 0x12345u) sink[0] = acc;
}
int main() {
  const size_t blocks = 256, per_block = (size_t)1 << 19;     // 256 x 4 MiB = 1 GiB
  double *p = nullptr; unsigned *sink = nullptr;
  if (hipMalloc(&p, blocks * per_block * 8) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(p, 1, blocks * per_block * 8);
  hipDeviceSynchronize();
  const size_t n = (size_t)1 << 13; const int reps = 1024;    // 256 x 64 KiB, 1024 times = 16 GiB of loads from 16 MiB
  hipLaunchKernelGGL((k_ld64_t<0>), dim3(blocks), dim3(512), 0, 0, p, per_block, sink); hipDeviceSynchronize();
  hipLaunchKernelGGL((k_ld64_t<16>), dim3(blocks), dim3(512), 0, 0, p, per_block, sink); hipDeviceSynchronize();
  hipLaunchKernelGGL((k_ld128_t<0>), dim3(blocks), dim3(512), 0, 0, p, per_block, sink); hipDeviceSynchronize();
  hipLaunchKernelGGL((k_ld128_t<16>), dim3(blocks), dim3(512), 0, 0, p, per_block, sink); hipDeviceSynchronize();
  hipLaunchKernelGGL((k_ld64_reread_t<0>), dim3(blocks), dim3(512), 0, 0, p, n, reps, sink); hipDeviceSynchronize();
  hipLaunchKernelGGL((k_ld64_reread_t<16>), dim3(blocks), dim3(512), 0, 0, p, n, reps, sink); hipDeviceSynchronize();
  hipLaunchKernelGGL((k_ld128_reread_t<16>), dim3(blocks), dim3(512), 0, 0, p, n, reps, sink); hipDeviceSynchronize();
  // keys = substrings of the (mangled) kernel names in rocprofv3's database: ILi0E = default policy, ILi16E = sc1
  printf("bytes k_ld64_tILi0E %zu\nbytes k_ld64_tILi16E %zu\nbytes k_ld128_tILi0E %zu\nbytes k_ld128_tILi16E %zu\n", blocks * per_block * 8, blocks * per_block * 8, blocks * per_block * 8,
         blocks * per_block * 8);
  printf("bytes k_ld64_reread_tILi0E %zu\nbytes k_ld64_reread_tILi16E %zu\nbytes k_ld128_reread_tILi16E %zu\n", blocks * n * 8 * reps, blocks * n * 8 * reps, blocks * n * 8 * reps);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
