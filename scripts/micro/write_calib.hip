// WRITE_SIZE calibration (MI355X_MICROARCH.md: "WRITE_SIZE is uncalibrated: calibrate on a known byte count in your own access
// pattern").  Three kernels store a known number of bytes with the store forms the sampler uses:
//   k_plain64   8 bytes per lane, contiguous, default policy      (the epilogue's leaf momentum / position stores)
//   k_sc1_64    8 bytes per lane, contiguous, sc1 (write-through) (positions other members read)
//   k_sc1_128   16 bytes per lane, contiguous, sc1                (exchange words {value, tag})
// over a buffer larger than L2 + Infinity Cache.  Run under
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace -d <dir> -o r -- ./write_calib
// and compare the counter of each kernel with the bytes printed here (scripts/micro/write_calib.py).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t mk(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000); }
// each workgroup owns a 1 GiB-addressable slice through its own resource (32-bit offsets)
__global__ void k_plain64(double *p, size_t per_block) {
  double *b = p + (size_t)blockIdx.x * per_block;
  const rsrc_t r = mk(b, (unsigned)(per_block * 8));
  for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)i, 1u}, r, (unsigned)(8 * i), 0, 0);
}
__global__ void k_sc1_64(double *p, size_t per_block) {
  double *b = p + (size_t)blockIdx.x * per_block;
  const rsrc_t r = mk(b, (unsigned)(per_block * 8));
  for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)i, 2u}, r, (unsigned)(8 * i), 0, 16);
}
__global__ void k_sc1_128(double *p, size_t per_block) {
  double *b = p + (size_t)blockIdx.x * per_block;
  const rsrc_t r = mk(b, (unsigned)(per_block * 8));
  for (size_t i = threadIdx.x; 2 * i < per_block; i += blockDim.x) __builtin_amdgcn_raw_buffer_store_b128(u32x4{(unsigned)i, 3u, 4u, 5u}, r, (unsigned)(16 * i), 0, 16);
}
// the same 1 MiB rewritten many times with sc1 stores: what a chain's state (resident in L2) looks like to the counter
__global__ void k_sc1_64_rewrite(double *p, size_t n, int reps) {
  const rsrc_t r = mk(p + (size_t)blockIdx.x * n, (unsigned)(n * 8));
  for (int k = 0; k < reps; k++)
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)i, (unsigned)k}, r, (unsigned)(8 * i), 0, 16);
}
__global__ void k_plain64_rewrite(double *p, size_t n, int reps) {
  const rsrc_t r = mk(p + (size_t)blockIdx.x * n, (unsigned)(n * 8));
  for (int k = 0; k < reps; k++)
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)i, (unsigned)k}, r, (unsigned)(8 * i), 0, 0);
}
int main() {
  const size_t blocks = 256, per_block = (size_t)1 << 19;     // 256 x 4 MiB = 1 GiB
  double *p = nullptr;
  if (hipMalloc(&p, blocks * per_block * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(p, 0, blocks * per_block * 8);
  hipDeviceSynchronize();
  k_plain64<<<blocks, 512>>>(p, per_block); hipDeviceSynchronize();
  k_sc1_64<<<blocks, 512>>>(p, per_block); hipDeviceSynchronize();
  k_sc1_128<<<blocks, 512>>>(p, per_block); hipDeviceSynchronize();
  const size_t n = (size_t)1 << 17; const int reps = 64;       // 256 x 1 MiB, 64 times = 16 GiB of stores onto 256 MiB
  k_sc1_64_rewrite<<<blocks, 512>>>(p, n, reps); hipDeviceSynchronize();
  k_plain64_rewrite<<<blocks, 512>>>(p, n, reps); hipDeviceSynchronize();
  printf("bytes k_plain64 %zu\nbytes k_sc1_64 %zu\nbytes k_sc1_128 %zu\nbytes k_sc1_64_rewrite %zu\nbytes k_plain64_rewrite %zu\n", blocks * per_block * 8, blocks * per_block * 8,
         blocks * per_block * 8, blocks * n * 8 * reps, blocks * n * 8 * reps);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
