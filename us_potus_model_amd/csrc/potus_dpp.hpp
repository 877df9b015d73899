// potus_dpp.hpp -- wave64 scans and reductions on the DPP data path of gfx950.
//
// __shfl_* on a double is two ds_bpermute_b32 through the LDS crossbar plus index arithmetic and a
// wait each: a 6-step scan of a few doubles costs thousands of cycles when the steps depend on each
// other.  The row_shr / row_bcast modifiers move data between lanes inside the VALU instead
// (v_mov_b32_dpp), so a step is a handful of ALU instructions.  Pattern (inclusive scan over 64 lanes):
// row_shr 1,2,4,8 inside each row of 16 lanes, then row_bcast:15 into rows 1 and 3, then row_bcast:31
// into rows 2 and 3.  Lanes without a source keep the identity passed as `old`.
#pragma once
#include <hip/hip_runtime.h>

#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define DPP_WAVE_SHR1 0x138

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_fetch(double identity, double v) {
  const unsigned long long iv = __builtin_bit_cast(unsigned long long, identity), vv = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)iv, (int)(unsigned)vv, CTRL, ROWMASK, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(iv >> 32), (int)(unsigned)(vv >> 32), CTRL, ROWMASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// value of lane-1 (lane 0 gets `first`)
__device__ __forceinline__ double dpp_prev_lane(double v, double first) { return dpp_fetch<DPP_WAVE_SHR1, 0xf>(first, v); }

// inclusive prefix sum over the wave; lane 63 ends up with the total
__device__ __forceinline__ double dpp_scan_sum(double v) {
  v += dpp_fetch<DPP_ROW_SHR(1), 0xf>(0.0, v);
  v += dpp_fetch<DPP_ROW_SHR(2), 0xf>(0.0, v);
  v += dpp_fetch<DPP_ROW_SHR(4), 0xf>(0.0, v);
  v += dpp_fetch<DPP_ROW_SHR(8), 0xf>(0.0, v);
  v += dpp_fetch<DPP_ROW_BCAST15, 0xa>(0.0, v);
  v += dpp_fetch<DPP_ROW_BCAST31, 0xc>(0.0, v);
  return v;
}
__device__ __forceinline__ double dpp_readlane_d(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// sum over the wave, returned in every lane (fixed order)
__device__ __forceinline__ double dpp_wave_sum(double v) { return dpp_readlane_d(dpp_scan_sum(v), 63); }

// Inclusive scan of affine maps x -> A x + B composed in lane order (lane 0 applied first):
// afterwards lane i holds the composite of lanes 0..i.  NB maps share the multiplier A.
template <int NB>
__device__ __forceinline__ void dpp_scan_affine(double &A, double (&B)[NB]) {
#define POTUS_AFFINE_STEP(CTRL, RM)                                   \
  {                                                                   \
    const double A2 = dpp_fetch<CTRL, RM>(1.0, A);                    \
    double B2[NB];                                                    \
    _Pragma("unroll") for (int i = 0; i < NB; i++) B2[i] = dpp_fetch<CTRL, RM>(0.0, B[i]); \
    _Pragma("unroll") for (int i = 0; i < NB; i++) B[i] = A * B2[i] + B[i];                \
    A = A * A2;                                                       \
  }
  POTUS_AFFINE_STEP(DPP_ROW_SHR(1), 0xf)
  POTUS_AFFINE_STEP(DPP_ROW_SHR(2), 0xf)
  POTUS_AFFINE_STEP(DPP_ROW_SHR(4), 0xf)
  POTUS_AFFINE_STEP(DPP_ROW_SHR(8), 0xf)
  POTUS_AFFINE_STEP(DPP_ROW_BCAST15, 0xa)
  POTUS_AFFINE_STEP(DPP_ROW_BCAST31, 0xc)
#undef POTUS_AFFINE_STEP
}
