// potus_model.hpp -- device-side log-density + gradient of the poll model for gfx950.
//
// One 512-thread workgroup (8 wave64: PT_NW, two waves per SIMD) evaluates log_prob and its gradient for ONE chain.
// The model is scripts/model/poll_model_2020.stan:56-131 (and the no_mode_adjustment
// variant), re-derived for the hardware instead of transcribed:
//
//  * the T-1 dependent 51x51 mat-vecs of stan:86 become a suffix sum over days,
//      mu_b[:,t] = (L_T z_T + prior) + L_W * C[:,t],  C[:,t] = sum_{u=t}^{T-2} Z[:,u]
//    held in LDS as C[k][t]; wave w owns days [PT_CH w, PT_CH (w+1)) (PT_CH = 32), lane k owns state k, so the S x T
//    block is read from HBM/L2 with lanes on consecutive addresses (coalesced) and scanned in
//    registers with one carry exchange (one entry per wave) through LDS;
//  * mu_b is only needed at polled (state,day) cells: one thread per poll does the 51-term
//    dot  L_W[s,:] . C[:,t]  out of LDS (national polls use the extra row v = L_W^T w);
//  * the adjoint is the mirror image: per-day gathers  gC[:,t] = sum_i r_i L_W[s_i,:]  (one
//    wave per day, days pre-balanced over waves on the host), a prefix sum over days in
//    registers, and  dZ = prefix - Z;
//  * all index-driven reductions (pollster / mode / population / state / day) run as two-level
//    segment sums over host-built lists, so the summation order is fixed: same inputs give
//    the same bytes on every run (no floating-point atomics anywhere);
//  * the AR(1) bias (stan:91-93) and its adjoint are affine scans done by one wave.
//
// Every parameter element is loaded exactly once (pol.q) and its gradient stored exactly once
// (pol.g) per pass; the leapfrog integrator exploits that to fuse its position and momentum
// updates into the same pass (see LeapPolicy in potus_nuts.hpp).
//
// Address spaces are spelled out (global / LDS / constant) so that every access is a
// global_load / ds_read / s_load rather than a flat operation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "potus_dpp.hpp"

#ifndef PT_NW
#define PT_NW 8           // waves per workgroup: 8 -> two waves per SIMD, 256 VGPRs each (no spills)
#endif
#define PT_THREADS (64 * PT_NW)
#define PT_CH (256 / PT_NW) // days per wave in the S x T block layout  (T <= 256)
#define PT_KJ (64 / PT_NW)  // rows of a 51 x 51 factor handled per wave in the split mat-vecs
#define PT_SUBLEN 16      // entries per level-1 segment-sum task

#define AS_G __attribute__((address_space(1)))
#define AS_L __attribute__((address_space(3)))
#define AS_C __attribute__((address_space(4)))
typedef double AS_G *gdp;
typedef const double AS_G *gcdp;
typedef const int AS_G *gcip;
typedef double AS_L *ldp;
template <class T> __device__ __forceinline__ T AS_G *as_g(T *p) { return (T AS_G *)p; }
template <class T> __device__ __forceinline__ const T AS_C *as_c(const T *p) { return (const T AS_C *)p; }

// Buffer addressing (SGPR resource + 32-bit VGPR byte offset + SGPR offset): one VGPR addresses the
// same element of every vector of a chain, instead of a 64-bit VGPR pair per access.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double bld(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ int bld_i(rsrc_t r, unsigned voff, unsigned soff) {
  return (int)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}
__device__ __forceinline__ u32x4 bld_i4(rsrc_t r, unsigned voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0); }
__device__ __forceinline__ void bld_d2(rsrc_t r, unsigned voff, unsigned soff, double &a, double &b) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  a = __hiloint2double((int)v[1], (int)v[0]); b = __hiloint2double((int)v[3], (int)v[2]);
}
__device__ __forceinline__ void bst(rsrc_t r, unsigned voff, unsigned soff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, 0);
}

struct DevModel {
  int S, T, P, M, Pop, Ns, Nn, Npoll, D, full;
  int SE, SP, TP;           // SE = S+1 (rows of Lw_ext); SP, TP odd LDS row strides
  int o_zT, o_Z, o_c, o_m, o_pop, o_mue, o_rho, o_ze, o_nn, o_ns, o_zb;
  int nmid;                 // o_nn - o_c : parameters between the S x T block and the noise blocks
  double sigma_c, sigma_m, sigma_pop, sigma_e, sigma_ns, sigma_nn;
  // stan:42-55: the three covariances are one matrix times three scalars, so cholesky_ss_cov_mu_b_T = aT * cholesky_ss_cov_mu_b_walk
  // and cholesky_ss_cov_poll_bias = aB * cholesky_ss_cov_mu_b_walk (aT = mu_b_T_scale / random_walk_scale, aB = polling_bias_scale /
  // random_walk_scale).  The cluster pass (potus_cluster.hpp) uses that: L_T z_T + L_B z_b + L_W C[:,t] = L_W (aT z_T + aB z_b + C[:,t]).
  double aT, aB;
  // Static data is packed into three buffers so that the descriptor costs few scalar registers.
  const double *mat;        // Lw_ext [SE][SP] | LT_t [k][s] | LB_t [k][s] | LT [s][k] | LB [s][k] | prior [S] | w [S] | prior_ext [S + 1]
  int m_LTt, m_LBt, m_LT, m_LB, m_prior, m_w;   // offsets into mat (Lw_ext at 0)
  int m_priorx;             // prior [S] followed by its national average sum_s w_s prior_s (the pseudo-state's entry)
  //   Lw_ext rows 0..S-1: L_W (zeros above the diagonal); row S: v = L_W^T w
  const int *pi;            // polls sorted by day, struct of arrays with stride Npad:
  const double *pd;         //   pi: ps | pt | pp | pm | ppop | pqidx      pd: py | pn | punadj | psig
  int Npad;
  // Static schedule, built on the host (all offsets in ints into sched):
  //   wd_t | wd_a | wd_b [NW][64]  days gathered by wave w (balanced over waves, LPT): day and its
  //                          poll range [a,b) in the day-sorted poll order; unused slots have a = b
  //   wpk [NW][64]           states of wave w's gathered polls in gather order, four bytes per word
  //   daymask [NW]           bit j of word w: day PT_CH*w+j has at least one poll
  //   sub16 [nsub][16]       level-1 segment-sum tasks: poll indices, padded with Npoll (a zero slot)
  //   seg_ptr [nseg+1] | seg_kind | seg_index   level-2: range of tasks, what the sum feeds
  const int *sched;
  int c_wda, c_wdb, c_wpk, c_mask, c_sub16, c_segptr, c_segkind, c_segidx;   // wd_t at 0
  const double *seg_scale;  // [nseg]
  const double *sub_wt16;   // [(nsub - sub_weighted_begin)][16] weights of the weighted tasks (0 for padding)
  int nsub, nseg, sub_weighted_begin;
  // LDS layout, offsets in doubles
  int l_C, l_Lw, l_X, l_Y, l_zT, l_zb, l_mid, l_bT, l_pb, l_e, l_gs, l_ge, l_scal, l_red, l_prof, l_st;
  int lds_doubles;
};
typedef const DevModel AS_C *CMp; // the model descriptor lives in device memory, read through scalar loads

// scalar slots in LDS (l_scal)
enum { SC_MUE = 0, SC_RHO, SC_SRHO, SC_XMUE, SC_XRHO, SC_LPRHO, SC_DSRHO, SC_N };
#define PT_NRED 8  // reduction slots
#define PT_NPROF 64

#ifndef POTUS_PROF_WAVES
#define POTUS_PROF_WAVES 1                  // the phase whose per-wave arrival times (cycles since the phase began) a profile build records in slots 32-39:
#endif                                      // 1 = B (WPROF_ACC), 3 = C, 4 = D, 5 = E, 6 = E2, 7 = F up to its first barrier
#ifdef POTUS_PROF
#define PROF_MARK(k) do { if (threadIdx.x == 0) { const long long t_ = clock64(); prof[k] += (double)(t_ - (long long)prof[PT_NPROF - 1]); prof[PT_NPROF - 1] = (double)t_; } } while (0)
#define PROF_START() do { if (threadIdx.x == 0) prof[PT_NPROF - 1] = (double)clock64(); } while (0)
#define PROF_SUB(k) do { if (threadIdx.x == 0) { prof[k] += (double)(clock64() - (long long)prof[PT_NPROF - 1]); } } while (0)
// per-wave timers: slot 32 + 8*ph + wave accumulates the time a wave spends between WPROF_T0 and WPROF_ACC(ph)
#define WPROF_T0() const long long wt0_ = clock64()
#define WPROF_ACC(ph) do { if ((threadIdx.x & 63) == 0) prof[32 + 8 * (ph) + (threadIdx.x >> 6)] += (double)(clock64() - wt0_); } while (0)
#define WAVE_ARRIVE(ph) do { if (POTUS_PROF_WAVES == (ph) && (threadIdx.x & 63) == 0) prof[32 + (threadIdx.x >> 6)] += (double)(clock64() - (long long)prof[PT_NPROF - 1]); } while (0)
// timeline stamps (device-wide 100 MHz clock) of one chosen leaf: slot 40 + k
#define TSTAMP(k) do { if (threadIdx.x == 0 && prof[16] == 3000.0) prof[40 + (k)] = (double)wall_clock64(); } while (0)
#define WPROF_PT(k) do { if (threadIdx.x == 0) prof[k] += (double)(clock64() - wt0_); } while (0)   // thread 0: time since WPROF_T0
#ifndef POTUS_PROF_CMASK
#define POTUS_PROF_CMASK 0                  // -DPOTUS_PROF -DPOTUS_PROF_CMASK=15: four stamps of thread 0 inside the poll phase of the cluster pass (slots 56-59,
#endif                                      // shared with -DPOTUS_PROF_FETCH: one or the other).  The stamps stay in scalar registers until the phase is over:
                                            // an LDS update under `if (threadIdx.x == 0)` inside the poll arithmetic made the allocator spill 150 vector registers.
#define WPROF_CT0() const long long wct0_ = clock64(); long long wct_[4] = {wct0_, wct0_, wct0_, wct0_}
#define WPROF_CSTAMP(i) do { if constexpr ((POTUS_PROF_CMASK >> (i)) & 1) wct_[i] = clock64(); } while (0)
#define WPROF_CFLUSH() do { if (POTUS_PROF_CMASK != 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < 4; i_++) prof[56 + i_] += (double)(wct_[i_] - wct0_); } } while (0)
#define WPROF_T0B() const long long wt0b_ = clock64()
#define WPROF_PTB(k) do { if ((threadIdx.x & 63) == 0) prof[k] += (double)(clock64() - wt0b_); } while (0)
#define WPROF_ACCB(ph) do { if ((threadIdx.x & 63) == 0) prof[32 + 8 * (ph) + (threadIdx.x >> 6)] += (double)(clock64() - wt0b_); } while (0)
#define WPROF_T0C() const long long wt0c_ = clock64()
#define WPROF_ACCC(ph) do { if ((threadIdx.x & 63) == 0) prof[32 + 8 * (ph) + (threadIdx.x >> 6)] += (double)(clock64() - wt0c_); } while (0)
#else
#define TSTAMP(k) do { } while (0)
#define WPROF_PT(k) do { } while (0)
#define WPROF_CT0() do { } while (0)
#define WPROF_CSTAMP(i) do { } while (0)
#define WPROF_CFLUSH() do { } while (0)
#define WPROF_T0B() do { } while (0)
#define WPROF_PTB(k) do { } while (0)
#define WPROF_ACCB(ph) do { } while (0)
#define WPROF_T0C() do { } while (0)
#define WPROF_ACCC(ph) do { } while (0)
#define WPROF_T0() do { } while (0)
#define WPROF_ACC(ph) do { } while (0)
#define WAVE_ARRIVE(ph) do { } while (0)
#define PROF_MARK(k) do { } while (0)
#define PROF_START() do { } while (0)
#define PROF_SUB(k) do { } while (0)
#endif

// exp(-a) for a >= 0, and log1p(e) together with 1 / (1 + e) for 0 <= e <= 1: what binomial_logit needs per poll (stan:130-131), written
// for LATENCY.  The library versions are ~200 instructions of which most depend on the one before (Horner polynomials, frexp / ldexp,
// div_scale / div_fixup, special cases that cannot occur here); with one poll per lane and one or two waves per SIMD that chain is
// long; whether the poll phase of the cluster pass waits for it was the question (it does not: profiles/r04_cl_fold.txt).  Here: argument
// reduction by rounding, polynomials in Estrin form (independent halves), one reciprocal seed + two Newton steps shared by the
// division of the logistic and the one inside the logarithm.  Accuracy (CPU prototype against glibc over 2 x 10^7 arguments,
// scripts/micro/fast_binomial_check.c): exp within 1.0, log1p within 2.0 units of 2^-52 relative -- the library's class.
__device__ __forceinline__ double d_recip_nr(double d) {   // 1 / d for a normal d of moderate size: v_rcp_f64 + two Newton steps (as the compiler's own division)
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0); y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0); y = __builtin_fma(y, e, y);
  return y;
}
__device__ __forceinline__ double d_exp_neg(double a) {
  const double x = -fmin(a, 745.2);                                    // (below exp(-745.2) the double is zero anyway)
  const double n = __builtin_rint(x * 1.4426950408889634074);
  double r = __builtin_fma(n, -6.93147180369123816490e-01, x);         // ln 2 in two pieces
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);                // |r| <= ln 2 / 2
  const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
  // exp(r) = 1 + r + r^2 q(r), q = sum_{k = 2}^{13} r^(k - 2) / k!  (the next term is below 2^-57)
  const double q01 = __builtin_fma(1.0 / 6, r, 1.0 / 2), q23 = __builtin_fma(1.0 / 120, r, 1.0 / 24), q45 = __builtin_fma(1.0 / 5040, r, 1.0 / 720);
  const double q67 = __builtin_fma(1.0 / 362880, r, 1.0 / 40320), q89 = __builtin_fma(1.0 / 39916800, r, 1.0 / 3628800);
  const double qab = __builtin_fma(1.0 / 6227020800.0, r, 1.0 / 479001600);
  const double q03 = __builtin_fma(q23, r2, q01), q47 = __builtin_fma(q67, r2, q45), q8b = __builtin_fma(qab, r2, q89);
  const double q = __builtin_fma(q8b, r8, __builtin_fma(q47, r4, q03));
  return ldexp(1.0 + __builtin_fma(r2, q, r), (int)n);
}
__device__ __forceinline__ void d_log1p_recip(double e, double &l1, double &inv) {   // l1 = log1p(e), inv = 1 / (1 + e);  0 <= e <= 1
  const double u = 1.0 + e, c = e - (u - 1.0);                         // c: what the sum lost
  const double yu = d_recip_nr(u);
  inv = yu;
  const bool k = u > 1.4142135623730951;                               // log u = k ln 2 + log u',  u' = u / 2^k in (0.707, 1.414]
  const double up = k ? 0.5 * u : u;
  const double f = up - 1.0, d = 2.0 + f;
  const double yd = d_recip_nr(d);
  double s = f * yd;
  s = __builtin_fma(__builtin_fma(-d, s, f), yd, s);                   // s = f / (2 + f), residual corrected
  const double z = s * s, z2 = z * z, z4 = z2 * z2;
  // log u' = 2 atanh(s) = 2 s + s z Q(z),  Q = sum_{i >= 1} 2 z^(i - 1) / (2 i + 1), twelve terms (z <= 0.0295)
  const double t01 = __builtin_fma(2.0 / 5, z, 2.0 / 3), t23 = __builtin_fma(2.0 / 9, z, 2.0 / 7), t45 = __builtin_fma(2.0 / 13, z, 2.0 / 11);
  const double t67 = __builtin_fma(2.0 / 17, z, 2.0 / 15), t89 = __builtin_fma(2.0 / 21, z, 2.0 / 19), tab = __builtin_fma(2.0 / 25, z, 2.0 / 23);
  const double t03 = __builtin_fma(t23, z2, t01), t47 = __builtin_fma(t67, z2, t45), t8b = __builtin_fma(tab, z2, t89);
  const double Q = __builtin_fma(t8b, z4 * z4, __builtin_fma(t47, z4, t03));
  const double lo = __builtin_fma(s * z, Q, c * yu) + (k ? 1.90821492927058770002e-10 : 0.0);
  l1 = (k ? 6.93147180369123816490e-01 : 0.0) + (2.0 * s + lo);
}
__device__ __forceinline__ double d_log_inv_logit(double x) { return x > 0 ? -log1p(exp(-x)) : x - log1p(exp(x)); }
__device__ __forceinline__ double d_inv_logit(double x) {
  if (x >= 0) return 1.0 / (1.0 + exp(-x));
  double e = exp(x);
  return e / (1.0 + e);
}

// The compiler tends to sink every LDS read next to its use (read, wait ~100 cycles, one FMA, next
// read ...).  ISSUE_FENCE() pins the order "all loads of the batch, then all uses".
#define ISSUE_FENCE() __builtin_amdgcn_sched_barrier(0)

// Sum N values over the workgroup; every thread returns with the totals.  Fixed order.
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], ldp red, int tid) {
  const int lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const double x = dpp_scan_sum(v[k]);          // lane 63 ends up with the wave's total (DPP, no LDS crossbar)
    if (lane == 63) red[w * N + k] = x;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; k++) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < PT_NW; i++) s += red[i * N + k];
    v[k] = s;
  }
  __syncthreads();
}

// Plain policy: read q, write grad (parity hook, Hamiltonian init).
// Policy API: q_load/q_fin fetch one position element (split so that callers can issue a batch
// of loads before the first dependent use); g_load/g_fin consume one gradient element, g_fin is
// also handed the element's position (the integrator needs it for the next position update).
// All element accesses take the element's BYTE offset (8*index) or PT_OOB: buffer loads beyond the
// resource return 0 and stores are dropped, so masked-off elements need no branch -- which matters,
// because the compiler drains the whole memory queue (s_waitcnt vmcnt(0)) at every branch join.
#define PT_OOB 0xFFFFFF00u
struct PlainPolicy {
  rsrc_t rq, rg;        // buffers holding q and receiving grad (may be the same buffer)
  unsigned sq, sg;      // byte offsets of the two vectors inside them
  static constexpr int NEXTRA = 0;
  static constexpr int GB = 8;    // gradient elements finished per batch in the S x T block
  double extra[1];  // unused
  struct QT { double q; };
  struct GT { };
  __device__ __forceinline__ void q_load(unsigned vo, QT &t) { t.q = bld(rq, vo, sq); }
  __device__ __forceinline__ double q_fin(unsigned, QT &t) { return t.q; }
  __device__ __forceinline__ void g_load(unsigned, GT &) {}
  __device__ __forceinline__ void g_fin(unsigned vo, double v, double /*q*/, const GT &) { bst(rg, vo, sg, v); }
  __device__ __forceinline__ double q(int i) { return bld(rq, 8u * i, sq); }
  __device__ __forceinline__ void g(int i, double v, double) { bst(rg, 8u * i, sg, v); }
};

// Stage the (S+1) x S walk factor in LDS once per kernel; it stays resident across passes.
// Per-thread registers that never change during a kernel: the day gathered by (wave, lane).
struct PassStatic {
  int d_t, d_a, d_b;        // lane j of wave w: j-th day gathered by the wave and its poll range
};

// Load the per-thread static registers (also used by the out-of-line functions, which cannot
// inherit them from the kernel).
__device__ __forceinline__ PassStatic model_load_static(CMp M) {
  PassStatic pst;
  gcip sc = as_g(M->sched);
  const int slot = threadIdx.x;   // [wave][lane]
  pst.d_t = sc[slot]; pst.d_a = sc[M->c_wda + slot]; pst.d_b = sc[M->c_wdb + slot];
  return pst;
}

__device__ __forceinline__ PassStatic model_setup_lds(CMp M, ldp lds) {
  ldp Lw = lds + M->l_Lw;
  gcdp src = as_g(M->mat);
  for (int i = threadIdx.x; i < M->SE * M->SP; i += PT_THREADS) Lw[i] = src[i];
  {
    // (pseudo-)state of every poll as one byte each: the gathers of phase D read them as LDS broadcasts
    unsigned char AS_L *st = (unsigned char AS_L *)(lds + M->l_st);
    gcip ps = as_g(M->pi);
    const int Npoll = M->Npoll;
    for (int i = threadIdx.x; i < Npoll + 8; i += PT_THREADS) st[i] = i < Npoll ? (unsigned char)ps[i] : (unsigned char)0;
  }
  const PassStatic pst = model_load_static(M);
#ifdef POTUS_PROF
  for (int i = threadIdx.x; i < PT_NPROF; i += PT_THREADS) (lds + M->l_prof)[i] = 0.0;
#endif
  __syncthreads();
  return pst;
}

// One full pass.  Returns lp (log_prob<propto,jacobian>) in every thread; pol.extra[] are
// block-summed alongside.  Ends with a barrier, so LDS may be reused immediately.
template <class Pol>
__device__ __forceinline__ double model_pass(CMp M_in, ldp lds, const PassStatic &pst, Pol &pol_io) {
  Pol pol = pol_io; // private copy: its address never escapes, so it lives in registers
  // The pass is inlined into loops (leaves of a tree, transitions).  Launder the thread id and the
  // descriptor pointer so that nothing derived from them is hoisted out of those loops and then
  // spilled: recomputing an address is cheaper than a scratch round trip.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  unsigned mlo_ = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)M_in);
  unsigned mhi_ = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)M_in >> 32));
  asm volatile("" : "+s"(mlo_), "+s"(mhi_));
  CMp M = (CMp)(((unsigned long long)mhi_ << 32) | mlo_);
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = M->S, T = M->T, SE = M->SE, SP = M->SP, TP = M->TP, full = M->full;
  const int o_Z = M->o_Z, o_c = M->o_c;
  ldp C = lds + M->l_C, Lw = lds + M->l_Lw, X = lds + M->l_X, Y = lds + M->l_Y;
  ldp s_zT = lds + M->l_zT, s_zb = lds + M->l_zb, s_mid = lds + M->l_mid, s_bT = lds + M->l_bT;
  ldp s_pb = lds + M->l_pb, s_e = lds + M->l_e, s_gs = lds + M->l_gs, s_ge = lds + M->l_ge;
  ldp s_scal = lds + M->l_scal, red = lds + M->l_red;
#ifdef POTUS_PROF
  ldp prof = lds + M->l_prof;
#endif
  double lp = 0.0;
  PROF_START();

  // ---------------- phase A: small parameters, then the S x T block with local suffix sums
  {
    // raw_mu_b_T / raw_polling_bias (wave w owns k = w, w+16, ... and starts both mat-vecs) and the
    // parameters between the S x T block and the noise blocks: every load issued up front, no branches
    const int kk = w + PT_NW * lane;
    const bool okz = lane < PT_KJ && kk < S;
    const unsigned vzT = okz ? 8u * (unsigned)(M->o_zT + kk) : PT_OOB, vzb = okz ? 8u * (unsigned)(M->o_zb + kk) : PT_OOB;
    typename Pol::QT qa, qb, qm;
    pol.q_load(vzT, qa);
    pol.q_load(vzb, qb);
    const int nmid = M->nmid, o_mue = M->o_mue, o_rho = M->o_rho;
    const unsigned vmid = tid < nmid ? 8u * (unsigned)(o_c + tid) : PT_OOB;
    pol.q_load(vmid, qm);
    const rsrc_t rm = make_rsrc(M->mat, 8u * (unsigned)(M->m_w + S));
    const unsigned sLT = 8u * (unsigned)M->m_LTt, sLB = 8u * (unsigned)M->m_LBt;
    double lt[PT_KJ], lb[PT_KJ];
#pragma unroll
    for (int j = 0; j < PT_KJ; j++) {
      const int k = w + PT_NW * j;
      const unsigned vo = (k < S && lane < S) ? 8u * (unsigned)(k * S + lane) : PT_OOB;
      lt[j] = bld(rm, vo, sLT);                 // L_T[lane][k], stan:85
      lb[j] = bld(rm, vo, sLB);                 // L_B[lane][k], stan:77
    }
    const double vT = pol.q_fin(vzT, qa), vB = pol.q_fin(vzb, qb); // 0 on masked lanes
    const int kz = okz ? kk : S;                // masked lanes write the spare slot
    s_zT[kz] = vT;
    s_zb[kz] = vB;
    lp -= 0.5 * (vT * vT + vB * vB);            // stan:117,128
    {
      const double v = pol.q_fin(vmid, qm);
      const int idx = o_c + tid;
      s_mid[tid < nmid ? tid : nmid] = v;
      if (!(full && (idx == o_mue || idx == o_rho))) lp -= 0.5 * v * v; // stan:120-122,125
    }
    double pT = 0.0, pB = 0.0;
#pragma unroll
    for (int j = 0; j < PT_KJ; j++) {
      pT += lt[j] * __shfl(vT, j, 64);
      pB += lb[j] * __shfl(vB, j, 64);
    }
    const int sx = lane < S ? lane : S;         // X rows are SE = S+1 wide
    X[w * SE + sx] = pT;
    X[(PT_NW + w) * SE + sx] = pB;
    // Only with more than PT_THREADS such parameters.  Uniform trip count and masked lanes instead of a per-lane loop: the exit
    // block of a divergent loop that no lane enters is where hipcc 7.2 once put the spill of a register that is live across it,
    // executed with an empty EXEC mask (scripts/check_spill_exec.py).
    for (int i0 = PT_THREADS; i0 < nmid; i0 += PT_THREADS) {
      const int i = i0 + tid, idx = o_c + i;
      const bool ok = i < nmid;
      typename Pol::QT qx;
      pol.q_load(ok ? 8u * (unsigned)idx : PT_OOB, qx);
      const double v = pol.q_fin(0u, qx);      // 0 on masked lanes
      s_mid[ok ? i : nmid] = v;
      lp -= (full && (idx == o_mue || idx == o_rho)) ? 0.0 : 0.5 * v * v;
    }
  }
  double cs[PT_CH], zq[PT_CH];   // suffix sums; positions (kept in registers until phase F)
  {
    const int t0 = w * PT_CH;
    typename Pol::QT qt[PT_CH];
#pragma unroll
    for (int j = 0; j < PT_CH; j++) {            // every load of the block issued together, no branches
      const int t = t0 + j;
      pol.q_load((lane < S && t < T) ? 8u * (unsigned)(o_Z + lane + S * t) : PT_OOB, qt[j]);
    }
    double run = 0.0;
#pragma unroll
    for (int j = PT_CH - 1; j >= 0; j--) {
      const int t = t0 + j;
      const double z = pol.q_fin(0u, qt[j]);    // 0 for masked-off elements
      zq[j] = z;
      lp -= 0.5 * z * z;                        // to_vector(raw_mu_b) ~ std_normal(), stan:119
      run += (t < T - 1) ? z : 0.0;             // column T is not part of the walk (stan:86)
      cs[j] = run;
    }
    if (lane < S) Y[w * SE + lane] = run;
  }
  __syncthreads();
  PROF_MARK(0);

  // ---------------- phase B: carries -> C in LDS; bT, polling bias; AR(1) forward
  if (lane < S) {
    double carry = 0.0, cy[PT_NW];
#pragma unroll
    for (int w2 = 0; w2 < PT_NW; w2++) cy[w2] = Y[w2 * SE + lane];
    ISSUE_FENCE();
#pragma unroll
    for (int w2 = 0; w2 < PT_NW; w2++) carry += w2 > w ? cy[w2] : 0.0;
    const int t0 = w * PT_CH;
#pragma unroll
    for (int j = 0; j < PT_CH; j++)
      if (t0 + j < T) C[lane * TP + t0 + j] = cs[j] + carry;
  }
  if (w == PT_NW - 1) {
    double bT = 0.0, pb = 0.0;
    if (lane < S) {
#pragma unroll
      for (int w2 = 0; w2 < PT_NW; w2++) { bT += X[w2 * SE + lane]; pb += X[(PT_NW + w2) * SE + lane]; }
      bT += (as_g(M->mat) + M->m_prior)[lane];
      s_bT[lane] = bT;
      s_pb[lane] = pb;
    }
    const double ww = lane < S ? (as_g(M->mat) + M->m_w)[lane] : 0.0;
    const double nb = dpp_scan_sum(ww * bT), np = dpp_scan_sum(ww * pb);   // stan:79 and the national average of mu_b[:,T]
    if (lane == 63) { s_bT[S] = nb; s_pb[S] = np; }
  }
  if (full && w == PT_NW - 2) {
    // e_bias (stan:91-93) as an affine scan over days: d[t] = e[t]-mu_e, d[t] = rho d[t-1] + sigma_rho z[t]
    const double sigma_e = M->sigma_e;
    ldp ze = s_mid + (M->o_ze - o_c);
    const double xm = s_mid[M->o_mue - o_c], xr = s_mid[M->o_rho - o_c];
    const double mue = 0.02 * xm, rho = d_inv_logit(xr);
    const double srho = sqrt(1.0 - rho * rho) * sigma_e;
    const int per = (T + 63) / 64, ta = lane * per, tb = min(T, ta + per);
    double A = 1.0, B = 0.0;
    for (int t = ta; t < tb; t++) {
      if (t == 0) { A = 0.0; B = ze[0] * sigma_e - mue; }
      else { A = rho * A; B = rho * B + srho * ze[t]; }
    }
    {
      double Bv[1] = {B};
      dpp_scan_affine(A, Bv);                     // affine composites across lanes on the DPP path
      B = Bv[0];
    }
    double d = dpp_prev_lane(B, 0.0);
    for (int t = ta; t < tb; t++) {
      d = (t == 0) ? ze[0] * sigma_e - mue : rho * d + srho * ze[t];
      s_e[t] = d + mue;
    }
    if (lane == 0) {
      s_scal[SC_MUE] = mue; s_scal[SC_RHO] = rho; s_scal[SC_SRHO] = srho; s_scal[SC_XMUE] = xm; s_scal[SC_XRHO] = xr;
      // Jacobians (stan:62-63) and the two informative priors (stan:123-124)
      lp += log(0.02) - 0.5 * xm * xm + log(rho) + log1p(-rho) - 0.5 * ((rho - 0.7) / 0.1) * ((rho - 0.7) / 0.1);
    }
  }
  PROF_SUB(26);
  __syncthreads();
  PROF_MARK(1);

  // ---------------- phase C: one thread per poll (stan:95-112, 130-131)
  ldp r_lds = X;
  if (tid == 0) r_lds[M->Npoll] = 0.0;
  {
    const unsigned Npad = M->Npad;
    const rsrc_t rpi = make_rsrc(M->pi, 6u * Npad * 4u), rpd = make_rsrc(M->pd, 4u * Npad * 8u);
    const int Npoll = M->Npoll, om = M->o_m - o_c, opop = M->o_pop - o_c;
    const double sigma_c = M->sigma_c, sigma_m = M->sigma_m, sigma_pop = M->sigma_pop;
    for (int i0 = 0; i0 < Npoll; i0 += 2 * PT_THREADS) {   // two polls per thread per trip, no branches inside
      int s[2], t[2], ip[2], qi[2], im[2], ipop[2];
      double un[2], y[2], N[2], sg[2];
      unsigned vq[2];
      typename Pol::QT qt[2];
      typename Pol::GT gt[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = i0 + u * PT_THREADS + tid;
        const bool ok = i < Npoll;
        const unsigned vi = ok ? 4u * i : PT_OOB, vd = ok ? 8u * i : PT_OOB;   // masked polls read zeros: N = y = 0
        s[u] = bld_i(rpi, vi, 0); t[u] = bld_i(rpi, vi, 4u * Npad); ip[u] = bld_i(rpi, vi, 8u * Npad); qi[u] = bld_i(rpi, vi, 20u * Npad);
        im[u] = 0; ipop[u] = 0; un[u] = 0.0;
        if (full) { im[u] = bld_i(rpi, vi, 12u * Npad); ipop[u] = bld_i(rpi, vi, 16u * Npad); un[u] = bld(rpd, vd, 16u * Npad); }
        y[u] = bld(rpd, vd, 0); N[u] = bld(rpd, vd, 8u * Npad); sg[u] = bld(rpd, vd, 24u * Npad);
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = i0 + u * PT_THREADS + tid;
        vq[u] = (i < Npoll) ? 8u * (unsigned)qi[u] : PT_OOB;
        pol.q_load(vq[u], qt[u]);
        pol.g_load(vq[u], gt[u]);
      }
      ldp L0 = Lw + s[0] * SP, C0 = C + t[0], L1 = Lw + s[1] * SP, C1 = C + t[1];
      double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
      for (int k0 = 0; k0 < S; k0 += 8) {        // 51-term dots, eight terms of both polls in flight at once
        double l0[8], c0[8], l1[8], c1[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int kk = min(k0 + j, S - 1);
          l0[j] = L0[kk]; c0[j] = C0[kk * TP]; l1[j] = L1[kk]; c1[j] = C1[kk * TP];
        }
        ISSUE_FENCE();
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const double m0 = k0 + j < S ? 1.0 : 0.0, m1 = k0 + j + 1 < S ? 1.0 : 0.0;
          a0 += m0 * l0[j] * c0[j];         b0 += m0 * l1[j] * c1[j];
          a1 += m1 * l0[j + 1] * c0[j + 1]; b1 += m1 * l1[j + 1] * c1[j + 1];
        }
      }
      const double dot[2] = {a0 + a1, b0 + b1};
      PROF_SUB(18);
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = i0 + u * PT_THREADS + tid;
        const double zn = pol.q_fin(vq[u], qt[u]);
        double eta = s_bT[s[u]] + s_pb[s[u]] + sg[u] * zn + sigma_c * s_mid[ip[u]] + dot[u];
        if (full) eta += sigma_m * s_mid[om + im[u]] + sigma_pop * s_mid[opop + ipop[u]] + un[u] * s_e[t[u]];
        // binomial_logit with one exp, one log1p, one division:  e = exp(-|eta|), l = log1p(e)
        //   log inv_logit(eta) = min(eta,0) - l,  log inv_logit(-eta) = min(-eta,0) - l,
        //   inv_logit(eta) = (eta >= 0 ? 1 : e) / (1 + e)
        const double ex = exp(-fabs(eta)), l1 = log1p(ex), pr = (eta >= 0.0 ? 1.0 : ex) / (1.0 + ex);
        const double r = y[u] - N[u] * pr;
        lp += y[u] * (fmin(eta, 0.0) - l1) + (N[u] - y[u]) * (fmin(-eta, 0.0) - l1) - 0.5 * zn * zn; // stan:126-127,130-131
        r_lds[i < Npoll ? i : Npoll + 1] = r;    // slot Npoll stays 0 (padding of the task lists), Npoll+1 is a dump
        pol.g_fin(vq[u], sg[u] * r - zn, zn, gt[u]);
      }
    }
  }
  PROF_SUB(25);
  __syncthreads();
  PROF_MARK(2);

  // ---------------- phase D: per-day gathers gC[:,t] = sum_i r_i Lw_ext[s_i,:]; level-1 segment sums
  // level-2 metadata for phase E is requested first so that its latency hides behind the gathers
  const rsrc_t rsc = make_rsrc(M->sched, 0x7ffffff0u);
  int sg_a = 0, sg_b = 0, sg_kind = 1, sg_index = 0;
  double sg_scale = 0.0;
  typename Pol::GT sg_gt;
  unsigned sg_vg = PT_OOB;
  {
    const int nseg = M->nseg;
    const unsigned vs = tid < nseg ? 4u * (unsigned)tid : PT_OOB;
    sg_a = bld_i(rsc, vs, 4u * (unsigned)M->c_segptr);
    sg_b = bld_i(rsc, vs, 4u * (unsigned)M->c_segptr + 4u);
    sg_kind = tid < nseg ? bld_i(rsc, vs, 4u * (unsigned)M->c_segkind) : 1;
    sg_index = bld_i(rsc, vs, 4u * (unsigned)M->c_segidx);
    const rsrc_t rss = make_rsrc(M->seg_scale, 8u * (unsigned)nseg);
    sg_scale = bld(rss, tid < nseg ? 8u * (unsigned)tid : PT_OOB, 0);
    sg_vg = (tid < nseg && sg_kind == 0) ? 8u * (unsigned)sg_index : PT_OOB;
    pol.g_load(sg_vg, sg_gt);
  }
  PROF_SUB(19);
  {
    // lanes = k; the wave walks the days it owns (day and poll range sit in registers, one day per
    // lane, broadcast with readlane); poll state and residual are LDS broadcasts: no memory latency
    const unsigned char AS_L *st = (const unsigned char AS_L *)(lds + M->l_st);
    const int lk = lane < S ? lane : 0, Npoll = M->Npoll;
    for (int dj = 0; dj < 64; dj++) {
      const int a = __builtin_amdgcn_readlane(pst.d_a, dj), b = __builtin_amdgcn_readlane(pst.d_b, dj);
      if (a >= b) break;                          // a wave's days are packed at the front
      const int t = __builtin_amdgcn_readlane(pst.d_t, dj);
      double acc0 = 0.0, acc1 = 0.0;
      for (int i = a; i < b; i += 8) {            // eight polls per trip, reads beyond b hit the zero slot
        int s8[8];
        double r8[8], l8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int ii = i + u < b ? i + u : Npoll; s8[u] = st[ii]; r8[u] = r_lds[ii]; }
        ISSUE_FENCE();
#pragma unroll
        for (int u = 0; u < 8; u++) l8[u] = Lw[s8[u] * SP + lk];
        ISSUE_FENCE();
#pragma unroll
        for (int u = 0; u < 8; u += 2) { acc0 += r8[u] * l8[u]; acc1 += r8[u + 1] * l8[u + 1]; }
      }
      if (lane < S) C[lane * TP + t] = acc0 + acc1;
    }
  }
  PROF_SUB(20);
  {
    const int nsub = M->nsub, wb = M->sub_weighted_begin;
    const unsigned o_sub = 4u * (unsigned)M->c_sub16;
    const rsrc_t rwt = make_rsrc(M->sub_wt16, 8u * 16u * (unsigned)(nsub - wb) + 16u);
    for (int sub0 = 0; sub0 < nsub; sub0 += PT_THREADS) {
      const int sub = sub0 + tid;
      const bool ok = sub < nsub, wtd = ok && sub >= wb;
      const unsigned vs = ok ? 64u * (unsigned)sub : PT_OOB, vw = wtd ? 128u * (unsigned)(sub - wb) : PT_OOB;
      u32x4 ix[4];
      double wt[PT_SUBLEN];
#pragma unroll
      for (int j = 0; j < 4; j++) ix[j] = bld_i4(rsc, vs + 16u * j, o_sub);
#pragma unroll
      for (int j = 0; j < PT_SUBLEN / 2; j++) bld_d2(rwt, vw + 16u * j, 0, wt[2 * j], wt[2 * j + 1]);
      double rr[PT_SUBLEN];
#pragma unroll
      for (int j = 0; j < PT_SUBLEN; j++) rr[j] = r_lds[ok ? ix[j >> 2][j & 3] : 0];
      ISSUE_FENCE();
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < PT_SUBLEN; j++) sum += wtd ? rr[j] * wt[j] : rr[j];
      if (ok) Y[sub] = sum;
    }
  }
  PROF_SUB(21);
  __syncthreads();
  PROF_MARK(3);

  // ---------------- phase E: local prefix sums of gC; level-2 segment sums
  double pre[PT_CH];
  {
    const int t0 = w * PT_CH;
    const unsigned mask = (unsigned)(as_c(M->sched) + M->c_mask)[w];   // days of this wave that have polls
    double run = 0.0;
#pragma unroll
    for (int j = 0; j < PT_CH; j++) {
      const int t = t0 + j;
      if (((mask >> j) & 1u) && lane < S && t < T - 1) run += C[lane * TP + t];
      pre[j] = run;
    }
    if (lane < S) X[w * SE + lane] = run;  // r_lds is dead from here on
  }
  {
    double sum = 0.0;
    for (int j0 = sg_a; j0 < sg_b; j0 += 8) {
      double yy[8];
#pragma unroll
      for (int u = 0; u < 8; u++) yy[u] = Y[min(j0 + u, sg_b - 1)];
      ISSUE_FENCE();
#pragma unroll
      for (int u = 0; u < 8; u++) sum += j0 + u < sg_b ? yy[u] : 0.0;
    }
    if (sg_kind == 0) { const double qv = s_mid[sg_index - o_c]; pol.g_fin(sg_vg, sg_scale * sum - qv, qv, sg_gt); }
    else if (tid < M->nseg) { if (sg_kind == 1) s_gs[sg_index] = sum; else s_ge[sg_index] = sum; }
    const int nseg = M->nseg;
    for (int seg = tid + PT_THREADS; seg < nseg; seg += PT_THREADS) {   // only with more than 1024 segments
      const int AS_C *sc = as_c(M->sched);
      const int a = sc[M->c_segptr + seg], b = sc[M->c_segptr + seg + 1], kind = sc[M->c_segkind + seg], index = sc[M->c_segidx + seg];
      double s2 = 0.0;
      for (int j = a; j < b; j++) s2 += Y[j];
      if (kind == 0) { const double qv = s_mid[index - o_c]; pol.g(index, as_g(M->seg_scale)[seg] * s2 - qv, qv); }
      else if (kind == 1) s_gs[index] = s2;
      else s_ge[index] = s2;
    }
  }
  PROF_SUB(24);
  __syncthreads();
  PROF_MARK(4);

  // ---------------- phase F: dZ; transposed mat-vecs; AR(1) adjoint
  if (lane < S) {
    double carry = 0.0, cy[PT_NW];
#pragma unroll
    for (int w2 = 0; w2 < PT_NW; w2++) cy[w2] = X[w2 * SE + lane];
    ISSUE_FENCE();
#pragma unroll
    for (int w2 = 0; w2 < PT_NW; w2++) carry += w2 < w ? cy[w2] : 0.0;
    const int t0 = w * PT_CH;
    // two batches in flight: the operands of batch b + 1 are requested before batch b is finished, so a batch does not wait a
    // full memory round trip with nothing outstanding (-2.5 % per leaf at 256 chains, -1 % at 8: profiles/r03_one_workgroup_leaf.txt)
    constexpr int GB = Pol::GB, NB = PT_CH / GB;
    typename Pol::GT gt[2][GB];
    auto off = [&](int h, int j) { const int t = t0 + h + j; return (t < T) ? 8u * (unsigned)(o_Z + lane + S * t) : PT_OOB; };
#pragma unroll
    for (int j = 0; j < GB; j++) pol.g_load(off(0, j), gt[0][j]);
#pragma unroll
    for (int b = 0; b < NB; b++) {
      if (b + 1 < NB) {
#pragma unroll
        for (int j = 0; j < GB; j++) pol.g_load(off((b + 1) * GB, j), gt[(b + 1) & 1][j]);
      }
      ISSUE_FENCE();
#pragma unroll
      for (int j = 0; j < GB; j++) {
        const int h = b * GB, t = t0 + h + j;
        pol.g_fin(off(h, j), (t < T - 1 ? pre[h + j] + carry : 0.0) - zq[h + j], zq[h + j], gt[b & 1][j]);
      }
    }
  }
  PROF_SUB(22);
  {
    // dbT[s] = dpolling_bias[s] = residuals of state s + w_s * national residuals; C region is free now
    gcdp LT = as_g(M->mat) + M->m_LT, LB = as_g(M->mat) + M->m_LB, wv = as_g(M->mat) + M->m_w;
    double pT = 0.0, pB = 0.0;
    const double gnat = s_gs[S];
#pragma unroll
    for (int j = 0; j < PT_KJ; j++) {
      const int s = w + PT_NW * j;
      if (s < S && lane < S) {
        const double G = s_gs[s] + wv[s] * gnat;
        pT += LT[s * S + lane] * G;
        pB += LB[s * S + lane] * G;
      }
    }
    if (lane < S) { C[w * SE + lane] = pT; C[(PT_NW + w) * SE + lane] = pB; }
  }
  if (full && w == PT_NW - 2) {
    // adjoint recursion a[t] = ge[t] + rho a[t+1]
    const double sigma_e = M->sigma_e;
    const int o_ze = M->o_ze;
    ldp ze = s_mid + (o_ze - o_c);
    const double mue = s_scal[SC_MUE], rho = s_scal[SC_RHO], srho = s_scal[SC_SRHO];
    const int per = (T + 63) / 64, ta = lane * per, tb = min(T, ta + per);
    double A = 1.0, B = 0.0;
    for (int t = tb - 1; t >= ta; t--) { A = rho * A; B = rho * B + s_ge[t]; }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double A2 = __shfl_down(A, off, 64), B2 = __shfl_down(B, off, 64);
      if (lane + off < 64) { B = A * B2 + B; A = A * A2; }
    }
    double a = __shfl_down(B, 1, 64);
    if (lane == 63) a = 0.0;
    double S1 = 0.0, S2 = 0.0, S3 = 0.0;
    constexpr int PER = 4;                       // T <= 256 -> at most four days per lane
    typename Pol::GT gz[PER];
    unsigned vz[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) { vz[u] = ta + u < tb ? 8u * (unsigned)(o_ze + ta + u) : PT_OOB; pol.g_load(vz[u], gz[u]); }
#pragma unroll
    for (int u = PER - 1; u >= 0; u--) {         // no branches: masked days add nothing and their stores are dropped
      const int t = ta + u;
      const bool ok = t < tb;
      const int tc = ok ? t : 0;
      const double a_new = s_ge[tc] + rho * a;
      a = ok ? a_new : a;
      const double z = ze[tc], dprev = s_e[tc >= 1 ? tc - 1 : 0] - mue;
      const bool inner = ok && t >= 1;
      S1 += inner ? a : 0.0; S2 += inner ? a * dprev : 0.0; S3 += inner ? a * z : 0.0;
      pol.g_fin(vz[u], a * (t >= 1 ? srho : sigma_e) - z, z, gz[u]);
    }
    S1 = dpp_scan_sum(S1); S2 = dpp_scan_sum(S2); S3 = dpp_scan_sum(S3);
    if (lane == 63) {
      const double xm = s_scal[SC_XMUE], xr = s_scal[SC_XRHO];
      const double adj_rho = S2 + S3 * sigma_e * (-rho / sqrt(1.0 - rho * rho));
      pol.g(M->o_mue, 0.02 * (1.0 - rho) * S1 - xm, xm);
      pol.g(M->o_rho, (adj_rho - (rho - 0.7) / 0.01) * rho * (1.0 - rho) + (1.0 - 2.0 * rho), xr);
    }
  }
  PROF_SUB(23);
  __syncthreads();
  PROF_MARK(5);

  // ---------------- phase G: finish the two transposed mat-vecs; reduce lp
  if (tid < 128) {
    const int which = tid >> 6;
    if (lane < S) {
      double sum = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < PT_NW; w2++) sum += C[(which * PT_NW + w2) * SE + lane];
      if (which == 0) pol.g(M->o_zT + lane, sum - s_zT[lane], s_zT[lane]);
      else pol.g(M->o_zb + lane, sum - s_zb[lane], s_zb[lane]);
    }
  }
  double v[1 + Pol::NEXTRA];
  v[0] = lp;
#pragma unroll
  for (int k = 0; k < Pol::NEXTRA; k++) v[1 + k] = pol.extra[k];
  block_sum(v, red, tid);
#pragma unroll
  for (int k = 0; k < Pol::NEXTRA; k++) pol_io.extra[k] = v[1 + k];
  PROF_MARK(6);
  return v[0];
}
