// potus_model.hpp -- device-side log-density + gradient of the poll model for gfx950.
//
// One 1024-thread workgroup (16 wave64) evaluates log_prob and its gradient for ONE chain.
// The model is scripts/model/poll_model_2020.stan:56-131 (and the no_mode_adjustment
// variant), re-derived for the hardware instead of transcribed:
//
//  * the T-1 dependent 51x51 mat-vecs of stan:86 become a suffix sum over days,
//      mu_b[:,t] = (L_T z_T + prior) + L_W * C[:,t],  C[:,t] = sum_{u=t}^{T-2} Z[:,u]
//    held in LDS as C[k][t]; wave w owns days [16w,16w+16), lane k owns state k, so the S x T
//    block is read from HBM/L2 with lanes on consecutive addresses (coalesced) and scanned in
//    registers with one 16-entry carry exchange through LDS;
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
  double sigma_c, sigma_m, sigma_pop, sigma_e;
  // Static data is packed into three buffers so that the descriptor costs few scalar registers.
  const double *mat;        // Lw_ext [SE][SP] | LT_t [k][s] | LB_t [k][s] | LT [s][k] | LB [s][k] | prior [S] | w [S]
  int m_LTt, m_LBt, m_LT, m_LB, m_prior, m_w;   // offsets into mat (Lw_ext at 0)
  //   Lw_ext rows 0..S-1: L_W (zeros above the diagonal); row S: v = L_W^T w
  const int *pi;            // polls sorted by day, struct of arrays with stride Npad:
  const double *pd;         //   pi: ps | pt | pp | pm | ppop | pqidx      pd: py | pn | punadj | psig
  int Npad;
  // Static schedule, built on the host (all offsets in ints into sched):
  //   we_ptr [NW+1]          entries of wave w: [we_ptr[w], we_ptr[w+1])
  //   we_idx | we_day | we_state   per entry: sorted poll index, its day, its (pseudo-)state; a wave's
  //                          entries are grouped by day, days balanced over waves (LPT)
  //   daymask [NW]           bit j of word w: day 16w+j has at least one poll
  //   sub16 [nsub][16]       level-1 segment-sum tasks: poll indices, padded with Npoll (a zero slot)
  //   seg_ptr [nseg+1] | seg_kind | seg_index   level-2: range of tasks, what the sum feeds
  const int *sched;
  int c_weidx, c_weday, c_west, c_mask, c_sub16, c_segptr, c_segkind, c_segidx;
  const double *seg_scale;  // [nseg]
  const double *sub_wt16;   // [(nsub - sub_weighted_begin)][16] weights of the weighted tasks (0 for padding)
  int nsub, nseg, sub_weighted_begin;
  // LDS layout, offsets in doubles
  int l_C, l_Lw, l_X, l_Y, l_zT, l_zb, l_mid, l_bT, l_pb, l_e, l_gs, l_ge, l_scal, l_red, l_prof;
  int lds_doubles;
};
typedef const DevModel AS_C *CMp; // the model descriptor lives in device memory, read through scalar loads

// scalar slots in LDS (l_scal)
enum { SC_MUE = 0, SC_RHO, SC_SRHO, SC_XMUE, SC_XRHO, SC_N };
#define PT_NRED 8  // reduction slots
#define PT_NPROF 32

#ifdef POTUS_PROF
#define PROF_MARK(k) do { if (threadIdx.x == 0) { const long long t_ = clock64(); prof[k] += (double)(t_ - (long long)prof[PT_NPROF - 1]); prof[PT_NPROF - 1] = (double)t_; } } while (0)
#define PROF_START() do { if (threadIdx.x == 0) prof[PT_NPROF - 1] = (double)clock64(); } while (0)
#else
#define PROF_MARK(k) do { } while (0)
#define PROF_START() do { } while (0)
#endif

__device__ __forceinline__ double d_log_inv_logit(double x) { return x > 0 ? -log1p(exp(-x)) : x - log1p(exp(x)); }
__device__ __forceinline__ double d_inv_logit(double x) {
  if (x >= 0) return 1.0 / (1.0 + exp(-x));
  double e = exp(x);
  return e / (1.0 + e);
}

// Sum N values over the workgroup; every thread returns with the totals.  Fixed order.
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], ldp red, int tid) {
  const int lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int k = 0; k < N; k++) {
    double x = v[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) red[w * N + k] = x;
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
// of loads before the first dependent store); g_load/g_fin consume one gradient element;
// g_load_q additionally returns the position again (t.q).
// All element accesses take the element's BYTE offset (8*index) or PT_OOB: buffer loads beyond the
// resource return 0 and stores are dropped, so masked-off elements need no branch -- which matters,
// because the compiler drains the whole memory queue (s_waitcnt vmcnt(0)) at every branch join.
#define PT_OOB 0xFFFFFF00u
struct PlainPolicy {
  rsrc_t rq, rg;        // buffers holding q and receiving grad (may be the same buffer)
  unsigned sq, sg;      // byte offsets of the two vectors inside them
  static constexpr int NEXTRA = 0;
  static constexpr int QB = 16;
  double extra[1];  // unused
  struct QT { double q; };
  struct GT { double q; };
  __device__ __forceinline__ void q_load(unsigned vo, QT &t) { t.q = bld(rq, vo, sq); }
  __device__ __forceinline__ double q_fin(unsigned, QT &t) { return t.q; }
  __device__ __forceinline__ void g_load(unsigned, GT &) {}
  __device__ __forceinline__ void g_load_q(unsigned vo, GT &t) { t.q = bld(rq, vo, sq); }
  __device__ __forceinline__ void g_from_q(const QT &, GT &) {}   // same thread did q_load earlier in the pass
  __device__ __forceinline__ void g_fin(unsigned vo, double v, const GT &) { bst(rg, vo, sg, v); }
  __device__ __forceinline__ double q(int i) { return bld(rq, 8u * i, sq); }
  __device__ __forceinline__ void g(int i, double v) { bst(rg, 8u * i, sg, v); }
};

// Stage the (S+1) x S walk factor in LDS once per kernel; it stays resident across passes.
__device__ __forceinline__ void model_setup_lds(CMp M, ldp lds) {
  ldp Lw = lds + M->l_Lw;
  gcdp src = as_g(M->mat);
  for (int i = threadIdx.x; i < M->SE * M->SP; i += PT_THREADS) Lw[i] = src[i];
#ifdef POTUS_PROF
  for (int i = threadIdx.x; i < PT_NPROF; i += PT_THREADS) (lds + M->l_prof)[i] = 0.0;
#endif
  __syncthreads();
}

// One full pass.  Returns lp (log_prob<propto,jacobian>) in every thread; pol.extra[] are
// block-summed alongside.  Ends with a barrier, so LDS may be reused immediately.
template <class Pol>
__device__ __forceinline__ double model_pass(CMp M_in, ldp lds, Pol &pol_io) {
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
    for (int i = tid + PT_THREADS; i < nmid; i += PT_THREADS) { // only when there are more than 1024 such parameters
      const int idx = o_c + i;
      const double v = pol.q(idx);
      s_mid[i] = v;
      if (!(full && (idx == o_mue || idx == o_rho))) lp -= 0.5 * v * v;
    }
  }
  double cs[PT_CH];
  {
    const int t0 = w * PT_CH;
    double run = 0.0;
#pragma unroll
    for (int h = PT_CH - Pol::QB; h >= 0; h -= Pol::QB) { // batches from the last day backwards; no branches inside
      typename Pol::QT qt[Pol::QB];
      unsigned vo[Pol::QB];
#pragma unroll
      for (int j = 0; j < Pol::QB; j++) {
        const int t = t0 + h + j;
        vo[j] = (lane < S && t < T) ? 8u * (unsigned)(o_Z + lane + S * t) : PT_OOB;
        pol.q_load(vo[j], qt[j]);
      }
#pragma unroll
      for (int j = Pol::QB - 1; j >= 0; j--) {
        const int t = t0 + h + j;
        const double z = pol.q_fin(vo[j], qt[j]);  // 0 for masked-off elements
        lp -= 0.5 * z * z;                      // to_vector(raw_mu_b) ~ std_normal(), stan:119
        run += (t < T - 1) ? z : 0.0;           // column T is not part of the walk (stan:86)
        cs[h + j] = run;
      }
    }
    if (lane < S) Y[w * SE + lane] = run;
  }
  __syncthreads();
  PROF_MARK(0);

  // ---------------- phase B: carries -> C in LDS; bT, polling bias; AR(1) forward
  if (lane < S) {
    double carry = 0.0;
    for (int w2 = w + 1; w2 < PT_NW; w2++) carry += Y[w2 * SE + lane];
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
    double nb = ww * bT, np = ww * pb;          // stan:79 and the national average of mu_b[:,T]
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { nb += __shfl_down(nb, off, 64); np += __shfl_down(np, off, 64); }
    if (lane == 0) { s_bT[S] = nb; s_pb[S] = np; }
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
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double A2 = __shfl_up(A, off, 64), B2 = __shfl_up(B, off, 64);
      if (lane >= off) { B = A * B2 + B; A = A * A2; }
    }
    double d = __shfl_up(B, 1, 64);
    if (lane == 0) d = 0.0;
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
      }
      ldp L0 = Lw + s[0] * SP, C0 = C + t[0], L1 = Lw + s[1] * SP, C1 = C + t[1];
      double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
      int k = 0;
      for (; k + 1 < S; k += 2) {
        a0 += L0[k] * C0[k * TP];
        b0 += L1[k] * C1[k * TP];
        a1 += L0[k + 1] * C0[(k + 1) * TP];
        b1 += L1[k + 1] * C1[(k + 1) * TP];
      }
      for (; k < S; k++) { a0 += L0[k] * C0[k * TP]; b0 += L1[k] * C1[k * TP]; }
      const double dot[2] = {a0 + a1, b0 + b1};
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
        typename Pol::GT gt;
        pol.g_from_q(qt[u], gt);
        pol.g_fin(vq[u], sg[u] * r - zn, gt);
      }
    }
  }
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
  {
    const int AS_C *we_ptr = as_c(M->sched);
    const unsigned o_idx = 4u * (unsigned)M->c_weidx, o_day = 4u * (unsigned)M->c_weday, o_st = 4u * (unsigned)M->c_west;
    const int e0 = we_ptr[w], e1 = we_ptr[w + 1], Npoll = M->Npoll;
    const int lk = lane < S ? lane : 0;
    double acc = 0.0;
    int cur_t = -1;
    for (int c0 = e0; c0 < e1; c0 += 64) {     // this wave's polls, 64 at a time: one per lane, then broadcast
      const int nb = min(64, e1 - c0);
      const unsigned ve = lane < nb ? 4u * (unsigned)(c0 + lane) : PT_OOB;
      const int ei = bld_i(rsc, ve, o_idx), et = bld_i(rsc, ve, o_day), es = bld_i(rsc, ve, o_st);
      const double rv = r_lds[lane < nb ? ei : Npoll];
      const int rlo = __double2loint(rv), rhi = __double2hiint(rv);
      for (int j = 0; j < nb; j++) {
        const int t = __builtin_amdgcn_readlane(et, j), s = __builtin_amdgcn_readlane(es, j);
        const double r = __hiloint2double(__builtin_amdgcn_readlane(rhi, j), __builtin_amdgcn_readlane(rlo, j));
        if (t != cur_t) {
          if (cur_t >= 0 && lane < S) C[lane * TP + cur_t] = acc;
          acc = 0.0;
          cur_t = t;
        }
        acc += r * Lw[s * SP + lk];
      }
    }
    if (cur_t >= 0 && lane < S) C[lane * TP + cur_t] = acc;
  }
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
      double sum = 0.0;
      if (wtd) {
#pragma unroll
        for (int j = 0; j < PT_SUBLEN; j++) sum += r_lds[ix[j >> 2][j & 3]] * wt[j];
      } else {
#pragma unroll
        for (int j = 0; j < PT_SUBLEN; j++) sum += r_lds[ok ? ix[j >> 2][j & 3] : 0];
      }
      if (ok) Y[sub] = sum;
    }
  }
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
    for (int j = sg_a; j < sg_b; j++) sum += Y[j];
    if (sg_kind == 0) pol.g_fin(sg_vg, sg_scale * sum - s_mid[sg_index - o_c], sg_gt);
    else if (tid < M->nseg) { if (sg_kind == 1) s_gs[sg_index] = sum; else s_ge[sg_index] = sum; }
    const int nseg = M->nseg;
    for (int seg = tid + PT_THREADS; seg < nseg; seg += PT_THREADS) {   // only with more than 1024 segments
      const int AS_C *sc = as_c(M->sched);
      const int a = sc[M->c_segptr + seg], b = sc[M->c_segptr + seg + 1], kind = sc[M->c_segkind + seg], index = sc[M->c_segidx + seg];
      double s2 = 0.0;
      for (int j = a; j < b; j++) s2 += Y[j];
      if (kind == 0) pol.g(index, as_g(M->seg_scale)[seg] * s2 - s_mid[index - o_c]);
      else if (kind == 1) s_gs[index] = s2;
      else s_ge[index] = s2;
    }
  }
  __syncthreads();
  PROF_MARK(4);

  // ---------------- phase F: dZ; transposed mat-vecs; AR(1) adjoint
  if (lane < S) {
    double carry = 0.0;
    for (int w2 = 0; w2 < w; w2++) carry += X[w2 * SE + lane];
    const int t0 = w * PT_CH;
#pragma unroll
    for (int h = 0; h < PT_CH; h += Pol::QB) {
      typename Pol::GT gt[Pol::QB];
      unsigned vo[Pol::QB];
#pragma unroll
      for (int j = 0; j < Pol::QB; j++) {
        const int t = t0 + h + j;
        vo[j] = (t < T) ? 8u * (unsigned)(o_Z + lane + S * t) : PT_OOB;
        pol.g_load_q(vo[j], gt[j]);
      }
#pragma unroll
      for (int j = 0; j < Pol::QB; j++) {
        const int t = t0 + h + j;
        pol.g_fin(vo[j], (t < T - 1 ? pre[h + j] + carry : 0.0) - gt[j].q, gt[j]);
      }
    }
  }
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
    for (int t = tb - 1; t >= ta; t--) {
      a = s_ge[t] + rho * a;
      if (t >= 1) {
        S1 += a; S2 += a * (s_e[t - 1] - mue); S3 += a * ze[t];
        pol.g(o_ze + t, a * srho - ze[t]);
      } else {
        pol.g(o_ze, a * sigma_e - ze[0]);
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      S1 += __shfl_down(S1, off, 64); S2 += __shfl_down(S2, off, 64); S3 += __shfl_down(S3, off, 64);
    }
    if (lane == 0) {
      const double xm = s_scal[SC_XMUE];
      const double adj_rho = S2 + S3 * sigma_e * (-rho / sqrt(1.0 - rho * rho));
      pol.g(M->o_mue, 0.02 * (1.0 - rho) * S1 - xm);
      pol.g(M->o_rho, (adj_rho - (rho - 0.7) / 0.01) * rho * (1.0 - rho) + (1.0 - 2.0 * rho));
    }
  }
  __syncthreads();
  PROF_MARK(5);

  // ---------------- phase G: finish the two transposed mat-vecs; reduce lp
  if (tid < 128) {
    const int which = tid >> 6;
    if (lane < S) {
      double sum = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < PT_NW; w2++) sum += C[(which * PT_NW + w2) * SE + lane];
      if (which == 0) pol.g(M->o_zT + lane, sum - s_zT[lane]);
      else pol.g(M->o_zb + lane, sum - s_zb[lane]);
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
