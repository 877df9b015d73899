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

#define PT_THREADS 1024
#define PT_NW 16          // waves per workgroup
#define PT_CH 16          // days per wave in the S x T block layout  (T <= PT_NW*PT_CH)
#define PT_SUBLEN 16      // entries per level-1 segment-sum task
#define PT_QB 8           // S x T elements per thread whose loads are issued together

#define AS_G __attribute__((address_space(1)))
#define AS_L __attribute__((address_space(3)))
#define AS_C __attribute__((address_space(4)))
typedef double AS_G *gdp;
typedef const double AS_G *gcdp;
typedef const int AS_G *gcip;
typedef double AS_L *ldp;
template <class T> __device__ __forceinline__ T AS_G *as_g(T *p) { return (T AS_G *)p; }

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
  const int *sched;         // day_ptr [T+1] | wave_task_ptr [NW+1] | task_day [T] | sub_ptr | sub_idx | seg_ptr | seg_kind | seg_index
  int c_wtp, c_td, c_subptr, c_subidx, c_segptr, c_segkind, c_segidx;  // offsets into sched (day_ptr at 0)
  const double *seg_scale;
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
struct PlainPolicy {
  gcdp q_;
  gdp g_;
  static constexpr int NEXTRA = 0;
  double extra[1];  // unused
  struct QT { double q; };
  struct GT { double q; };
  __device__ __forceinline__ void q_load(int i, QT &t) { t.q = q_[i]; }
  __device__ __forceinline__ double q_fin(int, const QT &t) { return t.q; }
  __device__ __forceinline__ void g_load(int, GT &) {}
  __device__ __forceinline__ void g_load_q(int i, GT &t) { t.q = q_[i]; }
  __device__ __forceinline__ void g_fin(int i, double v, const GT &) { g_[i] = v; }
  __device__ __forceinline__ double q(int i) { return q_[i]; }
  __device__ __forceinline__ void g(int i, double v) { g_[i] = v; }
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
    // raw_mu_b_T / raw_polling_bias: wave w owns k = w, w+16, ... and starts both mat-vecs
    double vT = 0.0, vB = 0.0;
    const int kk = w + PT_NW * lane;
    if (lane < 4 && kk < S) {
      vT = pol.q(M->o_zT + kk);
      vB = pol.q(M->o_zb + kk);
      s_zT[kk] = vT;
      s_zb[kk] = vB;
      lp -= 0.5 * (vT * vT + vB * vB);          // stan:117,128
    }
    gcdp LT_t = as_g(M->mat) + M->m_LTt, LB_t = as_g(M->mat) + M->m_LBt;
    double pT = 0.0, pB = 0.0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = w + PT_NW * j;
      const double a = __shfl(vT, j, 64), b = __shfl(vB, j, 64);
      if (k < S && lane < S) {
        pT += LT_t[k * S + lane] * a;           // stan:85
        pB += LB_t[k * S + lane] * b;           // stan:77
      }
    }
    if (lane < S) { X[w * SE + lane] = pT; X[(PT_NW + w) * SE + lane] = pB; }
  }
  {
    const int nmid = M->nmid, o_mue = M->o_mue, o_rho = M->o_rho;
    for (int i = tid; i < nmid; i += PT_THREADS) {
      const int idx = o_c + i;
      const double v = pol.q(idx);
      s_mid[i] = v;
      if (!(full && (idx == o_mue || idx == o_rho))) lp -= 0.5 * v * v; // stan:120-122,125
    }
  }
  double cs[PT_CH];
  {
    const int t0 = w * PT_CH;
    double run = 0.0;
#pragma unroll
    for (int h = PT_CH - PT_QB; h >= 0; h -= PT_QB) { // batches from the last day backwards: loads of a batch issue before its stores
      typename Pol::QT qt[PT_QB];
#pragma unroll
      for (int j = 0; j < PT_QB; j++) if (lane < S && t0 + h + j < T) pol.q_load(o_Z + lane + S * (t0 + h + j), qt[j]);
#pragma unroll
      for (int j = PT_QB - 1; j >= 0; j--) {
        const int t = t0 + h + j;
        const double z = (lane < S && t < T) ? pol.q_fin(o_Z + lane + S * t, qt[j]) : 0.0;
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
  {
    const int Npad = M->Npad;
    gcip ps = as_g(M->pi), pt = ps + Npad, pp = ps + 2 * Npad, pm = ps + 3 * Npad, ppop = ps + 4 * Npad, pqidx = ps + 5 * Npad;
    gcdp py = as_g(M->pd), pn = py + Npad, punadj = py + 2 * Npad, psig = py + 3 * Npad;
    const int Npoll = M->Npoll, om = M->o_m - o_c, opop = M->o_pop - o_c;
    const double sigma_c = M->sigma_c, sigma_m = M->sigma_m, sigma_pop = M->sigma_pop;
    for (int i = tid; i < Npoll; i += PT_THREADS) {
      const int s = ps[i], t = pt[i], qi = pqidx[i], ip = pp[i];
      const double y = py[i], N = pn[i], sg = psig[i];
      const double zn = pol.q(qi);
      double eta = s_bT[s] + s_pb[s] + sg * zn + sigma_c * s_mid[ip];
      if (full) eta += sigma_m * s_mid[om + pm[i]] + sigma_pop * s_mid[opop + ppop[i]] + punadj[i] * s_e[t];
      ldp Lrow = Lw + s * SP, Ccol = C + t;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int k = 0;
      for (; k + 3 < S; k += 4) {
        a0 += Lrow[k] * Ccol[k * TP];
        a1 += Lrow[k + 1] * Ccol[(k + 1) * TP];
        a2 += Lrow[k + 2] * Ccol[(k + 2) * TP];
        a3 += Lrow[k + 3] * Ccol[(k + 3) * TP];
      }
      for (; k < S; k++) a0 += Lrow[k] * Ccol[k * TP];
      eta += (a0 + a1) + (a2 + a3);
      const double r = y - N * d_inv_logit(eta);
      lp += y * d_log_inv_logit(eta) + (N - y) * d_log_inv_logit(-eta) - 0.5 * zn * zn; // stan:126-127,130-131
      r_lds[i] = r;
      pol.g(qi, sg * r - zn);
    }
  }
  __syncthreads();
  PROF_MARK(2);

  // ---------------- phase D: per-day gathers gC[:,t] = sum_i r_i Lw_ext[s_i,:]; level-1 segment sums
  {
    gcip day_ptr = as_g(M->sched), wave_task_ptr = day_ptr + M->c_wtp, task_day = day_ptr + M->c_td, ps = as_g(M->pi);
    const int ta = wave_task_ptr[w], tb = wave_task_ptr[w + 1];
    const int lk = lane < S ? lane : 0;
    for (int ti = ta; ti < tb; ti++) {
      const int t = __builtin_amdgcn_readfirstlane(task_day[ti]);
      const int a = __builtin_amdgcn_readfirstlane(day_ptr[t]), b = __builtin_amdgcn_readfirstlane(day_ptr[t + 1]);
      double acc = 0.0;
      for (int i = a; i < b; i++) {
        const int s = __builtin_amdgcn_readfirstlane(ps[i]);
        acc += r_lds[i] * Lw[s * SP + lk];
      }
      if (lane < S) C[lane * TP + t] = acc;
    }
  }
  {
    gcip sub_ptr = as_g(M->sched) + M->c_subptr, sub_idx = as_g(M->sched) + M->c_subidx;
    gcdp punadj = as_g(M->pd) + 2 * M->Npad;
    const int nsub = M->nsub, wb = M->sub_weighted_begin;
    for (int sub = tid; sub < nsub; sub += PT_THREADS) {
      const int a = sub_ptr[sub], b = sub_ptr[sub + 1];
      double sum = 0.0;
      if (sub >= wb) for (int j = a; j < b; j++) { const int i = sub_idx[j]; sum += r_lds[i] * punadj[i]; }
      else for (int j = a; j < b; j++) sum += r_lds[sub_idx[j]];
      Y[sub] = sum;
    }
  }
  __syncthreads();
  PROF_MARK(3);

  // ---------------- phase E: local prefix sums of gC; level-2 segment sums
  double pre[PT_CH];
  {
    const int t0 = w * PT_CH;
    double run = 0.0;
#pragma unroll
    for (int j = 0; j < PT_CH; j++) {
      const int t = t0 + j;
      if (lane < S && t < T - 1) run += C[lane * TP + t];
      pre[j] = run;
    }
    if (lane < S) X[w * SE + lane] = run;  // r_lds is dead from here on
  }
  {
    gcip seg_ptr = as_g(M->sched) + M->c_segptr, seg_kind = as_g(M->sched) + M->c_segkind, seg_index = as_g(M->sched) + M->c_segidx;
    gcdp seg_scale = as_g(M->seg_scale);
    const int nseg = M->nseg;
    for (int seg = tid; seg < nseg; seg += PT_THREADS) {
      const int a = seg_ptr[seg], b = seg_ptr[seg + 1];
      double sum = 0.0;
      for (int j = a; j < b; j++) sum += Y[j];
      const int kind = seg_kind[seg], index = seg_index[seg];
      if (kind == 0) pol.g(index, seg_scale[seg] * sum - s_mid[index - o_c]);
      else if (kind == 1) s_gs[index] = sum;
      else s_ge[index] = sum;
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
    for (int h = 0; h < PT_CH; h += PT_QB) {
      typename Pol::GT gt[PT_QB];
#pragma unroll
      for (int j = 0; j < PT_QB; j++) if (t0 + h + j < T) pol.g_load_q(o_Z + lane + S * (t0 + h + j), gt[j]);
#pragma unroll
      for (int j = 0; j < PT_QB; j++) {
        const int t = t0 + h + j;
        if (t < T) pol.g_fin(o_Z + lane + S * t, (t < T - 1 ? pre[h + j] + carry : 0.0) - gt[j].q, gt[j]);
      }
    }
  }
  {
    // dbT[s] = dpolling_bias[s] = residuals of state s + w_s * national residuals; C region is free now
    gcdp LT = as_g(M->mat) + M->m_LT, LB = as_g(M->mat) + M->m_LB, wv = as_g(M->mat) + M->m_w;
    double pT = 0.0, pB = 0.0;
    const double gnat = s_gs[S];
#pragma unroll
    for (int j = 0; j < 4; j++) {
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
