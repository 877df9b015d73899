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
// updates into the same pass (see LeapPolicy in potus_hmc.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_THREADS 1024
#define PT_NW 16          // waves per workgroup
#define PT_CH 16          // days per wave in the S x T block layout  (T <= PT_NW*PT_CH)
#define PT_SUBLEN 16      // entries per level-1 segment-sum task

struct DevModel {
  int S, T, P, M, Pop, Ns, Nn, Npoll, D, full;
  int SE, SP, TP;           // SE = S+1 (rows of Lw_ext); SP, TP odd LDS row strides
  int o_zT, o_Z, o_c, o_m, o_pop, o_mue, o_rho, o_ze, o_nn, o_ns, o_zb;
  int nmid;                 // o_nn - o_c : parameters between the S x T block and the noise blocks
  double sigma_c, sigma_m, sigma_pop, sigma_e;
  const double *Lw_ext;     // [SE][SP]  rows 0..S-1: L_W (zeros above diagonal); row S: v = L_W^T w
  const double *LT_t, *LB_t; // [k][s]   forward mat-vec, lanes = s
  const double *LT, *LB;     // [s][k]   transposed mat-vec, lanes = k
  const double *prior, *w;
  // polls, state and national merged, sorted by day
  const int *ps, *pt, *pp, *pm, *ppop, *pqidx;
  const double *py, *pn, *punadj, *psig;
  const int *day_ptr;       // [T+1] poll range of each day
  const int *wave_task_ptr; // [PT_NW+1]
  const int *task_day;      // [T] days, grouped by the wave that gathers them
  int nsub, nseg, sub_weighted_begin;
  const int *sub_ptr, *sub_idx, *seg_ptr, *seg_kind, *seg_index;
  const double *seg_scale;
  // LDS layout, offsets in doubles
  int l_C, l_Lw, l_X, l_Y, l_zT, l_zb, l_mid, l_bT, l_pb, l_e, l_gs, l_ge, l_scal, l_red;
  int lds_doubles;
};

// scalar slots in LDS (l_scal)
enum { SC_MUE = 0, SC_RHO, SC_SRHO, SC_XMUE, SC_XRHO, SC_N };
#define PT_NRED 8 // reduction slots

__device__ __forceinline__ double d_log_inv_logit(double x) { return x > 0 ? -log1p(exp(-x)) : x - log1p(exp(x)); }
__device__ __forceinline__ double d_inv_logit(double x) {
  if (x >= 0) return 1.0 / (1.0 + exp(-x));
  double e = exp(x);
  return e / (1.0 + e);
}

// Sum N values over the workgroup; every thread returns with the totals.  Fixed order.
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double *red, int tid) {
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
    for (int i = 0; i < PT_NW; i++) s += red[i * N + k];
    v[k] = s;
  }
  __syncthreads();
}

// Plain policy: read q, write grad (parity hook, Hamiltonian init).
struct PlainPolicy {
  const double *q_;
  double *g_;
  static constexpr int NEXTRA = 0;
  double extra[1];  // unused
  __device__ __forceinline__ double q(int i) { return q_[i]; }
  __device__ __forceinline__ double q_again(int i) { return q_[i]; } // position as returned by q(i) earlier in the pass
  __device__ __forceinline__ void g(int i, double v) { g_[i] = v; }
};

// Stage the (S+1) x S walk factor in LDS once per kernel; it stays resident across passes.
__device__ __forceinline__ void model_setup_lds(const DevModel &M, double *lds) {
  double *Lw = lds + M.l_Lw;
  for (int i = threadIdx.x; i < M.SE * M.SP; i += PT_THREADS) Lw[i] = M.Lw_ext[i];
  __syncthreads();
}

// One full pass.  Returns lp (log_prob<propto,jacobian>) in every thread; pol.extra[] are
// block-summed alongside.  Ends with a barrier, so LDS may be reused immediately.
template <class Pol>
__device__ __noinline__ double model_pass(const DevModel &M, double *lds, Pol &pol) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = M.S, T = M.T, SE = M.SE, SP = M.SP, TP = M.TP;
  double *C = lds + M.l_C, *Lw = lds + M.l_Lw, *X = lds + M.l_X, *Y = lds + M.l_Y;
  double *s_zT = lds + M.l_zT, *s_zb = lds + M.l_zb, *s_mid = lds + M.l_mid, *s_bT = lds + M.l_bT;
  double *s_pb = lds + M.l_pb, *s_e = lds + M.l_e, *s_gs = lds + M.l_gs, *s_ge = lds + M.l_ge;
  double *s_scal = lds + M.l_scal, *red = lds + M.l_red;
  double lp = 0.0;

  // ---------------- phase A: small parameters, then the S x T block with local suffix sums
  {
    // raw_mu_b_T / raw_polling_bias: wave w owns k = w, w+16, ... and starts both mat-vecs
    double vT = 0.0, vB = 0.0;
    const int kk = w + PT_NW * lane;
    if (lane < 4 && kk < S) {
      vT = pol.q(M.o_zT + kk);
      vB = pol.q(M.o_zb + kk);
      s_zT[kk] = vT;
      s_zb[kk] = vB;
      lp -= 0.5 * (vT * vT + vB * vB);          // stan:117,128
    }
    double pT = 0.0, pB = 0.0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = w + PT_NW * j;
      const double a = __shfl(vT, j, 64), b = __shfl(vB, j, 64);
      if (k < S && lane < S) {
        pT += M.LT_t[k * S + lane] * a;         // stan:85
        pB += M.LB_t[k * S + lane] * b;         // stan:77
      }
    }
    if (lane < S) { X[w * SE + lane] = pT; X[(PT_NW + w) * SE + lane] = pB; }
  }
  for (int i = tid; i < M.nmid; i += PT_THREADS) {
    const int idx = M.o_c + i;
    const double v = pol.q(idx);
    s_mid[i] = v;
    if (!(M.full && (idx == M.o_mue || idx == M.o_rho))) lp -= 0.5 * v * v; // stan:120-122,125
  }
  double cs[PT_CH];
  {
    const int t0 = w * PT_CH;
    double run = 0.0;
#pragma unroll
    for (int j = PT_CH - 1; j >= 0; j--) {
      const int t = t0 + j;
      double z = 0.0;
      if (lane < S && t < T) {
        z = pol.q(M.o_Z + lane + S * t);
        lp -= 0.5 * z * z;                      // to_vector(raw_mu_b) ~ std_normal(), stan:119
      }
      run += (t < T - 1) ? z : 0.0;             // column T is not part of the walk (stan:86)
      cs[j] = run;
    }
    if (lane < S) Y[w * SE + lane] = run;
  }
  __syncthreads();

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
      for (int w2 = 0; w2 < PT_NW; w2++) { bT += X[w2 * SE + lane]; pb += X[(PT_NW + w2) * SE + lane]; }
      bT += M.prior[lane];
      s_bT[lane] = bT;
      s_pb[lane] = pb;
    }
    const double ww = lane < S ? M.w[lane] : 0.0;
    double nb = ww * bT, np = ww * pb;          // stan:79 and the national average of mu_b[:,T]
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { nb += __shfl_down(nb, off, 64); np += __shfl_down(np, off, 64); }
    if (lane == 0) { s_bT[S] = nb; s_pb[S] = np; }
  }
  if (M.full && w == PT_NW - 2) {
    // e_bias (stan:91-93) as an affine scan over days: d[t] = e[t]-mu_e, d[t] = rho d[t-1] + sigma_rho z[t]
    const double *ze = s_mid + (M.o_ze - M.o_c);
    const double xm = s_mid[M.o_mue - M.o_c], xr = s_mid[M.o_rho - M.o_c];
    const double mue = 0.02 * xm, rho = d_inv_logit(xr);
    const double srho = sqrt(1.0 - rho * rho) * M.sigma_e;
    const int per = (T + 63) / 64, ta = lane * per, tb = min(T, ta + per);
    double A = 1.0, B = 0.0;
    for (int t = ta; t < tb; t++) {
      if (t == 0) { A = 0.0; B = ze[0] * M.sigma_e - mue; }
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
      d = (t == 0) ? ze[0] * M.sigma_e - mue : rho * d + srho * ze[t];
      s_e[t] = d + mue;
    }
    if (lane == 0) {
      s_scal[SC_MUE] = mue; s_scal[SC_RHO] = rho; s_scal[SC_SRHO] = srho; s_scal[SC_XMUE] = xm; s_scal[SC_XRHO] = xr;
      // Jacobians (stan:62-63) and the two informative priors (stan:123-124)
      lp += log(0.02) - 0.5 * xm * xm + log(rho) + log1p(-rho) - 0.5 * ((rho - 0.7) / 0.1) * ((rho - 0.7) / 0.1);
    }
  }
  __syncthreads();

  // ---------------- phase C: one thread per poll (stan:95-112, 130-131)
  double *r_lds = X;
  for (int i = tid; i < M.Npoll; i += PT_THREADS) {
    const int s = M.ps[i], t = M.pt[i];
    double eta = s_bT[s] + s_pb[s];
    const double *Lrow = Lw + s * SP, *Ccol = C + t;
    double a0 = 0.0, a1 = 0.0;
    int k = 0;
    for (; k + 1 < S; k += 2) { a0 += Lrow[k] * Ccol[k * TP]; a1 += Lrow[k + 1] * Ccol[(k + 1) * TP]; }
    if (k < S) a0 += Lrow[k] * Ccol[k * TP];
    eta += a0 + a1;
    eta += M.sigma_c * s_mid[M.pp[i]];
    if (M.full) {
      eta += M.sigma_m * s_mid[M.o_m - M.o_c + M.pm[i]] + M.sigma_pop * s_mid[M.o_pop - M.o_c + M.ppop[i]]
             + M.punadj[i] * s_e[t];
    }
    const int qi = M.pqidx[i];
    const double zn = pol.q(qi), sg = M.psig[i];
    eta += sg * zn;
    const double y = M.py[i], N = M.pn[i];
    const double r = y - N * d_inv_logit(eta);
    lp += y * d_log_inv_logit(eta) + (N - y) * d_log_inv_logit(-eta) - 0.5 * zn * zn; // stan:126-127,130-131
    r_lds[i] = r;
    pol.g(qi, sg * r - zn);
  }
  __syncthreads();

  // ---------------- phase D: per-day gathers gC[:,t] = sum_i r_i Lw_ext[s_i,:]; level-1 segment sums
  {
    const int ta = M.wave_task_ptr[w], tb = M.wave_task_ptr[w + 1];
    for (int ti = ta; ti < tb; ti++) {
      const int t = M.task_day[ti];
      const int a = M.day_ptr[t], b = M.day_ptr[t + 1];
      double acc = 0.0;
      for (int i = a; i < b; i++) {
        const int s = M.ps[i];
        if (lane < S) acc += r_lds[i] * Lw[s * SP + lane];
      }
      if (lane < S) C[lane * TP + t] = acc;
    }
  }
  for (int sub = tid; sub < M.nsub; sub += PT_THREADS) {
    const int a = M.sub_ptr[sub], b = M.sub_ptr[sub + 1];
    double sum = 0.0;
    if (sub >= M.sub_weighted_begin) for (int j = a; j < b; j++) { const int i = M.sub_idx[j]; sum += r_lds[i] * M.punadj[i]; }
    else for (int j = a; j < b; j++) sum += r_lds[M.sub_idx[j]];
    Y[sub] = sum;
  }
  __syncthreads();

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
  for (int seg = tid; seg < M.nseg; seg += PT_THREADS) {
    const int a = M.seg_ptr[seg], b = M.seg_ptr[seg + 1];
    double sum = 0.0;
    for (int j = a; j < b; j++) sum += Y[j];
    const int kind = M.seg_kind[seg], index = M.seg_index[seg];
    if (kind == 0) pol.g(index, M.seg_scale[seg] * sum - s_mid[index - M.o_c]);
    else if (kind == 1) s_gs[index] = sum;
    else s_ge[index] = sum;
  }
  __syncthreads();

  // ---------------- phase F: dZ; transposed mat-vecs; AR(1) adjoint
  if (lane < S) {
    double carry = 0.0;
    for (int w2 = 0; w2 < w; w2++) carry += X[w2 * SE + lane];
    const int t0 = w * PT_CH;
#pragma unroll
    for (int j = 0; j < PT_CH; j++) {
      const int t = t0 + j;
      if (t < T) {
        const int idx = M.o_Z + lane + S * t;
        pol.g(idx, (t < T - 1 ? pre[j] + carry : 0.0) - pol.q_again(idx));
      }
    }
  }
  {
    // dbT[s] = dpolling_bias[s] = residuals of state s + w_s * national residuals; C region is free now
    double pT = 0.0, pB = 0.0;
    const double gnat = s_gs[S];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int s = w + PT_NW * j;
      if (s < S && lane < S) {
        const double G = s_gs[s] + M.w[s] * gnat;
        pT += M.LT[s * S + lane] * G;
        pB += M.LB[s * S + lane] * G;
      }
    }
    if (lane < S) { C[w * SE + lane] = pT; C[(PT_NW + w) * SE + lane] = pB; }
  }
  if (M.full && w == PT_NW - 2) {
    // adjoint recursion a[t] = ge[t] + rho a[t+1]
    const double *ze = s_mid + (M.o_ze - M.o_c);
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
        pol.g(M.o_ze + t, a * srho - ze[t]);
      } else {
        pol.g(M.o_ze, a * M.sigma_e - ze[0]);
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      S1 += __shfl_down(S1, off, 64); S2 += __shfl_down(S2, off, 64); S3 += __shfl_down(S3, off, 64);
    }
    if (lane == 0) {
      const double xm = s_scal[SC_XMUE];
      const double adj_rho = S2 + S3 * M.sigma_e * (-rho / sqrt(1.0 - rho * rho));
      pol.g(M.o_mue, 0.02 * (1.0 - rho) * S1 - xm);
      pol.g(M.o_rho, (adj_rho - (rho - 0.7) / 0.01) * rho * (1.0 - rho) + (1.0 - 2.0 * rho));
    }
  }
  __syncthreads();

  // ---------------- phase G: finish the two transposed mat-vecs; reduce lp
  if (tid < 128) {
    const int which = tid >> 6;
    if (lane < S) {
      double sum = 0.0;
      for (int w2 = 0; w2 < PT_NW; w2++) sum += C[(which * PT_NW + w2) * SE + lane];
      if (which == 0) pol.g(M.o_zT + lane, sum - s_zT[lane]);
      else pol.g(M.o_zb + lane, sum - s_zb[lane]);
    }
  }
  double v[1 + Pol::NEXTRA];
  v[0] = lp;
#pragma unroll
  for (int k = 0; k < Pol::NEXTRA; k++) v[1 + k] = pol.extra[k];
  block_sum(v, red, tid);
#pragma unroll
  for (int k = 0; k < Pol::NEXTRA; k++) pol.extra[k] = v[1 + k];
  return v[0];
}
