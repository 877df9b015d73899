// potus_cluster.hpp -- one chain on a CLUSTER of K workgroups (K compute units) of a gfx950 XCD.
//
// potus_model.hpp runs one chain per workgroup: with the reference's 4-8 chains that leaves 248 of
// the 256 CUs idle and every leapfrog pays ~35 dependent L2 round trips per thread.  Here the K
// members of a cluster split the S x T block by DAYS (contiguous ranges, balanced by polls), each
// member owning
//     its days of raw_mu_b | the measurement-noise parameters of the polls on those days |
//     its days of raw_e_bias | a 1/K share of the small vectors (raw_mu_b_T, raw_polling_bias,
//     raw_mu_c, raw_mu_m, raw_mu_pop, mu_e_bias, rho_e_bias)
// stored CONTIGUOUSLY (the chain's vectors are kept in this internal order; ClModel::perm maps
// back to Stan's order when draws are saved).  Everything elementwise (leapfrog updates, U-turn
// dot products, Welford windows) is local to the owner; the members meet in three exchanges per
// gradient (stan:56-131 restated as in potus_model.hpp):
//   X1  per-state suffix totals of each member's days      -> C[:,t] needs the days after its own
//   X2  per-state prefix totals of the adjoint, the AR(1) adjoint composite of its days, partial
//       sums of everything indexed by pollster/mode/population and of the two transposed
//       51x51 mat-vecs                                       -> owners finish their gradients
//   X3  log density and kinetic energy partials             -> every member takes the same decision
// An exchange is: payload written with sc1 (write-through) stores, one agent-scope fetch_add on the
// cluster's counter, one lane polling it, payload of the others read back with sc1 loads; measured
// 1.4-1.5 us for K <= 16 on one XCD (scripts/micro/cluster_exchange.hip).  All members sum the
// partials in the same fixed order, so they hold bit-identical scalars and run the NUTS control
// flow redundantly without ever diverging; results are reproducible run to run for a given K.
//
// Launch: grid = chains * K, block b -> chain b % chains, member b / chains, so that with 8 chains
// the members of a chain land on one XCD (blocks are dealt round-robin to the 8 XCDs) and share
// its L2.  All blocks must be co-resident (grid <= number of CUs; checked on the host).
#pragma once
#include "potus_nuts.hpp"

#define CL_DW 8                          // days per wave
#define CL_MAXDAYS (PT_NW * CL_DW)       // days per member
#define CL_LPP 4                         // lanes cooperating on one poll's 51-term dot
#define CL_PPR (PT_THREADS / CL_LPP)     // polls per round of the poll phase
#define CL_MAXK 32
#define CL_AUX_SC1 16                    // cache-policy bit of the buffer intrinsics: sc1 (agent scope)
#define CL_SPIN_LIMIT 8000000u

// fields of one member's part descriptor (ints)
enum { CP_D0 = 0, CP_ND, CP_P0, CP_NP, CP_E0, CP_NE, CP_R0, CP_NR, CP_NSUB, CP_NSEG, CP_WB,
       CP_O_WD, CP_O_MASK, CP_O_SUB, CP_O_SEGPTR, CP_O_SEGKIND, CP_O_SEGIDX, CP_O_WT, CP_N = 20 };
// payload layout of exchange X2 (doubles); X1 and the scalar all-reduces use the first words
enum { XP_PRE = 0, XP_AR = 64, XP_S = 66, XP_P = 72 };

struct ClModel {
  int K, XW, NR, NREP, NDP, npmax, nsubmax, pad0;
  const int *part;          // [K][CP_N]
  const int *sched;         // per member: wd_t|wd_a|wd_b [PT_THREADS] each, daymask [PT_NW], sub16, seg_ptr, seg_kind, seg_index
  const double *wt;         // weights of the weighted level-1 tasks
  const int *rep_pos;       // [NREP] internal index of the small parameters every member reads: zT | zb | c,m,pop,mue,rho,ze
  const double *rep_scale;  // [NR] scale of owned slot r (sigma_c ... ; 1 for zT, zb)
  const int *perm;          // [D] internal index -> Stan index
  int l_C, l_Lw, l_X, l_Y, l_r, l_rep, l_bT, l_pb, l_e, l_c1, l_c2, l_c3, l_gs, l_ge, l_P, l_scal, l_red, l_st, l_prof;
  int lds_doubles;
};
typedef const ClModel AS_C *CCp;
typedef const int AS_C *cip;

__device__ __forceinline__ double bld_s(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, CL_AUX_SC1));
}
__device__ __forceinline__ void bst_s(rsrc_t r, unsigned voff, unsigned soff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, CL_AUX_SC1);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <class P> __device__ __forceinline__ P launder_s(P p) { // keep address math inside the loop it belongs to
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)p);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)p >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return (P)(((unsigned long long)hi << 32) | lo);
}

#ifdef POTUS_PROF
#define PROFPTR prof
#define CPROFPTR(c) (c).prof
#else
#define PROFPTR nullptr
#define CPROFPTR(c) nullptr
#endif
// ---------------------------------------------------------------- exchanges
struct Xch {
  rsrc_t xb;               // the chain's exchange buffer [2][K][XW] doubles
  unsigned *cnt;           // the chain's arrival counter (zeroed by the host before every launch)
  unsigned epoch;          // exchanges completed so far in this launch (identical in every member)
  int K, m, XW;
};
// byte offset of member mm's payload: w = the exchange being assembled, r = the one just completed
__device__ __forceinline__ unsigned xch_wslot(const Xch &x, int mm) { return ((((x.epoch + 1u) & 1u) * (unsigned)x.K + (unsigned)mm) * (unsigned)x.XW) * 8u; }
__device__ __forceinline__ unsigned xch_rslot(const Xch &x, int mm) { return (((x.epoch & 1u) * (unsigned)x.K + (unsigned)mm) * (unsigned)x.XW) * 8u; }
// Wave 0 only, after a barrier that follows every payload store of the workgroup: publish and wait.
__device__ __forceinline__ void xch_signal_wait(const Xch &x) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) {
    __hip_atomic_fetch_add(x.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned want = (x.epoch + 1u) * (unsigned)x.K;
    unsigned spins = 0;
    while (__hip_atomic_load(x.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > CL_SPIN_LIMIT) __builtin_trap();   // a member is missing: fail loudly instead of hanging the GPU
    }
  }
}
// Cluster-wide barrier that also orders sc1 stores before it against sc1 loads after it.
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void cl_sync(Xch &x) {
  drain_vmem();
  __syncthreads();
  if (threadIdx.x < 64) xch_signal_wait(x);
  __syncthreads();
  x.epoch++;
}
// Sum N <= 8 values over every thread of every member; all threads of all members return the same bits.
template <int N>
__device__ __forceinline__ void cl_allreduce(double (&v)[N], ldp red, Xch &x, int tid, ldp prof = nullptr) {
  (void)prof;
  static_assert(N <= 8, "payload words");
  const int lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int k = 0; k < N; k++) {
    double t = v[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) t += __shfl_down(t, off, 64);
    if (lane == 0) red[w * N + k] = t;
  }
  PROF_MARK(20);
  drain_vmem();   // this wave's write-through stores are complete before the workgroup signals
  PROF_MARK(21);
  __syncthreads();
  PROF_MARK(22);
  if (w == 0) {
    double s = 0.0;
    const int k = lane < N ? lane : N - 1;
#pragma unroll
    for (int i = 0; i < PT_NW; i++) s += red[i * N + k];
    bst_s(x.xb, lane < N ? 8u * (unsigned)lane : PT_OOB, xch_wslot(x, x.m), s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PROF_MARK(23);
    xch_signal_wait(x);
    PROF_MARK(24);
  }
  __syncthreads();
  PROF_MARK(25);
  x.epoch++;
  // lane l fetches word (l & 7) of members (l >> 3) + 8u; fixed-shape tree over the members
  double s[CL_MAXK / 8];
  const unsigned base = xch_rslot(x, 0);
#pragma unroll
  for (int u = 0; u < CL_MAXK / 8; u++) {
    const int mm = (lane >> 3) + 8 * u;
    s[u] = bld_s(x.xb, (mm < x.K && (lane & 7) < N) ? (unsigned)mm * (unsigned)x.XW * 8u + 8u * (unsigned)(lane & 7) : PT_OOB, base);
  }
  double t = ((s[0] + s[1]) + s[2]) + s[3];
  t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
#pragma unroll
  for (int k = 0; k < N; k++) v[k] = readlane_d(t, k);
  PROF_MARK(26);
}

// ---------------------------------------------------------------- policies (internal element order)
struct ClPlainPolicy {
  rsrc_t rq, rg;
  unsigned sq, sg;
  static constexpr int NEXTRA = 0;
  double extra[1];
  struct QT { double q; };
  struct GT { };
  __device__ __forceinline__ void q_load(unsigned vo, QT &t) { t.q = bld(rq, vo, sq); }
  __device__ __forceinline__ void qs_load(unsigned vo, QT &t) { t.q = bld_s(rq, vo, sq); }   // written by another member
  __device__ __forceinline__ double q_fin(QT &t) { return t.q; }
  __device__ __forceinline__ void g_load(unsigned, GT &) {}
  __device__ __forceinline__ void g_fin(unsigned vo, double v, double, const GT &) { bst(rg, vo, sg, v); }
  __device__ __forceinline__ void gs_fin(unsigned vo, double v, double, const GT &) { bst(rg, vo, sg, v); }
};
// Pre-kicked leapfrog (LeapPolicy of potus_nuts.hpp); positions that other members read next pass
// (the small vectors, raw_e_bias) are stored write-through.
struct ClLeapPolicy {
  rsrc_t r;
  unsigned sQc, sQn, sPH, sM, sL;
  double he, e;
  static constexpr int NEXTRA = 1;
  double extra[1];
  struct QT { double q; };
  struct GT { double p, m; };
  __device__ __forceinline__ void q_load(unsigned vo, QT &t) { t.q = bld(r, vo, sQc); }
  __device__ __forceinline__ void qs_load(unsigned vo, QT &t) { t.q = bld_s(r, vo, sQc); }
  __device__ __forceinline__ double q_fin(QT &t) { return t.q; }
  __device__ __forceinline__ void g_load(unsigned vo, GT &t) { t.p = bld(r, vo, sPH); t.m = bld(r, vo, sM); }
  __device__ __forceinline__ void g_fin(unsigned vo, double v, double q, const GT &t) {
    const double pf = t.p + he * v;
    bst(r, vo, sL, pf);
    const double ph = pf + he * v;
    bst(r, vo, sPH, ph);
    bst(r, vo, sQn, q + e * t.m * ph);
    extra[0] += t.m * pf * pf;
  }
  __device__ __forceinline__ void gs_fin(unsigned vo, double v, double q, const GT &t) {
    const double pf = t.p + he * v;
    bst(r, vo, sL, pf);
    const double ph = pf + he * v;
    bst(r, vo, sPH, ph);
    bst_s(r, vo, sQn, q + e * t.m * ph);
    extra[0] += t.m * pf * pf;
  }
};

struct ClStatic {           // per-thread registers that never change during a kernel
  int d_t, d_a, d_b;        // lane j of wave w: j-th day gathered by the wave (local day, local poll range)
  unsigned rep_vo[2];       // byte offsets of the small parameters this thread fetches for the workgroup
};

__device__ __forceinline__ ClStatic cl_load_static(CCp CL, cip part) {
  ClStatic c;
  gcip sc = as_g(CL->sched) + part[CP_O_WD];
  const int tid = threadIdx.x;
  c.d_t = sc[tid]; c.d_a = sc[PT_THREADS + tid]; c.d_b = sc[2 * PT_THREADS + tid];
  gcip rp = as_g(CL->rep_pos);
  const int NREP = CL->NREP;
#pragma unroll
  for (int u = 0; u < 2; u++) { const int j = tid + u * PT_THREADS; c.rep_vo[u] = j < NREP ? 8u * (unsigned)rp[j] : PT_OOB; }
  return c;
}
// Stage the walk factor and the (pseudo-)states of the member's polls in LDS, once per kernel.
__device__ __forceinline__ ClStatic cl_setup_lds(CMp M, CCp CL, cip part, ldp lds) {
  ldp Lw = lds + CL->l_Lw;
  gcdp src = as_g(M->mat);
  for (int i = threadIdx.x; i < M->SE * M->SP; i += PT_THREADS) Lw[i] = src[i];
  unsigned char AS_L *st = (unsigned char AS_L *)(lds + CL->l_st);
  gcip ps = as_g(M->pi) + part[CP_P0];
  const int np = part[CP_NP];
  for (int i = threadIdx.x; i < np + 8; i += PT_THREADS) st[i] = i < np ? (unsigned char)ps[i] : (unsigned char)0;
  const ClStatic c = cl_load_static(CL, part);
#ifdef POTUS_PROF
  for (int i = threadIdx.x; i < PT_NPROF; i += PT_THREADS) (lds + CL->l_prof)[i] = 0.0;
#endif
  __syncthreads();
  return c;
}

// ---------------------------------------------------------------- one pass of the member's share
// Returns the chain's lp in every thread of every member; pol.extra[] are summed alongside.
template <class Pol>
__device__ __forceinline__ double cl_pass(CMp M_in, CCp CL_in, cip part_in, ldp lds, const ClStatic &cst, Xch &x, Pol &pol_io) {
  Pol pol = pol_io;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  CMp M = launder_s(M_in);
  CCp CL = launder_s(CL_in);
  cip part = launder_s(part_in);
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = M->S, T = M->T, SE = M->SE, SP = M->SP, full = M->full, o_c = M->o_c;
  const int NDP = CL->NDP, NR = CL->NR, NREP = CL->NREP;
  const int d0 = part[CP_D0], nd = part[CP_ND], p0 = part[CP_P0], np = part[CP_NP], e0 = part[CP_E0], r0 = part[CP_R0], nr = part[CP_NR];
  const int K = x.K, m = x.m;
  ldp C = lds + CL->l_C, Lw = lds + CL->l_Lw, X = lds + CL->l_X, Y = lds + CL->l_Y, r_lds = lds + CL->l_r;
  ldp s_rep = lds + CL->l_rep;
  ldp s_zT = s_rep, s_zb = s_rep + S, s_mid = s_rep + 2 * S;
  ldp s_bT = lds + CL->l_bT, s_pb = lds + CL->l_pb, s_e = lds + CL->l_e, s_c1 = lds + CL->l_c1, s_c2 = lds + CL->l_c2, s_c3 = lds + CL->l_c3;
  ldp s_gs = lds + CL->l_gs, s_ge = lds + CL->l_ge, s_P = lds + CL->l_P, s_scal = lds + CL->l_scal, red = lds + CL->l_red;
  const int e_noise = e0 + S * nd, e_ze = e_noise + np, e_rep = e_ze + (full ? nd : 0);
  double lp = 0.0;
#ifdef POTUS_PROF
  ldp prof = lds + CL->l_prof;
#endif
  PROF_START();

  // ---------------- phase A: the small vectors (all members read all of them), own days of the S x T block
  double cs[CL_DW], zq[CL_DW];
  {
    typename Pol::QT qr[2], qt[CL_DW];
    pol.qs_load(cst.rep_vo[0], qr[0]);
    pol.qs_load(cst.rep_vo[1], qr[1]);
#pragma unroll
    for (int j = 0; j < CL_DW; j++) {
      const int tl = w * CL_DW + j;
      pol.q_load((lane < S && tl < nd) ? 8u * (unsigned)(e0 + lane + S * tl) : PT_OOB, qt[j]);
    }
    for (int i = tid; i < NR + 8; i += PT_THREADS) s_P[i] = 0.0;   // accumulators that this member's polls may not cover
    if (tid < SE) s_gs[tid] = 0.0;
    if (tid >= 64 && tid < 64 + CL_MAXDAYS) s_ge[tid - 64] = 0.0;
#pragma unroll
    for (int u = 0; u < 2; u++) { const int j = tid + u * PT_THREADS; s_rep[j < NREP ? j : NREP] = pol.q_fin(qr[u]); }
    double run = 0.0;
#pragma unroll
    for (int j = CL_DW - 1; j >= 0; j--) {
      const int t = d0 + w * CL_DW + j;
      const double z = pol.q_fin(qt[j]);          // 0 for masked-off elements
      zq[j] = z;
      lp -= 0.5 * z * z;                          // stan:119
      run += (t < T - 1) ? z : 0.0;               // column T is not part of the walk (stan:86)
      cs[j] = run;
    }
    if (lane < S) Y[w * SE + lane] = run;
  }
  __syncthreads();
  PROF_MARK(0);

  // ---------------- phase B: X1 (suffix totals); meanwhile mu_b_T / polling-bias mat-vecs and the AR(1) bias
  if (w == 0) {
    double tot = 0.0;
    if (lane < S) {
#pragma unroll
      for (int w2 = 0; w2 < PT_NW; w2++) tot += Y[w2 * SE + lane];
    }
    bst_s(x.xb, lane < S ? 8u * (unsigned)lane : PT_OOB, xch_wslot(x, m), tot);
    xch_signal_wait(x);
  } else if (w == 1) {
    if (full) {
      // e_bias (stan:91-93): d[t] = e[t]-mu_e = rho d[t-1] + sigma_rho z[t], plus the three tangent
      // recurrences that turn the adjoint sums of mu_e_bias / rho_e_bias into per-day dot products:
      //   c1[t] = rho c1[t-1] + 1, c2[t] = rho c2[t-1] + d[t-1], c3[t] = rho c3[t-1] + z[t]   (c.[0] = 0)
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
      double d_in = __shfl_up(B, 1, 64);
      if (lane == 0) d_in = 0.0;
      double A2 = 1.0, B1 = 0.0, B2 = 0.0, B3 = 0.0, d = d_in;
      for (int t = ta; t < tb; t++) {
        const double dprev = d;
        d = (t == 0) ? ze[0] * sigma_e - mue : rho * d + srho * ze[t];
        s_e[t] = d + mue;
        if (t == 0) { A2 = 0.0; B1 = 0.0; B2 = 0.0; B3 = 0.0; }
        else { A2 = rho * A2; B1 = rho * B1 + 1.0; B2 = rho * B2 + dprev; B3 = rho * B3 + ze[t]; }
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double a2 = __shfl_up(A2, off, 64), b1 = __shfl_up(B1, off, 64), b2 = __shfl_up(B2, off, 64), b3 = __shfl_up(B3, off, 64);
        if (lane >= off) { B1 = A2 * b1 + B1; B2 = A2 * b2 + B2; B3 = A2 * b3 + B3; A2 = A2 * a2; }
      }
      double c1 = __shfl_up(B1, 1, 64), c2 = __shfl_up(B2, 1, 64), c3 = __shfl_up(B3, 1, 64);
      if (lane == 0) { c1 = 0.0; c2 = 0.0; c3 = 0.0; }
      d = d_in;
      for (int t = ta; t < tb; t++) {
        const double dprev = d;
        d = (t == 0) ? ze[0] * sigma_e - mue : rho * d + srho * ze[t];
        if (t == 0) { c1 = 0.0; c2 = 0.0; c3 = 0.0; }
        else { c1 = rho * c1 + 1.0; c2 = rho * c2 + dprev; c3 = rho * c3 + ze[t]; }
        s_c1[t] = c1; s_c2[t] = c2; s_c3[t] = c3;
      }
      if (lane == 0) { s_scal[SC_MUE] = mue; s_scal[SC_RHO] = rho; s_scal[SC_SRHO] = srho; s_scal[SC_XMUE] = xm; s_scal[SC_XRHO] = xr; }
    }
  } else {
    // partial products of the two 51 x 51 factors: wave w-2 takes columns k = w-2, w+4, ...
    const int wj = w - 2;
    const rsrc_t rm = make_rsrc(M->mat, 8u * (unsigned)(M->m_w + S));
    const unsigned sLT = 8u * (unsigned)M->m_LTt, sLB = 8u * (unsigned)M->m_LBt;
    constexpr int NJ = 11;                        // 6 waves x 11 columns >= 63
    double lt[NJ], lb[NJ], zt[NJ], zb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int k = wj + 6 * j;
      const unsigned vo = (k < S && lane < S) ? 8u * (unsigned)(k * S + lane) : PT_OOB;
      lt[j] = bld(rm, vo, sLT);                   // L_T[lane][k], stan:85
      lb[j] = bld(rm, vo, sLB);                   // L_B[lane][k], stan:77
      zt[j] = s_zT[k < S ? k : 0];
      zb[j] = s_zb[k < S ? k : 0];
    }
    ISSUE_FENCE();
    double pT = 0.0, pB = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; j++) { pT += lt[j] * zt[j]; pB += lb[j] * zb[j]; }
    const int sx = lane < S ? lane : S;
    X[wj * SE + sx] = pT;
    X[(6 + wj) * SE + sx] = pB;
  }
  __syncthreads();
  PROF_MARK(1);
  x.epoch++;
  {
    // C[k][t] for the member's days: local suffix + later waves + later members
    double carry = 0.0;
    for (int mm0 = m + 1; mm0 < K; mm0 += 8) {
      double t8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int mm = mm0 + u;
        t8[u] = bld_s(x.xb, (mm < K && lane < S) ? 8u * (unsigned)lane : PT_OOB, xch_rslot(x, mm < K ? mm : 0));
      }
#pragma unroll
      for (int u = 0; u < 8; u++) carry += t8[u];
    }
    if (lane < S) {
      double cy[PT_NW];
#pragma unroll
      for (int w2 = 0; w2 < PT_NW; w2++) cy[w2] = Y[w2 * SE + lane];
      ISSUE_FENCE();
#pragma unroll
      for (int w2 = 0; w2 < PT_NW; w2++) carry += w2 > w ? cy[w2] : 0.0;
#pragma unroll
      for (int j = 0; j < CL_DW; j++) {
        const int tl = w * CL_DW + j;
        if (tl < nd) C[lane * NDP + tl] = cs[j] + carry;
      }
    }
  }
  if (w == PT_NW - 1) {
    double bT = 0.0, pb = 0.0;
    if (lane < S) {
#pragma unroll
      for (int w2 = 0; w2 < 6; w2++) { bT += X[w2 * SE + lane]; pb += X[(6 + w2) * SE + lane]; }
      bT += (as_g(M->mat) + M->m_prior)[lane];
      s_bT[lane] = bT;
      s_pb[lane] = pb;
    }
    const double ww = lane < S ? (as_g(M->mat) + M->m_w)[lane] : 0.0;
    double nb = ww * bT, npb = ww * pb;          // stan:79 and the national average of mu_b[:,T]
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { nb += __shfl_down(nb, off, 64); npb += __shfl_down(npb, off, 64); }
    if (lane == 0) { s_bT[S] = nb; s_pb[S] = npb; }
  }
  if (tid == 0) r_lds[np] = 0.0;
  __syncthreads();
  PROF_MARK(2);

  // ---------------- phase C: the member's polls, CL_LPP lanes per poll (stan:95-112, 130-131)
  {
    const unsigned Npad = M->Npad;
    const rsrc_t rpi = make_rsrc(M->pi, 6u * Npad * 4u), rpd = make_rsrc(M->pd, 4u * Npad * 8u);
    const int om = M->o_m - o_c, opop = M->o_pop - o_c;
    const double sigma_c = M->sigma_c, sigma_m = M->sigma_m, sigma_pop = M->sigma_pop;
    const int sub = tid & (CL_LPP - 1);
    for (int i0 = 0; i0 < np; i0 += CL_PPR) {
      const int il = i0 + (tid >> 2);
      const bool ok = il < np, lead = ok && sub == 0;
      const unsigned gi = (unsigned)(p0 + il);
      const unsigned vi = ok ? 4u * gi : PT_OOB, vd = lead ? 8u * gi : PT_OOB;   // masked polls read zeros: N = y = 0
      const int s = bld_i(rpi, vi, 0), t = bld_i(rpi, vi, 4u * Npad), ip = bld_i(rpi, vi, 8u * Npad);
      int im = 0, ipop = 0;
      double un = 0.0;
      if (full) { im = bld_i(rpi, vi, 12u * Npad); ipop = bld_i(rpi, vi, 16u * Npad); un = bld(rpd, vd, 16u * Npad); }
      const double y = bld(rpd, vd, 0), N = bld(rpd, vd, 8u * Npad), sg = bld(rpd, vd, 24u * Npad);
      const unsigned vq = lead ? 8u * (unsigned)(e_noise + il) : PT_OOB;
      typename Pol::QT qt;
      typename Pol::GT gt;
      pol.q_load(vq, qt);
      pol.g_load(vq, gt);
      const int tl = ok ? t - d0 : 0;
      ldp L0 = Lw + s * SP, C0 = C + tl;
      constexpr int NT = (64 + CL_LPP - 1) / CL_LPP;   // terms per lane, S <= 63
      double l[NT], c[NT];
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const int kk = min(sub + CL_LPP * j, S - 1);
        l[j] = L0[kk]; c[j] = C0[kk * NDP];
      }
      ISSUE_FENCE();
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int j = 0; j < NT; j += 2) {
        a0 += (sub + CL_LPP * j < S ? 1.0 : 0.0) * l[j] * c[j];
        if (j + 1 < NT) a1 += (sub + CL_LPP * (j + 1) < S ? 1.0 : 0.0) * l[j + 1] * c[j + 1];
      }
      double dot = a0 + a1;
      dot += __shfl_xor(dot, 1, 64);
      dot += __shfl_xor(dot, 2, 64);
      const double zn = pol.q_fin(qt);
      double eta = s_bT[s] + s_pb[s] + sg * zn + sigma_c * s_mid[ip] + dot;
      if (full) eta += sigma_m * s_mid[om + im] + sigma_pop * s_mid[opop + ipop] + un * s_e[ok ? t : 0];
      const double ex = exp(-fabs(eta)), l1 = log1p(ex), pr = (eta >= 0.0 ? 1.0 : ex) / (1.0 + ex);
      const double r = y - N * pr;                       // y = N = 0 on the helper lanes
      lp += y * (fmin(eta, 0.0) - l1) + (N - y) * (fmin(-eta, 0.0) - l1) - (lead ? 0.5 * zn * zn : 0.0); // stan:126-127,130-131
      r_lds[lead ? il : np + 1] = r;                     // slot np stays 0 (padding of the task lists), np+1 is a dump
      pol.g_fin(vq, sg * r - zn, zn, gt);
    }
  }
  __syncthreads();
  PROF_MARK(3);

  // ---------------- phase D: per-day gathers gC[:,t] = sum_i r_i Lw_ext[s_i,:]; level-1 segment sums
  const rsrc_t rsc = make_rsrc(CL->sched, 0x7ffffff0u);
  int sg_a = 0, sg_b = 0, sg_kind = 3, sg_index = 0;
  {
    const int nseg = part[CP_NSEG];
    const unsigned vs = tid < nseg ? 4u * (unsigned)tid : PT_OOB;
    sg_a = bld_i(rsc, vs, 4u * (unsigned)part[CP_O_SEGPTR]);
    sg_b = bld_i(rsc, vs, 4u * (unsigned)part[CP_O_SEGPTR] + 4u);
    sg_kind = tid < nseg ? bld_i(rsc, vs, 4u * (unsigned)part[CP_O_SEGKIND]) : 3;
    sg_index = bld_i(rsc, vs, 4u * (unsigned)part[CP_O_SEGIDX]);
  }
  {
    const unsigned char AS_L *st = (const unsigned char AS_L *)(lds + CL->l_st);
    const int lk = lane < S ? lane : 0;
    for (int dj = 0; dj < 64; dj++) {
      const int a = __builtin_amdgcn_readlane(cst.d_a, dj), b = __builtin_amdgcn_readlane(cst.d_b, dj);
      if (a >= b) break;                          // a wave's days are packed at the front
      const int tl = __builtin_amdgcn_readlane(cst.d_t, dj);
      double acc0 = 0.0, acc1 = 0.0;
      for (int i = a; i < b; i += 8) {            // eight polls per trip, reads beyond b hit the zero slot
        int s8[8];
        double r8[8], l8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int ii = i + u < b ? i + u : np; s8[u] = st[ii]; r8[u] = r_lds[ii]; }
        ISSUE_FENCE();
#pragma unroll
        for (int u = 0; u < 8; u++) l8[u] = Lw[s8[u] * SP + lk];
        ISSUE_FENCE();
#pragma unroll
        for (int u = 0; u < 8; u += 2) { acc0 += r8[u] * l8[u]; acc1 += r8[u + 1] * l8[u + 1]; }
      }
      if (lane < S) C[lane * NDP + tl] = acc0 + acc1;
    }
  }
  {
    const int nsub = part[CP_NSUB], wb = part[CP_WB];
    const unsigned o_sub = 4u * (unsigned)part[CP_O_SUB];
    const rsrc_t rwt = make_rsrc(CL->wt, 0x7ffffff0u);
    const unsigned o_wt = 8u * (unsigned)part[CP_O_WT];
    for (int sub0 = 0; sub0 < nsub; sub0 += PT_THREADS) {
      const int sub = sub0 + tid;
      const bool ok = sub < nsub, wtd = ok && sub >= wb;
      const unsigned vs = ok ? 64u * (unsigned)sub : PT_OOB, vw = wtd ? 128u * (unsigned)(sub - wb) : PT_OOB;
      u32x4 ix[4];
      double wt[PT_SUBLEN];
#pragma unroll
      for (int j = 0; j < 4; j++) ix[j] = bld_i4(rsc, vs + 16u * j, o_sub);
#pragma unroll
      for (int j = 0; j < PT_SUBLEN / 2; j++) bld_d2(rwt, vw + 16u * j, o_wt, wt[2 * j], wt[2 * j + 1]);
      double rr[PT_SUBLEN];
#pragma unroll
      for (int j = 0; j < PT_SUBLEN; j++) rr[j] = r_lds[ok ? ix[j >> 2][j & 3] : np];
      ISSUE_FENCE();
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < PT_SUBLEN; j++) sum += wtd ? rr[j] * wt[j] : rr[j];
      if (ok) Y[sub] = sum;
    }
  }
  __syncthreads();
  PROF_MARK(4);

  // ---------------- phase E: local prefix sums of gC; level-2 segment sums
  double pre[CL_DW];
  {
    const unsigned mask = (unsigned)(as_g(CL->sched) + part[CP_O_MASK])[w];   // days of this wave that have polls
    double run = 0.0;
#pragma unroll
    for (int j = 0; j < CL_DW; j++) {
      const int tl = w * CL_DW + j;
      if (((mask >> j) & 1u) && lane < S && d0 + tl < T - 1) run += C[lane * NDP + tl];
      pre[j] = run;
    }
    if (lane < S) X[w * SE + lane] = run;
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
    if (sg_kind == 0) s_P[sg_index] = sum;          // pollster / mode / population partial (slot index)
    else if (sg_kind == 1) s_gs[sg_index] = sum;    // residual sum of (pseudo-)state
    else if (sg_kind == 2) s_ge[sg_index] = sum;    // sum of unadjusted * residual over a local day
  }
  __syncthreads();
  PROF_MARK(5);

  // ---------------- phase E2: payload of X2
  double arA = 1.0, arB = 0.0;                      // wave 1 keeps its per-day adjoint composites for phase F
  if (w == 0) {
    double tot = 0.0;
    if (lane < S) {
#pragma unroll
      for (int w2 = 0; w2 < PT_NW; w2++) tot += X[w2 * SE + lane];
    }
    bst_s(x.xb, lane < S ? 8u * (unsigned)(XP_PRE + lane) : PT_OOB, xch_wslot(x, m), tot);
  } else if (w == 1) {
    if (full) {
      // adjoint of the AR(1) recursion over the member's days (one lane per day): a[t] = ge[t] + rho a[t+1]
      const double rho = s_scal[SC_RHO];
      const bool in = lane < nd;
      const double ge = in ? s_ge[lane] : 0.0;
      arA = in ? rho : 1.0; arB = ge;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double A2 = __shfl_down(arA, off, 64), B2 = __shfl_down(arB, off, 64);
        if (lane + off < 64) { arB = arB + arA * B2; arA = arA * A2; }
      }
      const int t = in ? d0 + lane : 0;
      double S1 = ge * s_c1[t], S2 = ge * s_c2[t], S3 = ge * s_c3[t];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) { S1 += __shfl_down(S1, off, 64); S2 += __shfl_down(S2, off, 64); S3 += __shfl_down(S3, off, 64); }
      const double v = lane == 0 ? arA : lane == 1 ? arB : lane == 2 ? S1 : lane == 3 ? S2 : S3;
      const double S1b = __shfl(S1, 0, 64), S2b = __shfl(S2, 0, 64), S3b = __shfl(S3, 0, 64);
      const double A0 = __shfl(arA, 0, 64), B0 = __shfl(arB, 0, 64);
      const double pv = lane == 0 ? A0 : lane == 1 ? B0 : lane == 2 ? S1b : lane == 3 ? S2b : S3b;
      (void)v;
      bst_s(x.xb, lane < 5 ? 8u * (unsigned)(XP_AR + lane) : PT_OOB, xch_wslot(x, m), pv);   // XP_S = XP_AR + 2
      drain_vmem();
    }
  } else {
    // dbT[s] = dpolling_bias[s] = residuals of state s + w_s * national residuals (this member's polls
    // only: the products are linear, the owners add the K partials); C region is free now
    const int wj = w - 2;
    gcdp LT = as_g(M->mat) + M->m_LT, LB = as_g(M->mat) + M->m_LB, wv = as_g(M->mat) + M->m_w;
    double pT = 0.0, pB = 0.0;
    const double gnat = s_gs[S];
    constexpr int NJ = 11;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int s = wj + 6 * j;
      if (s < S && lane < S) {
        const double G = s_gs[s] + wv[s] * gnat;
        pT += LT[s * S + lane] * G;
        pB += LB[s * S + lane] * G;
      }
    }
    if (lane < S) { C[wj * SE + lane] = pT; C[(6 + wj) * SE + lane] = pB; }
  }
  __syncthreads();
  PROF_MARK(6);
  if (w == 0) {
    for (int rr = lane; rr < NR; rr += 64) {
      double v;
      if (rr < 2 * S) {
        const int which = rr >= S, k = rr - which * S;
        v = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < 6; w2++) v += C[(which * 6 + w2) * SE + k];
      } else v = s_P[rr];
      bst_s(x.xb, 8u * (unsigned)(XP_P + rr), xch_wslot(x, m), v);
    }
    xch_signal_wait(x);
  }
  // loads that do not depend on the exchange are issued while wave 0 waits
  typename Pol::GT gz[CL_DW];
  unsigned voz[CL_DW];
#pragma unroll
  for (int j = 0; j < CL_DW; j++) {
    const int tl = w * CL_DW + j;
    voz[j] = (lane < S && tl < nd) ? 8u * (unsigned)(e0 + lane + S * tl) : PT_OOB;
    pol.g_load(voz[j], gz[j]);
  }
  const bool arl = full && w == 1 && lane < nd;                 // owner of raw_e_bias[d0 + lane]
  const int jr = tid - 128;
  const bool repl = jr >= 0 && jr < nr;                         // owner of small-vector slot r0 + jr
  const int rslot = repl ? r0 + jr : 0;
  const unsigned vo_x = arl ? 8u * (unsigned)(e_ze + lane) : repl ? 8u * (unsigned)(e_rep + jr) : PT_OOB;
  typename Pol::GT gx;
  pol.g_load(vo_x, gx);
  const rsrc_t rscale = make_rsrc(CL->rep_scale, 8u * (unsigned)NR);
  const double scale_r = bld(rscale, repl ? 8u * (unsigned)rslot : PT_OOB, 0);
  __syncthreads();
  PROF_MARK(7);
  x.epoch++;

  // ---------------- phase F: finish the gradients of everything this member owns
  {
    double carry = 0.0;
    for (int mm0 = 0; mm0 < m; mm0 += 8) {
      double t8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int mm = mm0 + u;
        t8[u] = bld_s(x.xb, (mm < m && lane < S) ? 8u * (unsigned)(XP_PRE + lane) : PT_OOB, xch_rslot(x, mm < m ? mm : 0));
      }
#pragma unroll
      for (int u = 0; u < 8; u++) carry += t8[u];
    }
    if (lane < S) {
      double cy[PT_NW];
#pragma unroll
      for (int w2 = 0; w2 < PT_NW; w2++) cy[w2] = X[w2 * SE + lane];
      ISSUE_FENCE();
#pragma unroll
      for (int w2 = 0; w2 < PT_NW; w2++) carry += w2 < w ? cy[w2] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < CL_DW; j++) {
      const int t = d0 + w * CL_DW + j;
      pol.g_fin(voz[j], (t < T - 1 ? pre[j] + carry : 0.0) - zq[j], zq[j], gz[j]);
    }
  }
  if (w == 1) {
    if (full) {
      // carry of the adjoint from the members that own later days, then raw_e_bias of the member's days
      const unsigned vA = lane < K ? (unsigned)lane * (unsigned)x.XW * 8u + 8u * (unsigned)XP_AR : PT_OOB;
      const double mA = bld_s(x.xb, vA, xch_rslot(x, 0)), mB = bld_s(x.xb, vA + 8u, xch_rslot(x, 0));
      double a_in = 0.0;
      for (int mm = K - 1; mm > m; mm--) a_in = readlane_d(mB, mm) + readlane_d(mA, mm) * a_in;
      const double a = arB + arA * a_in;
      const int t = d0 + lane;
      const double z = s_mid[M->o_ze - o_c + (arl ? t : 0)];
      const double gv = a * (t >= 1 ? s_scal[SC_SRHO] : M->sigma_e) - z;
      lp -= arl ? 0.5 * z * z : 0.0;               // stan:125
      pol.gs_fin(vo_x, gv, z, gx);
    }
  } else if (w >= 2) {
    // owned slots of the small vectors: sum the K partials in member order
    double sum = 0.0, sum2 = 0.0, sum3 = 0.0;
    const bool is_mue = full && repl && rslot == NR - 2, is_rho = full && repl && rslot == NR - 1;
    const unsigned v1 = !repl ? PT_OOB : is_mue ? 8u * (unsigned)XP_S : is_rho ? 8u * (unsigned)(XP_S + 1) : 8u * (unsigned)(XP_P + rslot);
    const unsigned v2 = is_rho ? 8u * (unsigned)(XP_S + 2) : PT_OOB;
    for (int mm0 = 0; mm0 < K; mm0 += 8) {
      double t8[8], u8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int mm = mm0 + u;
        t8[u] = bld_s(x.xb, mm < K ? v1 : PT_OOB, xch_rslot(x, mm < K ? mm : 0));
        u8[u] = bld_s(x.xb, mm < K ? v2 : PT_OOB, xch_rslot(x, mm < K ? mm : 0));
      }
#pragma unroll
      for (int u = 0; u < 8; u++) { sum += t8[u]; sum2 += u8[u]; }
    }
    (void)sum3;
    const double qv = s_rep[rslot];
    double gv = scale_r * sum - qv;
    double dl = -0.5 * qv * qv;                    // stan:117,120-122,128
    if (full && (is_mue || is_rho)) {
      const double rho = s_scal[SC_RHO];
      if (is_mue) {
        gv = 0.02 * (1.0 - rho) * sum - qv;       // sum = S1
        dl = log(0.02) - 0.5 * qv * qv;           // Jacobian of mu_e_bias (stan:62) + its prior (stan:123)
      } else {
        const double adj_rho = sum + sum2 * M->sigma_e * (-rho / sqrt(1.0 - rho * rho));   // S2, S3
        gv = (adj_rho - (rho - 0.7) / 0.01) * rho * (1.0 - rho) + (1.0 - 2.0 * rho);
        dl = log(rho) + log1p(-rho) - 0.5 * ((rho - 0.7) / 0.1) * ((rho - 0.7) / 0.1);      // stan:63,124
      }
    }
    lp += repl ? dl : 0.0;
    pol.gs_fin(vo_x, gv, qv, gx);
  }
  PROF_MARK(18);
  double v[1 + Pol::NEXTRA];
  v[0] = lp;
#pragma unroll
  for (int k = 0; k < Pol::NEXTRA; k++) v[1 + k] = pol.extra[k];
  cl_allreduce(v, red, x, tid, PROFPTR);
  PROF_MARK(19);
#pragma unroll
  for (int k = 0; k < Pol::NEXTRA; k++) pol_io.extra[k] = v[1 + k];
  return v[0];
}

// ================================================================ NUTS on a cluster
// Same algorithm and RNG contract as potus_nuts.hpp (Stan 2.24 base_nuts / adapt_diag_e_nuts); the
// vector sweeps run over the member's own elements and their dot products are all-reduced.
struct ClChain {
  CMp M;
  CCp CL;
  cip part;
  ldp lds;
  ltp ts;
  gsc sc;                  // this member's replica of the chain scalars (all replicas hold the same bits)
  RngKey key;
  ClStatic cst;
  rsrc_t st;               // the chain's state block [V_COUNT][Dpad], internal element order
  gcip perm;
  Xch x;
  int D, Dpad, tid, e0, e1, max_depth, num_warmup, init_buffer, term_buffer;
  double delta, gamma, kappa, t0;
#ifdef POTUS_PROF
  ldp prof;
#endif
  __device__ __forceinline__ unsigned soff(int slot) const { return (unsigned)slot * (unsigned)Dpad * 8u; }
  __device__ __forceinline__ ldp red() const { return lds + CL->l_red; }
};

#define CL_UNR 4
__device__ __forceinline__ int cl_first(const ClChain &c) { int t = c.e0 + c.tid; asm volatile("" : "+v"(t)); return t; }

template <bool SHARED_DST>
__device__ __forceinline__ void cl_vop_copy(const ClChain &c, unsigned s_dst, unsigned s_src) {
  for (int base = cl_first(c); base < c.e1; base += CL_UNR * PT_THREADS) {
    double v[CL_UNR];
#pragma unroll
    for (int k = 0; k < CL_UNR; k++) { const int i = base + k * PT_THREADS; v[k] = bld(c.st, i < c.e1 ? 8u * i : PT_OOB, s_src); }
#pragma unroll
    for (int k = 0; k < CL_UNR; k++) {
      const int i = base + k * PT_THREADS;
      if (SHARED_DST) bst_s(c.st, i < c.e1 ? 8u * i : PT_OOB, s_dst, v[k]);
      else bst(c.st, i < c.e1 ? 8u * i : PT_OOB, s_dst, v[k]);
    }
  }
  __syncthreads();
}
// diag_e_metric::sample_p: element with Stan index i takes normal (i & 1) of Philox block i >> 1, as the
// one-workgroup sampler does; returns sum_i minv_i p_i^2 over the whole chain
__device__ __forceinline__ double cl_vop_momentum(ClChain &c, unsigned sP, uint32_t iter, uint32_t purpose, uint32_t aux) {
  const unsigned sM = c.soff(V_MINV);
  double v[1] = {0.0};
  for (int i = cl_first(c); i < c.e1; i += PT_THREADS) {
    const int si = c.perm[i];
    const double mi = bld(c.st, 8u * i, sM);
    double a, b;
    rng_normal_pair(c.key, iter, purpose, aux, (uint32_t)(si >> 1), a, b);
    const double n = (si & 1) ? b : a;
    bst(c.st, 8u * i, sP, n / sqrt(mi));
    v[0] += n * n;
  }
  cl_allreduce(v, c.red(), c.x, c.tid, CPROFPTR(c));
  return v[0];
}
__device__ __forceinline__ bool cl_vop_merge(ClChain &c, unsigned a_beg, unsigned a_end, unsigned a_rho, unsigned b_beg, unsigned b_end,
                                             unsigned b_rho, unsigned out) {
  const unsigned sM = c.soff(V_MINV);
  double v[6] = {0, 0, 0, 0, 0, 0};
  for (int base = cl_first(c); base < c.e1; base += CL_UNR * PT_THREADS) {
    double mi[CL_UNR], ab[CL_UNR], ae[CL_UNR], ar[CL_UNR], bb[CL_UNR], be[CL_UNR], br[CL_UNR];
#pragma unroll
    for (int k = 0; k < CL_UNR; k++) {
      const int i = base + k * PT_THREADS;
      const unsigned o = i < c.e1 ? 8u * i : PT_OOB;   // masked elements read zeros and add nothing
      mi[k] = bld(c.st, o, sM); ab[k] = bld(c.st, o, a_beg); ae[k] = bld(c.st, o, a_end); ar[k] = bld(c.st, o, a_rho);
      bb[k] = bld(c.st, o, b_beg); be[k] = bld(c.st, o, b_end); br[k] = bld(c.st, o, b_rho);
    }
#pragma unroll
    for (int k = 0; k < CL_UNR; k++) {
      const int i = base + k * PT_THREADS;
      const double rs = ar[k] + br[k];
      bst(c.st, i < c.e1 ? 8u * i : PT_OOB, out, rs);
      const double sab = mi[k] * ab[k], sbe = mi[k] * be[k];
      v[0] += sab * rs;                 // p#_beg . rho_subtree
      v[1] += sbe * rs;                 // p#_end . rho_subtree
      const double e1 = ar[k] + bb[k];  // rho_init + p_final_beg
      v[2] += sab * e1;
      v[3] += mi[k] * bb[k] * e1;
      const double e2 = br[k] + ae[k];  // rho_final + p_init_end
      v[4] += mi[k] * ae[k] * e2;
      v[5] += sbe * e2;
    }
  }
  cl_allreduce(v, c.red(), c.x, c.tid, CPROFPTR(c));
  return v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0 && v[4] > 0 && v[5] > 0;
}
// PH[e] = p + he*g ; position buffer dst = q + e*minv*PH[e] ; PF[e] = p.  Ends with a cluster barrier:
// the next pass of every member reads the new position of the small vectors.
__device__ __forceinline__ void cl_vop_prekick(ClChain &c, unsigned sq, unsigned sp, unsigned sg, unsigned s_ph, unsigned s_dst, unsigned s_pf,
                                               double he, double e) {
  const unsigned sM = c.soff(V_MINV);
  for (int base = cl_first(c); base < c.e1; base += CL_UNR * PT_THREADS) {
    double q[CL_UNR], p[CL_UNR], g[CL_UNR], m[CL_UNR];
#pragma unroll
    for (int k = 0; k < CL_UNR; k++) {
      const int i = base + k * PT_THREADS;
      const unsigned o = i < c.e1 ? 8u * i : PT_OOB;
      q[k] = bld(c.st, o, sq); p[k] = bld(c.st, o, sp); g[k] = bld(c.st, o, sg); m[k] = bld(c.st, o, sM);
    }
#pragma unroll
    for (int k = 0; k < CL_UNR; k++) {
      const int i = base + k * PT_THREADS;
      const unsigned o = i < c.e1 ? 8u * i : PT_OOB;
      const double ph = p[k] + he * g[k];
      bst(c.st, o, s_ph, ph);
      bst_s(c.st, o, s_dst, q[k] + e * m[k] * ph);
      bst(c.st, o, s_pf, p[k]);
    }
  }
  cl_sync(c.x);
}

__device__ __forceinline__ void cl_transition_begin(ClChain &c, uint32_t iter) {
  ltp ts = c.ts;
  const int tid = c.tid;
  const double eps = c.sc->nom_eps; // sample_stepsize(): no jitter
  const double kin0 = cl_vop_momentum(c, c.soff(V_PC), iter, RNG_MOMENTUM, 0);
  ClPlainPolicy pp{c.st, c.st, c.soff(V_QC), c.soff(V_GC), {0}};
  const double lp0 = cl_pass(c.M, c.CL, c.part, c.lds, c.cst, c.x, pp); // hamiltonian.init
  if (tid == 0) {
    ts->H0 = 0.5 * kin0 - lp0;
    ts->lsw = 0.0; ts->sum_metro = 0.0; ts->n_leap = 0; ts->depth = 0; ts->divergent = 0; ts->stop = 0; ts->eps = eps;
    ts->qsel[0] = 0; ts->qsel[1] = 0;
    unsigned qm = 0;
    const int id = pool_alloc(qm, PT_NPQ);
    ts->qmask = qm;
    ts->sample_qid = id; ts->q_lp[id] = lp0; ts->q_h[id] = 0.5 * kin0 - lp0;
  }
  __syncthreads();
  {
    const unsigned s_rt = c.soff(V_RHOTOP), s_qs = c.soff(V_POOLQ + ts->sample_qid);
    for (int base = cl_first(c); base < c.e1; base += CL_UNR * PT_THREADS) {
      double q[CL_UNR], p[CL_UNR];
#pragma unroll
      for (int k = 0; k < CL_UNR; k++) {
        const int i = base + k * PT_THREADS;
        const unsigned o = i < c.e1 ? 8u * i : PT_OOB;
        q[k] = bld(c.st, o, c.soff(V_QC)); p[k] = bld(c.st, o, c.soff(V_PC));
      }
#pragma unroll
      for (int k = 0; k < CL_UNR; k++) {
        const int i = base + k * PT_THREADS;
        const unsigned o = i < c.e1 ? 8u * i : PT_OOB;
        bst(c.st, o, s_rt, p[k]); bst(c.st, o, s_qs, q[k]);
      }
    }
  }
  cl_vop_prekick(c, c.soff(V_QC), c.soff(V_PC), c.soff(V_GC), c.soff(V_PH1), c.soff(V_QA1), c.soff(V_PF1), 0.5 * eps, eps);
  cl_vop_prekick(c, c.soff(V_QC), c.soff(V_PC), c.soff(V_GC), c.soff(V_PH0), c.soff(V_QA0), c.soff(V_PF0), -0.5 * eps, -eps);
}

__device__ __forceinline__ void cl_transition_tree(ClChain &c, uint32_t iter) {
  ltp ts = c.ts;
  const int tid = c.tid;
  const double eps = ts->eps;
  while (true) {
    __syncthreads();
    if (ts->depth >= c.max_depth || ts->stop) break;
    const int depth = ts->depth;
    if (tid == 0) {
      ts->dir = rng_uniform(c.key, iter, RNG_DIRECTION, 0, (uint32_t)depth) > 0.5 ? 1 : 0;
      ts->pmask = 0;
      ts->qmask = 1u << ts->sample_qid;
    }
    __syncthreads();
    const int dir = ts->dir;
    CPROF_START(c);
    cl_vop_copy<false>(c, c.soff(V_PNEAR), c.soff(V_PF0 + dir));
    CPROF_MARK(c, PF_PNEAR);
    bool valid = true;
    const int nleaf = 1 << depth;
    for (int n = 0; n < nleaf; n++) {
      if (tid == 0) { unsigned pm = ts->pmask; ts->leaf_id = pool_alloc(pm, PT_NPP); ts->pmask = pm; }
      __syncthreads();
      const double e = dir ? eps : -eps;
      const int sel = ts->qsel[dir];              // buffer holding this leaf's position
      const unsigned s_leaf = c.soff(V_POOLP + ts->leaf_id);
      ClLeapPolicy lp{c.st, c.soff((sel ? V_QB0 : V_QA0) + dir), c.soff((sel ? V_QA0 : V_QB0) + dir), c.soff(V_PH0 + dir), c.soff(V_MINV),
                      s_leaf, 0.5 * e, e, {0.0}};
      const double lpv = cl_pass(c.M, c.CL, c.part, c.lds, c.cst, c.x, lp);
      CPROF_START(c);
      CPROF_COUNT(c, PF_LEAVES);
      if (tid == 0) {
        const double H0 = ts->H0;
        double h = 0.5 * lp.extra[0] - lpv;
        if (isnan(h)) h = INFINITY;
        const int div = (h - H0 > 1000.0) ? 1 : ts->divergent;
        ts->divergent = div;
        const double wgt = H0 - h;
        ts->sum_metro += wgt > 0 ? 1.0 : exp(wgt);
        ts->n_leap += 1;
        ts->cur_beg = ts->cur_end = ts->leaf_id;
        ts->cur_lsw = wgt; ts->cur_prop = -1; ts->cur_lp = lpv; ts->cur_h = h;
        ts->abort = div;
        ts->m = __builtin_ctz(~(unsigned)n);
        ts->qsel[dir] = sel ^ 1;                  // the next leaf of this end reads the buffer just written
      }
      __syncthreads();
      CPROF_MARK(c, PF_LEAF_SCALAR);
      if (ts->abort) { valid = false; break; }
      const int m = ts->m;
      for (int j = 1; j <= m; j++) {
        const int ib = ts->pend_beg[j - 1], ie = ts->pend_end[j - 1], cb = ts->cur_beg, ce = ts->cur_end;
        const unsigned a_rho = j == 1 ? c.soff(V_POOLP + ib) : c.soff(V_RHOLEV + j - 1);
        const unsigned b_rho = j == 1 ? c.soff(V_POOLP + cb) : c.soff(V_SCR0 + ((j - 1) & 1));
        const unsigned out = j == m ? c.soff(V_RHOLEV + j) : c.soff(V_SCR0 + (j & 1));
        const bool persist = cl_vop_merge(c, c.soff(V_POOLP + ib), c.soff(V_POOLP + ie), a_rho, c.soff(V_POOLP + cb),
                                          c.soff(V_POOLP + ce), b_rho, out);
        __syncthreads();
        if (tid == 0) {
          const double cur_lsw = ts->cur_lsw;
          const double lsw_sub = d_lse(ts->pend_lsw[j - 1], cur_lsw);
          bool take_final;
          if (cur_lsw > lsw_sub) take_final = true;
          else {
            const uint32_t slot = ((uint32_t)depth << 24) | ((uint32_t)j << 16) | (uint32_t)(n >> j);
            take_final = rng_uniform(c.key, iter, RNG_SUB_ACCEPT, 0, slot) < exp(cur_lsw - lsw_sub);
          }
          unsigned qm = ts->qmask, pm = ts->pmask;
          if (take_final) pool_free(qm, ts->pend_prop[j - 1]);
          else { pool_free(qm, ts->cur_prop); ts->cur_prop = ts->pend_prop[j - 1]; }
          if (ie != ib) pool_free(pm, ie);
          if (cb != ce) pool_free(pm, cb);
          ts->qmask = qm; ts->pmask = pm;
          ts->cur_beg = ib;
          ts->cur_lsw = lsw_sub;
          ts->abort = !persist;
        }
        __syncthreads();
        CPROF_COUNT(c, PF_MERGES);
        CPROF_MARK(c, PF_MERGE);
        if (ts->abort) { valid = false; break; }
      }
      if (!valid) break;
      if (tid == 0) {
        int cq = -1, prop = ts->cur_prop;
        if (prop < 0) { // the leaf itself is this subtree's proposal: keep its position
          unsigned qm = ts->qmask;
          const int id = pool_alloc(qm, PT_NPQ);
          ts->qmask = qm;
          ts->q_lp[id] = ts->cur_lp; ts->q_h[id] = ts->cur_h;
          prop = id; cq = id;
        }
        ts->copy_q_id = cq;
        ts->pend_beg[m] = ts->cur_beg; ts->pend_end[m] = ts->cur_end; ts->pend_lsw[m] = ts->cur_lsw; ts->pend_prop[m] = prop;
      }
      __syncthreads();
      if (ts->copy_q_id >= 0) cl_vop_copy<false>(c, c.soff(V_POOLQ + ts->copy_q_id), c.soff((sel ? V_QB0 : V_QA0) + dir));
      CPROF_MARK(c, PF_COPYQ);
    }
    if (!valid) break;
    // merge the finished subtree with the old trajectory (the checks at the end of transition())
    const int nb = ts->pend_beg[depth], ne = ts->pend_end[depth];
    cl_vop_copy<false>(c, c.soff(V_PF0 + dir), c.soff(V_POOLP + ne));   // the last leaf is the new end point
    const unsigned n_rho = depth == 0 ? c.soff(V_POOLP + nb) : c.soff(V_RHOLEV + depth);
    const bool persist = cl_vop_merge(c, c.soff(V_PF1 - dir), c.soff(V_PNEAR), c.soff(V_RHOTOP), c.soff(V_POOLP + nb), c.soff(V_POOLP + ne),
                                      n_rho, c.soff(V_RHOTOP));
    __syncthreads();
    if (tid == 0) {
      ts->depth = depth + 1;
      const double lsw_sub = ts->pend_lsw[depth], lsw = ts->lsw;
      bool accept;
      if (lsw_sub > lsw) accept = true;
      else accept = rng_uniform(c.key, iter, RNG_TOP_ACCEPT, 0, (uint32_t)depth) < exp(lsw_sub - lsw);
      unsigned qm = ts->qmask;
      if (accept) { pool_free(qm, ts->sample_qid); ts->sample_qid = ts->pend_prop[depth]; }
      else pool_free(qm, ts->pend_prop[depth]);
      ts->qmask = qm;
      ts->lsw = d_lse(lsw, lsw_sub);
      if (!persist) ts->stop = 1;
    }
    CPROF_MARK(c, PF_MERGE);
  }
  __syncthreads();
}

// base_hmc::init_stepsize; the chain's point is QC with gradient GC (already evaluated).
__device__ __forceinline__ void cl_init_stepsize(ClChain &c, uint32_t iter) {
  ltp ts = c.ts;
  const int tid = c.tid;
  const double lp0 = c.sc->lp_cur;
  if (tid == 0) { ts->done = 0; ts->direction = 0; }
  __syncthreads();
  {
    const double e0 = c.sc->nom_eps;
    if (e0 == 0 || e0 > 1e7 || isnan(e0)) return;
  }
  for (uint32_t attempt = 0;; attempt++) {
    const double eps = c.sc->nom_eps;
    const double kin0 = cl_vop_momentum(c, c.soff(V_PC), iter, RNG_INIT_EPS, attempt);
    const double H0 = 0.5 * kin0 - lp0;
    cl_vop_prekick(c, c.soff(V_QC), c.soff(V_PC), c.soff(V_GC), c.soff(V_PH1), c.soff(V_QA1), c.soff(V_PF1), 0.5 * eps, eps);
    ClLeapPolicy lp{c.st, c.soff(V_QA1), c.soff(V_QB1), c.soff(V_PH1), c.soff(V_MINV), c.soff(V_SCR0), 0.5 * eps, eps, {0.0}};
    const double lpv = cl_pass(c.M, c.CL, c.part, c.lds, c.cst, c.x, lp);
    __syncthreads();
    if (tid == 0) {
      double h = 0.5 * lp.extra[0] - lpv;
      if (isnan(h)) h = INFINITY;
      const double delta_H = H0 - h, thr = log(0.8);
      if (attempt == 0) ts->direction = delta_H > thr ? 1 : -1;
      else {
        const int dirn = ts->direction;
        if (dirn == 1 && !(delta_H > thr)) ts->done = 1;
        else if (dirn == -1 && !(delta_H < thr)) ts->done = 1;
        else {
          const double ne = dirn == 1 ? 2.0 * eps : 0.5 * eps;
          c.sc->nom_eps = ne;
          if (ne > 1e7 || ne == 0) { ts->done = 1; c.sc->status = 2; } // upstream throws here
        }
      }
    }
    __syncthreads();
    if (ts->done) break;
  }
  __syncthreads();
}

// adapt_diag_e_nuts::transition; the new sample is already the chain's point QC.
__device__ __forceinline__ void cl_adapt_after_transition(ClChain &c, uint32_t iter) {
  ltp ts = c.ts;
  gsc sc = c.sc;
  const int tid = c.tid;
  if (tid == 0) {
    const double cnt = sc->ad_counter + 1;       // stepsize_adaptation::learn_stepsize
    sc->ad_counter = cnt;
    const double as = ts->accept_stat > 1 ? 1.0 : ts->accept_stat;
    const double eta = 1.0 / (cnt + c.t0);
    const double s_bar = (1.0 - eta) * sc->s_bar + eta * (c.delta - as);
    sc->s_bar = s_bar;
    const double xx = sc->mu - s_bar * sqrt(cnt) / c.gamma;
    const double x_eta = pow(cnt, -c.kappa);
    sc->x_bar = (1.0 - x_eta) * sc->x_bar + x_eta * xx;
    sc->nom_eps = exp(xx);
    const int nw = c.num_warmup, ib = c.init_buffer, tb = c.term_buffer, wc = sc->win_counter;   // var_adaptation::learn_variance
    int fa = 0, fb = 0;
    if (nw >= 20) {
      fa = wc >= ib && wc < nw - tb && wc != nw;
      fb = wc == sc->win_next && wc != nw;
      if (fa) sc->wf_n += 1;
    }
    ts->flag_a = fa; ts->flag_b = fb;
  }
  __syncthreads();
  const int in_window = ts->flag_a, end_window = ts->flag_b;
  const unsigned sQ = c.soff(V_QC), sMean = c.soff(V_WMEAN), sM2 = c.soff(V_WM2), sMinv = c.soff(V_MINV);
  if (in_window) { // welford_var_estimator::add_sample
    const double n = sc->wf_n;
    for (int i = cl_first(c); i < c.e1; i += PT_THREADS) {
      const double q = bld(c.st, 8u * i, sQ), mo = bld(c.st, 8u * i, sMean), delta = q - mo, mn = mo + delta / n;
      bst(c.st, 8u * i, sMean, mn);
      bst(c.st, 8u * i, sM2, bld(c.st, 8u * i, sM2) + (q - mn) * delta);
    }
  }
  if (end_window) {
    const double n = sc->wf_n;
    for (int i = cl_first(c); i < c.e1; i += PT_THREADS) {
      const double var = bld(c.st, 8u * i, sM2) / (n - 1.0);
      bst(c.st, 8u * i, sMinv, (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0)));
      bst(c.st, 8u * i, sMean, 0.0); bst(c.st, 8u * i, sM2, 0.0);
    }
  }
  __syncthreads();
  if (tid == 0 && c.num_warmup >= 20) {
    if (end_window) { // windowed_adaptation::compute_next_window
      const int last = c.num_warmup - c.term_buffer - 1;
      int wn = sc->win_next, wsz = sc->win_size;
      if (wn != last) {
        wsz *= 2;
        wn = sc->win_counter + wsz;
        if (wn != last) {
          const int boundary = wn + 2 * wsz;
          if (boundary >= c.num_warmup - c.term_buffer) wn = last;
        }
      }
      sc->win_next = wn; sc->win_size = wsz;
      sc->wf_n = 0;
    }
    sc->win_counter += 1;
  }
  __syncthreads();
  if (end_window) {
    ClPlainPolicy pol{c.st, c.st, c.soff(V_QC), c.soff(V_GC), {0}};
    const double lpq = cl_pass(c.M, c.CL, c.part, c.lds, c.cst, c.x, pol);
    if (tid == 0) sc->lp_cur = lpq;
    __syncthreads();
    cl_init_stepsize(c, iter);
    if (tid == 0) { sc->mu = log(10.0 * sc->nom_eps); sc->s_bar = 0; sc->x_bar = 0; sc->ad_counter = 0; }
    __syncthreads();
  }
  if (tid == 0 && (int)iter == c.num_warmup - 1) sc->nom_eps = exp(sc->x_bar); // complete_adaptation
  __syncthreads();
}
