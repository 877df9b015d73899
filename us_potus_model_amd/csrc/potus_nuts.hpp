// potus_nuts.hpp -- device-resident NUTS with windowed diagonal-metric adaptation.
//
// One 512-thread workgroup (8 wave64, PT_NW) runs ONE chain for any number of transitions without returning
// to the host: momentum refresh, leapfrogs (fused with the model pass), the multinomial
// tree with Stan's generalised U-turn checks, step-size dual averaging, Welford variance
// windows and the step-size re-initialisation all happen inside the kernel.  Chains never
// communicate (exactly as the reference, where each chain is a separate CmdStan process:
// scripts/model/final_2016.R:533-541), so chains on different workgroups run free of each
// other -- no lock-step, no divergence cost from different tree depths.
//
// The algorithm is Stan 2.24's (third-party sources, cited by upstream file name):
//   base_nuts.hpp::transition / build_tree   -> nuts_transition (iterative, see below)
//   expl_leapfrog.hpp, diag_e_metric.hpp      -> LeapPolicy fused into model_pass
//   stepsize_adaptation.hpp                   -> learn_stepsize
//   windowed_adaptation.hpp, var_adaptation.hpp, welford_var_estimator.hpp -> adapt_*
//   base_hmc.hpp::init_stepsize               -> init_stepsize
// build_tree's recursion is restated as a loop over leaves with a cascade of merges after
// every leaf (a subtree of level j is merged when its last leaf has j trailing one bits), so
// only O(depth) vectors are live:
//   - every leaf writes its momentum once into a small pool slot;
//   - a pending (left-child) subtree keeps {first/last leaf slot, rho, log weight, proposal};
//   - one fused sweep per merge produces the six dot products of the three U-turn checks and
//     the merged rho;
//   - proposals are pool slots selected by index, copied only when a leaf survives its merges.
// Randomness is Philox4x32-10 keyed by (seed, chain) with the (iteration, purpose, slot)
// counter, so results do not depend on scheduling or on how chains are split across GPUs.
#pragma once
#include "potus_model.hpp"

extern __shared__ __attribute__((aligned(16))) double lds_dyn[];

#define PT_MAXD 12
#define PT_NPP (2 * PT_MAXD + 6)
#define PT_NPQ (PT_MAXD + 6)

enum { RNG_MOMENTUM = 0, RNG_DIRECTION = 1, RNG_TOP_ACCEPT = 2, RNG_SUB_ACCEPT = 3, RNG_INIT_EPS = 4, RNG_INITS = 5 };
#define PT_ITER_PRE 0xFFFFFFFFu

// vector slots of one chain's state block (each Dpad doubles).
// The chain's point lives in QC/GC.  Each end e of the trajectory (0 = backward, 1 = forward) keeps
//   QA[e], QB[e]  ping-pong positions: one holds the position the NEXT leapfrog of this end will
//                 evaluate, the other still holds the previous leaf's position (proposal copies)
//   PH[e]         momentum already kicked half a step towards that next position
//   PF[e]         full-step momentum at the end point (for the U-turn checks across subtrees)
// so a leapfrog reads one vector (the position) in the model pass and finishes with
// {PH, minv} -> {leaf slot, PH, next position}: the gradient itself never goes to memory.
enum {
  V_QC = 0, V_GC, V_PC, V_QA0, V_QA1, V_QB0, V_QB1, V_PH0, V_PH1, V_PF0, V_PF1,
  V_MINV, V_RHOTOP, V_PNEAR, V_WMEAN, V_WM2, V_SCR0, V_SCR1,
  V_RHOLEV, /* PT_MAXD+1 */
  V_POOLP = V_RHOLEV + PT_MAXD + 1,
  V_POOLQ = V_POOLP + PT_NPP,
  V_COUNT = V_POOLQ + PT_NPQ
};

// profile slots (POTUS_PROF builds): 0-6 model pass phases, then
enum { PF_MOMENTUM = 8, PF_INITCOPY, PF_LEAF_SCALAR, PF_MERGE, PF_COPYQ, PF_PNEAR, PF_ADAPT, PF_SAVE, PF_LEAVES, PF_MERGES };

struct ChainScalars { // persistent per chain, global memory
  double nom_eps, mu, s_bar, x_bar, ad_counter, wf_n, lp_cur;
  long long total_leapfrogs;
  int iter, win_counter, win_next, win_size, status, n_divergent, saved;
  int leaves_run;          // twin mode: leaves this side has integrated, those of dropped (speculative) subtrees included
  int spec_limit;          // twin mode: doublings (combines taken) of the last four transitions, a byte each: a doubling beyond their
                           // maximum is not started ahead of its turn
  int xcd_local;           // cluster mode: what the last launch found (cl_find_local): 1 = every member of the cluster on one XCD, exchange words published with plain stores
};
typedef ChainScalars AS_G *gsc;

struct RunParams {
  int chains, chain_id_offset, num_warmup, num_samples, max_depth, init_buffer, term_buffer, window;
  int save_warmup, n_save_max, Dpad, row;
  double delta, gamma, kappa, t0, stepsize, init_radius;
  unsigned seed_lo, seed_hi;
  double *state;           // [chains][V_COUNT][Dpad]
  ChainScalars *scal;      // [chains]
  double *draws;           // [chains][n_save_max][7 + D]
  double *prof;            // [chains][PT_NPROF] (POTUS_PROF builds) or null
  double *xbuf;            // cluster mode: [chains][2][K][XW] exchange words of 16 bytes {value, tag}
  unsigned *xcnt;          // cluster mode: [chains][64] arrival counters (one cache line apart)
  int K;                   // workgroups per chain (1: potus_nuts.hpp; > 1: potus_cluster.hpp, scal is [chains][K])
  int twin;                // cluster mode: two clusters per chain, one per end of the trajectory (state, scal, xbuf hold 2 x chains
                           // blocks: side s of chain c uses block c + s * chains); twbuf = their mailbox
  double *twbuf;           // [chains][TWB_WORDS] words of 16 bytes {value, tag}
  int debug_drop_member;   // test hook (POTUS_DEBUG_DROP_MEMBER = m + 1): member m of every cluster leaves k_cl_run at once, the
                           // rest must find out through the watchdog (tests/test_gpu_parity.py); -1: nobody leaves, but the exchange words are
                           // published write-through whatever the placement (the two store paths must give the same bytes)
};

typedef const RunParams AS_C *CRp;

// make a value provably wave-uniform (arguments of non-inlined functions arrive in VGPRs)
__device__ __forceinline__ unsigned uni32(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
template <class P> __device__ __forceinline__ P uni_ptr(P p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = uni32((unsigned)v), hi = uni32((unsigned)(v >> 32));
  return (P)(((unsigned long long)hi << 32) | lo);
}

struct TS { // transition state, LDS
  double H0, lsw, sum_metro, eps;
  double cur_lsw, cur_lp, cur_h;
  double pend_lsw[PT_MAXD + 1];
  double q_lp[PT_NPQ], q_h[PT_NPQ];
  double accept_stat, out_lp, out_h, delta_H;
  double u_sub[2][PT_MAXD + 1], u_top;   // cluster mode: uniforms of a leaf's accept steps, drawn ahead by idle lanes (by leaf parity)
  int pend_beg[PT_MAXD + 1], pend_end[PT_MAXD + 1], pend_prop[PT_MAXD + 1];
  int cur_beg, cur_end, cur_prop;
  unsigned pmask, qmask;
  int depth, dir, divergent, abort, m, leaf_id, copy_q_id, sample_qid, n_leap, stop;
  int flag_a, flag_b, direction, done;
  int qsel[2];   // which of QA/QB holds the position the next leapfrog of end e evaluates (0 = QA)
  int nextq[2], out_q;   // cluster mode: proposal-pool slots holding the next position of each end / receiving this leaf's output
  // two clusters per chain (potus_cluster.hpp, "twin" mode): this side's copy of the trajectory-level state (TT_* words) as of
  // combine number tw_seq; the pool slot it must keep (its latest accepted proposal, or the initial point); the directions
  // of the transition's doublings (bit j = doubling j goes forward); whether the last combine ended the trajectory
  double tt[12];
  double metro_base;     // cluster mode: the trajectory's sum of metropolis terms before the current doubling
  int tw_seq, tw_keep, tw_dirs, tw_over, tw_ext, tw_pad;   // tw_ext: the other side's STOP word was up when this leaf started
};
typedef TS AS_L *ltp;

// ---------------------------------------------------------------- RNG
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  const uint64_t x = ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
  return ((double)x + 0.5) * (1.0 / 9007199254740992.0);
}
struct RngKey { uint32_t k0, k1, chain; };
__device__ __forceinline__ double rng_uniform(const RngKey &K, uint32_t iter, uint32_t purpose, uint32_t aux, uint32_t index) {
  uint32_t o[4];
  philox4x32_10(index, purpose | (aux << 8), iter, K.chain, K.k0, K.k1, o);
  return u53(o[0], o[1]);
}
__device__ __forceinline__ void rng_normal_pair(const RngKey &K, uint32_t iter, uint32_t purpose, uint32_t aux, uint32_t index,
                                                double &n0, double &n1) {
  uint32_t o[4];
  philox4x32_10(index, purpose | (aux << 8), iter, K.chain, K.k0, K.k1, o);
  const double u1 = u53(o[0], o[1]), u2 = u53(o[2], o[3]);
  const double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincos(6.283185307179586476925286766559 * u2, &s, &c);
  n0 = r * c; n1 = r * s;
}

// ---------------------------------------------------------------- leapfrog fused into the model pass
// expl_leapfrog: p -= eps/2 dV/dq ; q += eps M^-1 p ; (V, dV/dq)(q) ; p -= eps/2 dV/dq, with dV/dq = -grad lp.
// Consecutive leapfrogs of one trajectory end always use the same signed step, so the second
// half-kick of one step and the first half-kick and drift of the next are applied together as
// soon as the gradient g is known (phase F of the pass):
//     pf  = ph + he*g          full-step momentum of THIS leaf  -> leaf slot
//     ph' = pf + he*g          half-kicked momentum for the next leaf -> PH
//     q'  = q + e*minv*ph'     position the next leaf evaluates -> the other ping-pong buffer
// The arithmetic per leaf is exactly Stan's; only the order of stores differs.
// A leaf that completes a pair (odd leaf of its subtree) also does the pair's U-turn check here, where its momentum is in
// registers: with p' the previous leaf's momentum, rho = p' + pf goes to the slot the level-1 merge would have written, and the six
// dot products of that merge collapse to two (begin = end = rho of a one-leaf subtree): extra[1] = p'.M^-1.rho, extra[2] =
// pf.M^-1.rho.  That removes half of the merge sweeps (7 vector loads each, a quarter of the one-workgroup leaf's time).
struct LeapPolicy {
  rsrc_t r;                          // the chain's state block
  unsigned sQc, sQn, sPH, sM, sL;    // byte offsets: position (this leaf / next), PH, inverse metric, leaf slot
  double he, e;
  unsigned sPrev, sRho;              // the previous leaf's momentum slot, the slot receiving rho of the pair
  int pair;                          // 1: this leaf completes a pair (else nothing is read from sPrev or written to sRho)
  static constexpr int NEXTRA = 3;
#ifndef POTUS_LEAP_GB
#define POTUS_LEAP_GB 4
#endif
  static constexpr int GB = POTUS_LEAP_GB;
  double extra[3];
  struct QT { double q; };
  struct GT { double p, m, pp; };
  __device__ __forceinline__ void q_load(unsigned vo, QT &t) { t.q = bld(r, vo, sQc); }
  __device__ __forceinline__ double q_fin(unsigned, QT &t) { return t.q; }
  __device__ __forceinline__ void g_load(unsigned vo, GT &t) { t.p = bld(r, vo, sPH); t.m = bld(r, vo, sM); t.pp = bld(r, pair ? vo : PT_OOB, sPrev); }
  __device__ __forceinline__ void g_fin(unsigned vo, double v, double q, const GT &t) {
    const double pf = t.p + he * v;
    bst(r, vo, sL, pf);
    const double ph = pf + he * v;
    bst(r, vo, sPH, ph);
    bst(r, vo, sQn, q + e * t.m * ph);
    extra[0] += t.m * pf * pf;   // masked-off elements loaded m = 0
    const double rs = t.pp + pf; // (the order vop_merge adds them in: rho_init + rho_final)
    bst(r, pair ? vo : PT_OOB, sRho, rs);
    extra[1] += t.m * t.pp * rs;
    extra[2] += t.m * pf * rs;
  }
  __device__ __forceinline__ double q(int i) { return bld(r, 8u * i, sQc); }
  __device__ __forceinline__ void g(int i, double v, double q) { GT t; g_load(8u * i, t); g_fin(8u * i, v, q, t); }
};

struct Chain {
  CMp M;
  ldp lds;
  ltp ts;
  gdp base;
  gsc sc;
  RngKey key;
  PassStatic pst;
  int D, Dpad, tid, max_depth, num_warmup, init_buffer, term_buffer;
  double delta, gamma, kappa, t0;
#ifdef POTUS_PROF
  ldp prof;
#endif
  rsrc_t st;   // buffer resource over this chain's whole state block
  __device__ __forceinline__ gdp vec(int slot) const { return base + (size_t)slot * Dpad; }
  // (slot numbers read from LDS are wave-uniform; readfirstlane tells the compiler, which otherwise wraps every access
  // with such a scalar offset in a waterfall loop)
  __device__ __forceinline__ unsigned soff(int slot) const { return __builtin_amdgcn_readfirstlane((unsigned)slot * (unsigned)Dpad * 8u); }
  __device__ __forceinline__ ldp red() const { return lds + M->l_red; }
};
#ifdef POTUS_PROF
#define CPROF_MARK(c, k) do { if ((c).tid == 0) { const long long t_ = clock64(); (c).prof[k] += (double)(t_ - (long long)(c).prof[PT_NPROF - 1]); (c).prof[PT_NPROF - 1] = (double)t_; } } while (0)
#define CPROF_START(c) do { if ((c).tid == 0) (c).prof[PT_NPROF - 1] = (double)clock64(); } while (0)
#define CPROF_COUNT(c, k) do { if ((c).tid == 0) (c).prof[k] += 1.0; } while (0)
#else
#define CPROF_MARK(c, k) do { } while (0)
#define CPROF_START(c) do { } while (0)
#define CPROF_COUNT(c, k) do { } while (0)
#endif

__device__ __forceinline__ double d_lse(double a, double b) {
  if (a == -INFINITY) return b;
  if (a == INFINITY && b == INFINITY) return INFINITY;
  return a > b ? a + log1p(exp(b - a)) : b + log1p(exp(a - b));
}
__device__ __forceinline__ int pool_alloc(unsigned &mask, int n) {
  for (int i = 0; i < n; i++) if (!(mask & (1u << i))) { mask |= 1u << i; return i; }
  return n - 1; // cannot happen: pools are sized for PT_MAXD
}
__device__ __forceinline__ void pool_free(unsigned &mask, int i) { if (i >= 0) mask &= ~(1u << i); }

// ---------------------------------------------------------------- block-wide vector sweeps (all end with a barrier)
// Each thread handles elements tid, tid+PT_THREADS, ...; PT_UNR elements are in flight per trip so the
// loads of a trip are issued together (the sweeps are latency-, not bandwidth-limited).
#define PT_UNR 8
__device__ __forceinline__ int fresh_tid(const Chain &c) { // keeps per-thread index math inside the loop it belongs to
  int t = c.tid;
  asm volatile("" : "+v"(t));
  return t;
}
__device__ __forceinline__ void vop_copy(const Chain &c, unsigned s_dst, unsigned s_src) {
  const int tid0 = fresh_tid(c);
  for (int base = tid0; base < c.D; base += PT_UNR * PT_THREADS) {
    double v[PT_UNR];
#pragma unroll
    for (int k = 0; k < PT_UNR; k++) { const int i = base + k * PT_THREADS; v[k] = bld(c.st, i < c.D ? 8u * i : PT_OOB, s_src); }
#pragma unroll
    for (int k = 0; k < PT_UNR; k++) { const int i = base + k * PT_THREADS; bst(c.st, i < c.D ? 8u * i : PT_OOB, s_dst, v[k]); }
  }
  __syncthreads();
}
// diag_e_metric::sample_p: p_i = N(0,1) / sqrt(minv_i); returns sum_i minv_i p_i^2
__device__ __forceinline__ double vop_momentum(const Chain &c, unsigned sP, uint32_t iter, uint32_t purpose, uint32_t aux) {
  const unsigned sM = c.soff(V_MINV);
  double v[1] = {0.0};
  const int tid0 = fresh_tid(c);
  for (int j = tid0; 2 * j < c.D; j += PT_THREADS) {
    const bool two = 2 * j + 1 < c.D;
    const double m0 = bld(c.st, 16u * j, sM), m1 = two ? bld(c.st, 16u * j + 8u, sM) : 1.0;
    double a, b;
    rng_normal_pair(c.key, iter, purpose, aux, (uint32_t)j, a, b);
    bst(c.st, 16u * j, sP, a / sqrt(m0));
    v[0] += a * a;
    if (two) { bst(c.st, 16u * j + 8u, sP, b / sqrt(m1)); v[0] += b * b; }
  }
  block_sum(v, c.red(), tid0);
  return v[0];
}
// One merge of an (init, final) pair of subtrees: the three checks of base_nuts::build_tree /
// transition need six metric-weighted dot products; also emits rho_init + rho_final.
__device__ __forceinline__ bool vop_merge(const Chain &c, unsigned a_beg, unsigned a_end, unsigned a_rho, unsigned b_beg, unsigned b_end,
                                          unsigned b_rho, unsigned out) {
  const unsigned sM = c.soff(V_MINV);
  double v[6] = {0, 0, 0, 0, 0, 0};
  const int tid0 = fresh_tid(c);
  for (int base = tid0; base < c.D; base += PT_UNR * PT_THREADS) {
    double mi[PT_UNR], ab[PT_UNR], ae[PT_UNR], ar[PT_UNR], bb[PT_UNR], be[PT_UNR], br[PT_UNR];
#pragma unroll
    for (int k = 0; k < PT_UNR; k++) {
      const int i = base + k * PT_THREADS;
      const unsigned o = i < c.D ? 8u * i : PT_OOB;   // masked elements read zeros and add nothing
      mi[k] = bld(c.st, o, sM); ab[k] = bld(c.st, o, a_beg); ae[k] = bld(c.st, o, a_end); ar[k] = bld(c.st, o, a_rho);
      bb[k] = bld(c.st, o, b_beg); be[k] = bld(c.st, o, b_end); br[k] = bld(c.st, o, b_rho);
    }
#pragma unroll
    for (int k = 0; k < PT_UNR; k++) {
      const int i = base + k * PT_THREADS;
      {
        const double rs = ar[k] + br[k];
        bst(c.st, i < c.D ? 8u * i : PT_OOB, out, rs);
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
  }
  block_sum(v, c.red(), tid0);
  return v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0 && v[4] > 0 && v[5] > 0;
}

// ---------------------------------------------------------------- one NUTS transition (base_nuts::transition)
// On return ts->sample_qid names the pool slot holding the new sample, ts->out_lp / out_h its
// log density and Hamiltonian, ts->accept_stat the adaptation statistic.
// Kick-and-drift from a point (q, p, grad lp g) towards the first leaf of end e:
//   PH[e] = p + he*g ; position buffer `dst` = q + e*minv*PH[e] ; optionally PF[e] = p.
__device__ __forceinline__ void vop_prekick(const Chain &c, unsigned sq, unsigned sp, unsigned sg, unsigned s_ph, unsigned s_dst,
                                            unsigned s_pf, double he, double e) {
  const unsigned sM = c.soff(V_MINV);
  const int tid0 = fresh_tid(c);
  for (int base = tid0; base < c.D; base += PT_UNR * PT_THREADS) {
    double q[PT_UNR], p[PT_UNR], g[PT_UNR], m[PT_UNR];
#pragma unroll
    for (int k = 0; k < PT_UNR; k++) {
      const int i = base + k * PT_THREADS;
      const unsigned o = i < c.D ? 8u * i : PT_OOB;
      q[k] = bld(c.st, o, sq); p[k] = bld(c.st, o, sp); g[k] = bld(c.st, o, sg); m[k] = bld(c.st, o, sM);
    }
#pragma unroll
    for (int k = 0; k < PT_UNR; k++) {
      const int i = base + k * PT_THREADS;
      const unsigned o = i < c.D ? 8u * i : PT_OOB;
      const double ph = p[k] + he * g[k];
      bst(c.st, o, s_ph, ph);
      bst(c.st, o, s_dst, q[k] + e * m[k] * ph);
      bst(c.st, o, s_pf, p[k]);
    }
  }
  __syncthreads();
}

// Part 1 (once per transition, kept out of line): momentum refresh, Hamiltonian at the start
// point, both trajectory ends := start point, kicked towards their first leaves.
__device__ __forceinline__ void transition_begin(const Chain &c, uint32_t iter) {
  ltp ts = c.ts;
  const int tid = c.tid;
  const double eps = c.sc->nom_eps; // sample_stepsize(): no jitter
  CPROF_START(c);
  const double kin0 = vop_momentum(c, c.soff(V_PC), iter, RNG_MOMENTUM, 0);
  CPROF_MARK(c, PF_MOMENTUM);
  PlainPolicy pp{c.st, c.st, c.soff(V_QC), c.soff(V_GC), {0}};
  const double lp0 = model_pass(c.M, c.lds, c.pst, pp); // hamiltonian.init
  CPROF_START(c);
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
  vop_prekick(c, c.soff(V_QC), c.soff(V_PC), c.soff(V_GC), c.soff(V_PH1), c.soff(V_QA1), c.soff(V_PF1), 0.5 * eps, eps);
  vop_prekick(c, c.soff(V_QC), c.soff(V_PC), c.soff(V_GC), c.soff(V_PH0), c.soff(V_QA0), c.soff(V_PF0), -0.5 * eps, -eps);
  {
    const unsigned s_rt = c.soff(V_RHOTOP), s_qs = c.soff(V_POOLQ + ts->sample_qid);
    const int tid0 = fresh_tid(c);
    for (int base = tid0; base < c.D; base += PT_UNR * PT_THREADS) {
      double q[PT_UNR], p[PT_UNR];
#pragma unroll
      for (int k = 0; k < PT_UNR; k++) {
        const int i = base + k * PT_THREADS;
        const unsigned o = i < c.D ? 8u * i : PT_OOB;
        q[k] = bld(c.st, o, c.soff(V_QC)); p[k] = bld(c.st, o, c.soff(V_PC));
      }
#pragma unroll
      for (int k = 0; k < PT_UNR; k++) {
        const int i = base + k * PT_THREADS;
        const unsigned o = i < c.D ? 8u * i : PT_OOB;
        bst(c.st, o, s_rt, p[k]); bst(c.st, o, s_qs, q[k]);
      }
    }
  }
  CPROF_MARK(c, PF_INITCOPY);
  __syncthreads();
}

// Part 2 (the hot loop, inlined into the kernel): doublings, leaves, merges.
__device__ __forceinline__ void transition_tree(const Chain &c, uint32_t iter) {
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
      // the metropolis terms of a doubling are summed on their own and then added to the trajectory's sum -- the order the
      // two-workgroup form (potus_nuts_twin.hpp) has to use: with it the two produce the same bytes
      ts->metro_base = ts->sum_metro; ts->sum_metro = 0.0;
    }
    __syncthreads();
    const int dir = ts->dir;
    CPROF_START(c);
    vop_copy(c, c.soff(V_PNEAR), c.soff(V_PF0 + dir));
    CPROF_MARK(c, PF_PNEAR);
    bool valid = true;
    const int nleaf = 1 << depth;
    for (int n = 0; n < nleaf; n++) {
      if (tid == 0) { unsigned pm = ts->pmask; ts->leaf_id = pool_alloc(pm, PT_NPP); ts->pmask = pm; }
      __syncthreads();
      const double e = dir ? eps : -eps;
      const int sel = ts->qsel[dir];              // buffer holding this leaf's position
      const unsigned s_leaf = c.soff(V_POOLP + ts->leaf_id);
      const int pair = n & 1, m_leaf = __builtin_ctz(~(unsigned)n);
      LeapPolicy lp{c.st, c.soff((sel ? V_QB0 : V_QA0) + dir), c.soff((sel ? V_QA0 : V_QB0) + dir), c.soff(V_PH0 + dir), c.soff(V_MINV),
                    s_leaf, 0.5 * e, e, c.soff(V_POOLP + (pair ? ts->pend_beg[0] : ts->leaf_id)), m_leaf == 1 ? c.soff(V_RHOLEV + 1) : c.soff(V_SCR1),
                    pair, {0.0, 0.0, 0.0}};
      const double lpv = model_pass(c.M, c.lds, c.pst, lp);
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
        // (level 1 came out of the leaf's epilogue: begin = end = rho on both sides, so v0 = v2 = v4 and v1 = v3 = v5)
        const bool persist = j == 1 ? (lp.extra[1] > 0 && lp.extra[2] > 0)
                                    : vop_merge(c, c.soff(V_POOLP + ib), c.soff(V_POOLP + ie), a_rho, c.soff(V_POOLP + cb), c.soff(V_POOLP + ce), b_rho, out);
        CPROF_COUNT(c, PF_MERGES);
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
      if (ts->copy_q_id >= 0) vop_copy(c, c.soff(V_POOLQ + ts->copy_q_id), c.soff((sel ? V_QB0 : V_QA0) + dir));
      CPROF_MARK(c, PF_COPYQ);
    }
    if (tid == 0) ts->sum_metro += ts->metro_base;
    if (!valid) break;
    // merge the finished subtree with the old trajectory (the checks at the end of transition())
    const int nb = ts->pend_beg[depth], ne = ts->pend_end[depth];
    vop_copy(c, c.soff(V_PF0 + dir), c.soff(V_POOLP + ne));   // the last leaf is the new end point
    const unsigned n_rho = depth == 0 ? c.soff(V_POOLP + nb) : c.soff(V_RHOLEV + depth);
    const bool persist = vop_merge(c, c.soff(V_PF1 - dir), c.soff(V_PNEAR), c.soff(V_RHOTOP), c.soff(V_POOLP + nb), c.soff(V_POOLP + ne),
                                   n_rho, c.soff(V_RHOTOP));
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

// ---------------------------------------------------------------- base_hmc::init_stepsize
// Works on end 1's buffers as scratch; the chain's point is QC with gradient GC (already evaluated).
__device__ __forceinline__ void init_stepsize(const Chain &c, uint32_t iter) {
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
    const double kin0 = vop_momentum(c, c.soff(V_PC), iter, RNG_INIT_EPS, attempt);
    const double H0 = 0.5 * kin0 - lp0;
    vop_prekick(c, c.soff(V_QC), c.soff(V_PC), c.soff(V_GC), c.soff(V_PH1), c.soff(V_QA1), c.soff(V_PF1), 0.5 * eps, eps);
    LeapPolicy lp{c.st, c.soff(V_QA1), c.soff(V_QB1), c.soff(V_PH1), c.soff(V_MINV), c.soff(V_SCR0), 0.5 * eps, eps, c.soff(V_SCR0), c.soff(V_SCR1), 0, {0.0, 0.0, 0.0}};
    const double lpv = model_pass(c.M, c.lds, c.pst, lp);
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
          if (ne > 1e7 || ne == 0) { ts->done = 1; c.sc->status = POTUS_ERR_STEPSIZE; } // upstream throws here
        }
      }
    }
    __syncthreads();
    if (ts->done) break;
  }
  __syncthreads();
}

// ---------------------------------------------------------------- adaptation (adapt_diag_e_nuts::transition)
// qs: the new sample, already stored as the chain's point Q0 (its gradient G0 may be stale).
__device__ __forceinline__ void adapt_after_transition(const Chain &c, uint32_t iter, gcdp qs) {
  ltp ts = c.ts;
  gsc sc = c.sc;
  const int tid = c.tid;
  if (tid == 0) {
    // stepsize_adaptation::learn_stepsize
    const double cnt = sc->ad_counter + 1;
    sc->ad_counter = cnt;
    const double as = ts->accept_stat > 1 ? 1.0 : ts->accept_stat;
    const double eta = 1.0 / (cnt + c.t0);
    const double s_bar = (1.0 - eta) * sc->s_bar + eta * (c.delta - as);
    sc->s_bar = s_bar;
    const double x = sc->mu - s_bar * sqrt(cnt) / c.gamma;
    const double x_eta = pow(cnt, -c.kappa);
    sc->x_bar = (1.0 - x_eta) * sc->x_bar + x_eta * x;
    sc->nom_eps = exp(x);
    // var_adaptation::learn_variance window logic
    const int nw = c.num_warmup, ib = c.init_buffer, tb = c.term_buffer, wc = sc->win_counter;
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
  gdp mean = c.vec(V_WMEAN), m2 = c.vec(V_WM2), minv = c.vec(V_MINV);
  if (in_window) { // welford_var_estimator::add_sample
    const double n = sc->wf_n;
    for (int i = tid; i < c.D; i += PT_THREADS) {
      const double q = qs[i], mo = mean[i], delta = q - mo, mn = mo + delta / n;
      mean[i] = mn;
      m2[i] += (q - mn) * delta;
    }
  }
  if (end_window) {
    const double n = sc->wf_n;
    for (int i = tid; i < c.D; i += PT_THREADS) {
      const double var = m2[i] / (n - 1.0);
      minv[i] = (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0));
      mean[i] = 0.0; m2[i] = 0.0;
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
    // init_stepsize starts from the current point: refresh its log density and gradient
    PlainPolicy pol{c.st, c.st, c.soff(V_QC), c.soff(V_GC), {0}};
    const double lpq = model_pass(c.M, c.lds, c.pst, pol);
    if (tid == 0) sc->lp_cur = lpq;
    __syncthreads();
    init_stepsize(c, iter);
    if (tid == 0) { sc->mu = log(10.0 * sc->nom_eps); sc->s_bar = 0; sc->x_bar = 0; sc->ad_counter = 0; }
    __syncthreads();
  }
  if (tid == 0 && (int)iter == c.num_warmup - 1) sc->nom_eps = exp(sc->x_bar); // complete_adaptation
  __syncthreads();
}
