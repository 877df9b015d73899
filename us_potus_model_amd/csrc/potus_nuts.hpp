// potus_nuts.hpp -- device-resident NUTS with windowed diagonal-metric adaptation.
//
// One 1024-thread workgroup runs ONE chain for any number of transitions without returning
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

#define PT_MAXD 12
#define PT_NPP (2 * PT_MAXD + 4)
#define PT_NPQ (PT_MAXD + 4)

enum { RNG_MOMENTUM = 0, RNG_DIRECTION = 1, RNG_TOP_ACCEPT = 2, RNG_SUB_ACCEPT = 3, RNG_INIT_EPS = 4, RNG_INITS = 5 };
#define PT_ITER_PRE 0xFFFFFFFFu

// vector slots of one chain's state block (each Dpad doubles)
enum {
  V_Q0 = 0, V_Q1, V_P0, V_P1, V_G0, V_G1, V_MINV, V_RHOTOP, V_PNEAR, V_WMEAN, V_WM2, V_SCR0, V_SCR1,
  V_RHOLEV, /* PT_MAXD+1 */
  V_POOLP = V_RHOLEV + PT_MAXD + 1,
  V_POOLQ = V_POOLP + PT_NPP,
  V_COUNT = V_POOLQ + PT_NPQ
};

struct ChainScalars { // persistent per chain, global memory
  double nom_eps, mu, s_bar, x_bar, ad_counter, wf_n, lp_cur;
  long long total_leapfrogs;
  int iter, win_counter, win_next, win_size, status, n_divergent, saved, pad;
};

struct RunParams {
  int chains, chain_id_offset, num_warmup, num_samples, max_depth, init_buffer, term_buffer, window;
  int save_warmup, n_save_max, Dpad, row;
  double delta, gamma, kappa, t0, stepsize, init_radius;
  unsigned seed_lo, seed_hi;
  double *state;           // [chains][V_COUNT][Dpad]
  ChainScalars *scal;      // [chains]
  double *draws;           // [chains][n_save_max][7 + D]
};

struct TS { // transition state, LDS
  double H0, lsw, sum_metro, eps;
  double cur_lsw, cur_lp, cur_h;
  double pend_lsw[PT_MAXD + 1];
  double q_lp[PT_NPQ], q_h[PT_NPQ];
  double accept_stat, out_lp, out_h, delta_H;
  int pend_beg[PT_MAXD + 1], pend_end[PT_MAXD + 1], pend_prop[PT_MAXD + 1];
  int cur_beg, cur_end, cur_prop;
  unsigned pmask, qmask;
  int depth, dir, divergent, abort, m, leaf_id, copy_q_id, sample_qid, n_leap, stop;
  int flag_a, flag_b, direction, done;
};

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
// G holds grad lp.  The full-step momentum is also written to the leaf's pool slot and the
// kinetic energy sum_i minv_i p_i^2 is accumulated for the Hamiltonian.
struct LeapPolicy {
  double *Q, *P, *G;
  const double *minv;
  double *leafp;
  double he, e;
  static constexpr int NEXTRA = 1;
  double extra[1];
  __device__ __forceinline__ double q(int i) {
    const double ph = P[i] + he * G[i];
    P[i] = ph;
    const double qn = Q[i] + e * minv[i] * ph;
    Q[i] = qn;
    return qn;
  }
  __device__ __forceinline__ double q_again(int i) { return Q[i]; }
  __device__ __forceinline__ void g(int i, double v) {
    G[i] = v;
    const double pf = P[i] + he * v;
    P[i] = pf;
    leafp[i] = pf;
    extra[0] += minv[i] * pf * pf;
  }
};

struct Chain {
  const DevModel &M;
  const RunParams &R;
  double *lds;
  TS &ts;
  double *base;
  ChainScalars *sc;
  RngKey key;
  int D, tid;
  __device__ __forceinline__ double *vec(int slot) const { return base + (size_t)slot * R.Dpad; }
};

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
__device__ void vop_copy(const Chain &c, double *dst, const double *src) {
  for (int i = c.tid; i < c.D; i += PT_THREADS) dst[i] = src[i];
  __syncthreads();
}
// diag_e_metric::sample_p: p_i = N(0,1) / sqrt(minv_i); returns sum_i minv_i p_i^2
__device__ double vop_momentum(const Chain &c, double *P, uint32_t iter, uint32_t purpose, uint32_t aux) {
  const double *minv = c.vec(V_MINV);
  double v[1] = {0.0};
  for (int j = c.tid; 2 * j < c.D; j += PT_THREADS) {
    double a, b;
    rng_normal_pair(c.key, iter, purpose, aux, (uint32_t)j, a, b);
    const double pa = a / sqrt(minv[2 * j]);
    P[2 * j] = pa;
    v[0] += a * a;
    if (2 * j + 1 < c.D) { P[2 * j + 1] = b / sqrt(minv[2 * j + 1]); v[0] += b * b; }
  }
  block_sum(v, c.lds + c.M.l_red, c.tid);
  return v[0];
}
// One merge of an (init, final) pair of subtrees: the three checks of base_nuts::build_tree /
// transition need six metric-weighted dot products; also emits rho_init + rho_final.
__device__ bool vop_merge(const Chain &c, const double *a_beg, const double *a_end, const double *a_rho, const double *b_beg,
                          const double *b_end, const double *b_rho, double *out) {
  const double *minv = c.vec(V_MINV);
  double v[6] = {0, 0, 0, 0, 0, 0};
  for (int i = c.tid; i < c.D; i += PT_THREADS) {
    const double mi = minv[i];
    const double ab = a_beg[i], ae = a_end[i], ar = a_rho[i], bb = b_beg[i], be = b_end[i], br = b_rho[i];
    const double rs = ar + br;
    out[i] = rs;
    const double sab = mi * ab, sbe = mi * be;
    v[0] += sab * rs;            // p#_beg . rho_subtree
    v[1] += sbe * rs;            // p#_end . rho_subtree
    const double e1 = ar + bb;   // rho_init + p_final_beg
    v[2] += sab * e1;
    v[3] += mi * bb * e1;
    const double e2 = br + ae;   // rho_final + p_init_end
    v[4] += mi * ae * e2;
    v[5] += sbe * e2;
  }
  block_sum(v, c.lds + c.M.l_red, c.tid);
  return v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0 && v[4] > 0 && v[5] > 0;
}

// ---------------------------------------------------------------- one NUTS transition (base_nuts::transition)
// On return ts.sample_qid names the pool slot holding the new sample, ts.out_lp / out_h its
// log density and Hamiltonian, ts.accept_stat the adaptation statistic.
__device__ void nuts_transition(const Chain &c, uint32_t iter) {
  TS &ts = c.ts;
  const int tid = c.tid;
  double *Q[2] = {c.vec(V_Q0), c.vec(V_Q1)}, *P[2] = {c.vec(V_P0), c.vec(V_P1)}, *G[2] = {c.vec(V_G0), c.vec(V_G1)};
  const double eps = c.sc->nom_eps; // sample_stepsize(): no jitter

  const double kin0 = vop_momentum(c, P[0], iter, RNG_MOMENTUM, 0);
  PlainPolicy pp{Q[0], G[0], {0}};
  const double lp0 = model_pass(c.M, c.lds, pp); // hamiltonian.init
  if (tid == 0) {
    ts.H0 = 0.5 * kin0 - lp0;
    ts.lsw = 0.0; ts.sum_metro = 0.0; ts.n_leap = 0; ts.depth = 0; ts.divergent = 0; ts.stop = 0; ts.eps = eps;
    ts.qmask = 0;
    const int id = pool_alloc(ts.qmask, PT_NPQ);
    ts.sample_qid = id; ts.q_lp[id] = lp0; ts.q_h[id] = ts.H0;
  }
  __syncthreads();
  {
    double *rt = c.vec(V_RHOTOP), *qs = c.vec(V_POOLQ + ts.sample_qid);
    for (int i = tid; i < c.D; i += PT_THREADS) {
      const double q = Q[0][i], p = P[0][i];
      Q[1][i] = q; P[1][i] = p; G[1][i] = G[0][i]; rt[i] = p; qs[i] = q;
    }
  }
  while (true) {
    __syncthreads();
    if (ts.depth >= c.R.max_depth || ts.stop) break;
    const int depth = ts.depth;
    if (tid == 0) {
      ts.dir = rng_uniform(c.key, iter, RNG_DIRECTION, 0, (uint32_t)depth) > 0.5 ? 1 : 0;
      ts.pmask = 0;
      ts.qmask = 1u << ts.sample_qid;
    }
    __syncthreads();
    const int dir = ts.dir;
    vop_copy(c, c.vec(V_PNEAR), P[dir]);
    bool valid = true;
    const int nleaf = 1 << depth;
    for (int n = 0; n < nleaf; n++) {
      if (tid == 0) ts.leaf_id = pool_alloc(ts.pmask, PT_NPP);
      __syncthreads();
      const double e = dir ? eps : -eps;
      LeapPolicy lp{Q[dir], P[dir], G[dir], c.vec(V_MINV), c.vec(V_POOLP + ts.leaf_id), 0.5 * e, e, {0.0}};
      const double lpv = model_pass(c.M, c.lds, lp);
      if (tid == 0) {
        double h = 0.5 * lp.extra[0] - lpv;
        if (isnan(h)) h = INFINITY;
        if (h - ts.H0 > 1000.0) ts.divergent = 1;
        const double wgt = ts.H0 - h;
        ts.sum_metro += wgt > 0 ? 1.0 : exp(wgt);
        ts.n_leap++;
        ts.cur_beg = ts.cur_end = ts.leaf_id;
        ts.cur_lsw = wgt; ts.cur_prop = -1; ts.cur_lp = lpv; ts.cur_h = h;
        ts.abort = ts.divergent;
        ts.m = __builtin_ctz(~(unsigned)n);
      }
      __syncthreads();
      if (ts.abort) { valid = false; break; }
      const int m = ts.m;
      for (int j = 1; j <= m; j++) {
        const int ib = ts.pend_beg[j - 1], ie = ts.pend_end[j - 1], cb = ts.cur_beg, ce = ts.cur_end;
        const double *a_rho = j == 1 ? c.vec(V_POOLP + ib) : c.vec(V_RHOLEV + j - 1);
        const double *b_rho = j == 1 ? c.vec(V_POOLP + cb) : c.vec(V_SCR0 + ((j - 1) & 1));
        double *out = j == m ? c.vec(V_RHOLEV + j) : c.vec(V_SCR0 + (j & 1));
        const bool persist = vop_merge(c, c.vec(V_POOLP + ib), c.vec(V_POOLP + ie), a_rho, c.vec(V_POOLP + cb),
                                       c.vec(V_POOLP + ce), b_rho, out);
        if (tid == 0) {
          const double lsw_sub = d_lse(ts.pend_lsw[j - 1], ts.cur_lsw);
          bool take_final;
          if (ts.cur_lsw > lsw_sub) take_final = true;
          else {
            const uint32_t slot = ((uint32_t)depth << 24) | ((uint32_t)j << 16) | (uint32_t)(n >> j);
            take_final = rng_uniform(c.key, iter, RNG_SUB_ACCEPT, 0, slot) < exp(ts.cur_lsw - lsw_sub);
          }
          if (take_final) pool_free(ts.qmask, ts.pend_prop[j - 1]);
          else { pool_free(ts.qmask, ts.cur_prop); ts.cur_prop = ts.pend_prop[j - 1]; }
          if (ie != ib) pool_free(ts.pmask, ie);
          if (cb != ce) pool_free(ts.pmask, cb);
          ts.cur_beg = ib;
          ts.cur_lsw = lsw_sub;
          ts.abort = !persist;
        }
        __syncthreads();
        if (ts.abort) { valid = false; break; }
      }
      if (!valid) break;
      if (tid == 0) {
        ts.copy_q_id = -1;
        if (ts.cur_prop < 0) { // the leaf itself is this subtree's proposal: keep its position
          const int id = pool_alloc(ts.qmask, PT_NPQ);
          ts.q_lp[id] = ts.cur_lp; ts.q_h[id] = ts.cur_h;
          ts.cur_prop = id; ts.copy_q_id = id;
        }
        ts.pend_beg[m] = ts.cur_beg; ts.pend_end[m] = ts.cur_end; ts.pend_lsw[m] = ts.cur_lsw; ts.pend_prop[m] = ts.cur_prop;
      }
      __syncthreads();
      if (ts.copy_q_id >= 0) vop_copy(c, c.vec(V_POOLQ + ts.copy_q_id), Q[dir]);
    }
    if (!valid) break;
    // merge the finished subtree with the old trajectory (the checks at the end of transition())
    const int nb = ts.pend_beg[depth], ne = ts.pend_end[depth];
    const double *n_rho = depth == 0 ? c.vec(V_POOLP + nb) : c.vec(V_RHOLEV + depth);
    const bool persist = vop_merge(c, P[1 - dir], c.vec(V_PNEAR), c.vec(V_RHOTOP), c.vec(V_POOLP + nb), c.vec(V_POOLP + ne),
                                   n_rho, c.vec(V_RHOTOP));
    if (tid == 0) {
      ts.depth = depth + 1;
      const double lsw_sub = ts.pend_lsw[depth];
      bool accept;
      if (lsw_sub > ts.lsw) accept = true;
      else accept = rng_uniform(c.key, iter, RNG_TOP_ACCEPT, 0, (uint32_t)depth) < exp(lsw_sub - ts.lsw);
      if (accept) { pool_free(ts.qmask, ts.sample_qid); ts.sample_qid = ts.pend_prop[depth]; }
      else pool_free(ts.qmask, ts.pend_prop[depth]);
      ts.lsw = d_lse(ts.lsw, lsw_sub);
      if (!persist) ts.stop = 1;
    }
  }
  __syncthreads();
  if (tid == 0) {
    ts.accept_stat = ts.sum_metro / (double)ts.n_leap;
    ts.out_lp = ts.q_lp[ts.sample_qid];
    ts.out_h = ts.q_h[ts.sample_qid];
    c.sc->total_leapfrogs += ts.n_leap;
    c.sc->n_divergent += ts.divergent;
  }
  __syncthreads();
}

// ---------------------------------------------------------------- base_hmc::init_stepsize
// Works on end 1 as scratch; the chain's point is Q0 with gradient G0 (already evaluated).
__device__ void init_stepsize(const Chain &c, uint32_t iter) {
  TS &ts = c.ts;
  const int tid = c.tid;
  double *Q1 = c.vec(V_Q1), *P1 = c.vec(V_P1), *G1 = c.vec(V_G1);
  const double *Q0 = c.vec(V_Q0), *G0 = c.vec(V_G0);
  const double lp0 = c.sc->lp_cur;
  if (tid == 0) { ts.done = 0; ts.direction = 0; }
  __syncthreads();
  {
    const double e0 = c.sc->nom_eps;
    if (e0 == 0 || e0 > 1e7 || isnan(e0)) return;
  }
  for (uint32_t attempt = 0;; attempt++) {
    const double eps = c.sc->nom_eps;
    for (int i = tid; i < c.D; i += PT_THREADS) { Q1[i] = Q0[i]; G1[i] = G0[i]; }
    const double kin0 = vop_momentum(c, P1, iter, RNG_INIT_EPS, attempt);
    const double H0 = 0.5 * kin0 - lp0;
    LeapPolicy lp{Q1, P1, G1, c.vec(V_MINV), c.vec(V_SCR0), 0.5 * eps, eps, {0.0}};
    const double lpv = model_pass(c.M, c.lds, lp);
    if (tid == 0) {
      double h = 0.5 * lp.extra[0] - lpv;
      if (isnan(h)) h = INFINITY;
      const double delta_H = H0 - h, thr = log(0.8);
      if (attempt == 0) ts.direction = delta_H > thr ? 1 : -1;
      else {
        if (ts.direction == 1 && !(delta_H > thr)) ts.done = 1;
        else if (ts.direction == -1 && !(delta_H < thr)) ts.done = 1;
        else {
          const double ne = ts.direction == 1 ? 2.0 * eps : 0.5 * eps;
          c.sc->nom_eps = ne;
          if (ne > 1e7 || ne == 0) { ts.done = 1; c.sc->status = 2; } // upstream throws here
        }
      }
    }
    __syncthreads();
    if (ts.done) break;
  }
  __syncthreads();
}

// ---------------------------------------------------------------- adaptation (adapt_diag_e_nuts::transition)
// qs: the new sample, already stored as the chain's point Q0 (its gradient G0 may be stale).
__device__ void adapt_after_transition(const Chain &c, uint32_t iter, const double *qs) {
  TS &ts = c.ts;
  ChainScalars *sc = c.sc;
  const RunParams &R = c.R;
  const int tid = c.tid;
  if (tid == 0) {
    // stepsize_adaptation::learn_stepsize
    sc->ad_counter += 1;
    const double as = ts.accept_stat > 1 ? 1.0 : ts.accept_stat;
    const double eta = 1.0 / (sc->ad_counter + R.t0);
    sc->s_bar = (1.0 - eta) * sc->s_bar + eta * (R.delta - as);
    const double x = sc->mu - sc->s_bar * sqrt(sc->ad_counter) / R.gamma;
    const double x_eta = pow(sc->ad_counter, -R.kappa);
    sc->x_bar = (1.0 - x_eta) * sc->x_bar + x_eta * x;
    sc->nom_eps = exp(x);
    // var_adaptation::learn_variance window logic
    int nw = R.num_warmup, ib = R.init_buffer, tb = R.term_buffer;
    ts.flag_a = ts.flag_b = 0;
    if (nw >= 20) {
      ts.flag_a = sc->win_counter >= ib && sc->win_counter < nw - tb && sc->win_counter != nw;
      ts.flag_b = sc->win_counter == sc->win_next && sc->win_counter != nw;
      if (ts.flag_a) sc->wf_n += 1;
    }
  }
  __syncthreads();
  const int in_window = ts.flag_a, end_window = ts.flag_b;
  double *mean = c.vec(V_WMEAN), *m2 = c.vec(V_WM2), *minv = c.vec(V_MINV);
  if (in_window) { // welford_var_estimator::add_sample
    const double n = sc->wf_n;
    for (int i = tid; i < c.D; i += PT_THREADS) {
      const double q = qs[i], delta = q - mean[i], mn = mean[i] + delta / n;
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
  if (tid == 0 && R.num_warmup >= 20) {
    if (end_window) { // windowed_adaptation::compute_next_window
      const int last = R.num_warmup - R.term_buffer - 1;
      if (sc->win_next != last) {
        sc->win_size *= 2;
        sc->win_next = sc->win_counter + sc->win_size;
        if (sc->win_next != last) {
          const int boundary = sc->win_next + 2 * sc->win_size;
          if (boundary >= R.num_warmup - R.term_buffer) sc->win_next = last;
        }
      }
      sc->wf_n = 0;
    }
    sc->win_counter += 1;
  }
  __syncthreads();
  if (end_window) {
    // init_stepsize starts from the current point: refresh its log density and gradient
    PlainPolicy pol{c.vec(V_Q0), c.vec(V_G0), {0}};
    const double lpq = model_pass(c.M, c.lds, pol);
    if (tid == 0) sc->lp_cur = lpq;
    __syncthreads();
    init_stepsize(c, iter);
    if (tid == 0) { sc->mu = log(10.0 * sc->nom_eps); sc->s_bar = 0; sc->x_bar = 0; sc->ad_counter = 0; }
    __syncthreads();
  }
  if (tid == 0 && (int)iter == R.num_warmup - 1) sc->nom_eps = exp(sc->x_bar); // complete_adaptation
  __syncthreads();
}
