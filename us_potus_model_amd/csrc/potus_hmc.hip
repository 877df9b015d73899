// potus_hmc.hip -- libpotus_hmc.so: C ABI (include/potus_hmc.h) over the gfx950 kernels.
//
// Host side: validates the Stan data block, builds transformed data (stan:42-55) and the
// static poll schedule, owns device memory / stream / events behind an integer handle.
// Device side: potus_model.hpp (log-density + gradient), potus_nuts.hpp (NUTS + adaptation).
#include "../../include/potus_hmc.h"
#include "potus_nuts.hpp"
#include "potus_cluster.hpp"
#include "potus_nuts_twin.hpp"
#include "potus_dense.hpp"
#include "potus_summary.hpp"
#include "potus_diag.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <charconv>
#include <chrono>
#include <thread>
#include <mutex>
#include <string>
#include <vector>

// ======================================================================== kernels

__global__ __launch_bounds__(PT_THREADS) void k_logprob_grad(const DevModel *Mg, const double *q, double *lp, double *grad, int n) {
  CMp M = (CMp)Mg;
  ldp lds = (ldp)lds_dyn;
  const PassStatic pst = model_setup_lds(M, lds);
  const int D = M->D;
  for (int b = blockIdx.x; b < n; b += gridDim.x) {
    PlainPolicy pol{make_rsrc(q + (size_t)b * D, 8u * D), make_rsrc(grad + (size_t)b * D, 8u * D), 0u, 0u, {0}};
    const double v = model_pass(M, lds, pst, pol);
    if (threadIdx.x == 0) lp[b] = v;
  }
}

// (two workgroups per chain, potus_nuts_twin.hpp: side s of a chain works in block chain + s * chains of state and scalars)
__device__ __forceinline__ Chain make_chain(CMp M, CRp R, int chain, int side = 0) {
  ldp lds = (ldp)lds_dyn;
  Chain c;
  const int Dpad = R->Dpad;
  const int blk = chain + side * R->chains;
  double *state = R->state + (size_t)blk * V_COUNT * Dpad;
  c.M = M; c.lds = lds; c.ts = (ltp)(lds + M->lds_doubles);
  c.base = as_g(state);
  c.st = make_rsrc(state, (unsigned)V_COUNT * (unsigned)Dpad * 8u);
  c.sc = (gsc)(R->scal + blk);
  c.key = RngKey{R->seed_lo, R->seed_hi, (uint32_t)(R->chain_id_offset + chain + 1)};
  c.D = M->D; c.Dpad = Dpad; c.tid = (int)threadIdx.x; c.max_depth = R->max_depth; c.num_warmup = R->num_warmup;
  c.init_buffer = R->init_buffer; c.term_buffer = R->term_buffer;
  c.delta = R->delta; c.gamma = R->gamma; c.kappa = R->kappa; c.t0 = R->t0;
#ifdef POTUS_PROF
  c.prof = lds + M->l_prof;
#endif
  return c;
}

// Initial values (stan::services::util::initialize: U(-R,R), <= 100 attempts), unit metric,
// adaptation windows (windowed_adaptation), mu = log(10*stepsize), initial init_stepsize.
__global__ __launch_bounds__(PT_THREADS) void k_init(const DevModel *Mg, const RunParams *Rg, const double *q0) {
  CMp M = (CMp)Mg;
  CRp R = (CRp)Rg;
  const int chain = blockIdx.x % R->chains, side = blockIdx.x / R->chains;   // side 1: two workgroups per chain only
  Chain c = make_chain(M, R, chain, side);
  c.pst = model_setup_lds(M, c.lds);
  const int tid = c.tid;
  gdp Q0 = c.vec(V_QC), G0 = c.vec(V_GC), minv = c.vec(V_MINV), mean = c.vec(V_WMEAN), m2 = c.vec(V_WM2);
  for (int i = tid; i < c.D; i += PT_THREADS) { minv[i] = 1.0; mean[i] = 0.0; m2[i] = 0.0; }
  if (tid == 0) {
    gsc sc = c.sc;
    sc->nom_eps = R->stepsize; sc->mu = log(10.0 * R->stepsize); sc->s_bar = 0; sc->x_bar = 0; sc->ad_counter = 0;
    sc->wf_n = 0; sc->total_leapfrogs = 0; sc->iter = 0; sc->status = 0; sc->n_divergent = 0; sc->saved = 0; sc->leaves_run = 0; sc->spec_limit = 0x7f7f7f7f;
    sc->win_counter = 0; sc->win_size = R->window; sc->win_next = R->init_buffer + R->window - 1;
  }
  __syncthreads();
  bool ok = false;
  const double radius = R->init_radius;
  for (uint32_t attempt = 0; attempt < 100 && !ok; attempt++) {
    for (int i = tid; i < c.D; i += PT_THREADS) {
      if (q0) Q0[i] = as_g(q0)[(size_t)chain * c.D + i];
      else Q0[i] = radius * (2.0 * rng_uniform(c.key, PT_ITER_PRE, RNG_INITS, attempt, (uint32_t)i) - 1.0);
    }
    __syncthreads();
    PlainPolicy pol{c.st, c.st, c.soff(V_QC), c.soff(V_GC), {0}};
    const double lp = model_pass(M, c.lds, c.pst, pol);
    double bad[1] = {0.0};
    for (int i = tid; i < c.D; i += PT_THREADS) bad[0] += isfinite(G0[i]) ? 0.0 : 1.0;
    block_sum(bad, c.red(), tid);
    ok = isfinite(lp) && bad[0] == 0.0;
    if (tid == 0 && ok) c.sc->lp_cur = lp;
    __syncthreads();
    if (q0) break;
  }
  if (!ok) { if (tid == 0) c.sc->status = POTUS_ERR_INIT; return; }
  init_stepsize(c, PT_ITER_PRE);
  if (tid == 0 && R->num_warmup == 0) c.sc->nom_eps = exp(c.sc->x_bar); // engage+disengage: complete_adaptation
}

// The once-per-transition work is kept out of line so that the register allocator only sees the
// tree loop (leaves + merges) in the kernel body.  Arguments are re-made wave-uniform on entry.
__device__ __noinline__ void cold_transition_begin(const DevModel *Mg, const RunParams *Rg, int chain, uint32_t iter, int side = 0) {
  CMp M = (CMp)uni_ptr(Mg);
  CRp R = (CRp)uni_ptr(Rg);
  Chain c = make_chain(M, R, (int)uni32((unsigned)chain), (int)uni32((unsigned)side));
  c.pst = model_load_static(M);
  transition_begin(c, uni32(iter));
}

// New sample -> chain position and draws array; warmup adaptation.
__device__ __noinline__ void cold_transition_end(const DevModel *Mg, const RunParams *Rg, int chain_, uint32_t iter) {
  CMp M = (CMp)uni_ptr(Mg);
  CRp R = (CRp)uni_ptr(Rg);
  const int chain = (int)uni32((unsigned)chain_);
  const int it = (int)uni32(iter);
  Chain c = make_chain(M, R, chain);
  c.pst = model_load_static(M);
  ltp ts = c.ts;
  const int tid = c.tid;
  if (tid == 0) {
    ts->accept_stat = ts->sum_metro / (double)ts->n_leap;
    ts->out_lp = ts->q_lp[ts->sample_qid];
    ts->out_h = ts->q_h[ts->sample_qid];
    c.sc->total_leapfrogs += ts->n_leap;
    c.sc->n_divergent += ts->divergent;
  }
  __syncthreads();
  CPROF_START(c);
  gcdp qs = c.vec(V_POOLQ + ts->sample_qid);
  gdp Q0 = c.vec(V_QC);
  const bool warm = it < R->num_warmup;
  const bool save = !warm || R->save_warmup;
  const int row_len = R->row;
  gdp row = as_g(R->draws) + ((size_t)chain * R->n_save_max + c.sc->saved) * row_len;
  if (save && tid == 0) {
    row[0] = ts->out_lp; row[1] = ts->accept_stat; row[2] = ts->eps; row[3] = ts->depth; row[4] = ts->n_leap;
    row[5] = ts->divergent; row[6] = ts->out_h;
  }
  for (int i = tid; i < c.D; i += PT_THREADS) {
    const double v = qs[i];
    Q0[i] = v;
    if (save) row[POTUS_N_SAMPLER_COLS + i] = v;
  }
  __syncthreads();
  if (tid == 0) {
    c.sc->lp_cur = ts->out_lp;
    if (save) c.sc->saved += 1;
  }
  CPROF_MARK(c, PF_SAVE);
  if (warm) adapt_after_transition(c, (uint32_t)it, Q0);
  CPROF_MARK(c, PF_ADAPT);
  __syncthreads();
  if (tid == 0) c.sc->iter = it + 1;
  __syncthreads();
}

// n_iter transitions per chain, adaptation during warmup, draws appended to the draws array.
__global__ __launch_bounds__(PT_THREADS) void k_run(const DevModel *Mg, const RunParams *Rg, int n_iter) {
  CMp M = (CMp)Mg;
  CRp R = (CRp)Rg;
  const int chain = blockIdx.x;
  Chain c = make_chain(M, R, chain);
  if (c.sc->status != 0) return;
  c.pst = model_setup_lds(M, c.lds);
  const int total = R->num_warmup + R->num_samples;
  for (int k = 0; k < n_iter; k++) {
    const int it = c.sc->iter;
    if (it >= total) break;
    cold_transition_begin(Mg, Rg, chain, (uint32_t)it);
    transition_tree(c, (uint32_t)it);
    cold_transition_end(Mg, Rg, chain, (uint32_t)it);
  }
#ifdef POTUS_PROF
  if (R->prof) for (int i = c.tid; i < PT_NPROF; i += PT_THREADS) as_g(R->prof)[(size_t)chain * PT_NPROF + i] += c.prof[i];
#endif
}


// ------------------------------------------------------------------------ two workgroups per chain (potus_nuts_twin.hpp): the cold parts
// The combine of doubling d by the side that built its subtree (valid: the subtree completed without an internal U-turn or
// divergence and was not dropped).  Sets ts->tw_over when the trajectory is over.  Follows cl_cold_twin_combine below.
__device__ __noinline__ void cold_twin1_combine(const DevModel *Mg, const RunParams *Rg, int chain_, int side_, unsigned launch_, uint32_t iter_,
                                                int depth_, int valid_) {
  CMp M = (CMp)uni_ptr(Mg);
  CRp R = (CRp)uni_ptr(Rg);
  const int chain = (int)uni32((unsigned)chain_), side = (int)uni32((unsigned)side_), d = (int)uni32((unsigned)depth_);
  const int valid = (int)uni32((unsigned)valid_);
  const uint32_t iter = uni32(iter_);
  Chain c = make_chain(M, R, chain, side);
  const Xch x = tw1_watch(R, chain + side * R->chains, launch_);
  const Twin t = make_twin(R, chain, side, x.launch, iter);
  ltp ts = c.ts;
  const int tid = c.tid;
  __syncthreads();
  if (tid == 0) c.sc->leaves_run += ts->n_leap;       // whether the subtree ends up in the trajectory or not
  // 1. the state this combine starts from: published by the combine of doubling d - 1 (or the end of the trajectory)
  if (uni_i(ts->tw_seq) < d) {
    if (tid < 64) tw_catch_up(x, t, ts, d);
    __syncthreads();
  }
  if (uni_i((int)ts->tt[TT_STOP])) {                 // the trajectory ended before this doubling: the subtree is dropped
    if (tid == 0) ts->tw_over = 1;
    __syncthreads();
    return;
  }
  const int rho_side = uni_i((int)ts->tt[TT_RHOSIDE]), dirs = uni_i(ts->tw_dirs);
  int stop = 1;
  if (valid) {
    // 2. rho_top += rho_subtree and the three checks of transition() across the whole trajectory; the other end's momentum
    //    and (if the last combine was the other side's) rho_top come from the other side's state block, where they were
    //    stored write-through before that combine was published.  The subtree's last leaf becomes this side's end point.
    const bool other_moved = ((side ? ~dirs : dirs) & ((1 << d) - 1)) != 0;
    const rsrc_t rA = other_moved ? t.ost : c.st, rR = rho_side == 1 - side ? t.ost : c.st;
    const int leaf = uni_i(ts->pend_end[d]);
    const unsigned sM = c.soff(V_MINV), a_beg = c.soff(V_PF0 + (1 - side)), a_end = c.soff(V_PNEAR), s_rt = c.soff(V_RHOTOP);
    const unsigned b_beg = c.soff(V_POOLP + uni_i(ts->pend_beg[d])), b_end = c.soff(V_POOLP + leaf);
    const unsigned b_rho = d == 0 ? c.soff(V_POOLP + leaf) : c.soff(V_RHOLEV + d), s_pf = c.soff(V_PF0 + side);
    double v[6] = {0, 0, 0, 0, 0, 0};
    const int tid0 = fresh_tid(c);
    for (int base = tid0; base < c.D; base += PT_UNR * PT_THREADS) {
      double mi[PT_UNR], ab[PT_UNR], ae[PT_UNR], ar[PT_UNR], bb[PT_UNR], be[PT_UNR], br[PT_UNR];
#pragma unroll
      for (int k = 0; k < PT_UNR; k++) {
        const int i = base + k * PT_THREADS;
        const unsigned o = i < c.D ? 8u * i : PT_OOB;   // masked elements read zeros and add nothing
        mi[k] = bld(c.st, o, sM); ab[k] = bld_s(rA, o, a_beg); ae[k] = bld(c.st, o, a_end); ar[k] = bld_s(rR, o, s_rt);
        bb[k] = bld(c.st, o, b_beg); be[k] = bld(c.st, o, b_end); br[k] = bld(c.st, o, b_rho);
      }
#pragma unroll
      for (int k = 0; k < PT_UNR; k++) {
        const int i = base + k * PT_THREADS;
        const unsigned o = i < c.D ? 8u * i : PT_OOB;
        const double rs = ar[k] + br[k];
        bst_s(c.st, o, s_rt, rs);
        bst_s(c.st, o, s_pf, be[k]);
        const double sab = mi[k] * ab[k], sbe = mi[k] * be[k];
        v[0] += sab * rs;                 // p#_beg . rho
        v[1] += sbe * rs;                 // p#_end . rho
        const double e1 = ar[k] + bb[k];  // rho_old + p of the subtree's first leaf
        v[2] += sab * e1;
        v[3] += mi[k] * bb[k] * e1;
        const double e2 = br[k] + ae[k];  // rho_subtree + p of the old end
        v[4] += mi[k] * ae[k] * e2;
        v[5] += sbe * e2;
      }
    }
    drain_vmem();                         // the stores above are complete before anything is published (barriers in block_sum)
    block_sum(v, c.red(), tid0);
    const bool persist = v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0 && v[4] > 0 && v[5] > 0;
    stop = (!persist || d + 1 >= c.max_depth) ? 1 : 0;
  }
  __syncthreads();
  if (tid == 0) {
    // the subtree's leaves count whether it is valid or not (base_nuts: n_leapfrog_, sum_metro_prob, divergent_)
    ts->tt[TT_METRO] += ts->sum_metro; ts->tt[TT_NLEAP] += (double)ts->n_leap;
    if (ts->divergent) ts->tt[TT_DIV] = 1.0;
    if (valid) {
      const double lsw_top = ts->tt[TT_LSW], lsw_sub = ts->pend_lsw[d];
      const int prop = ts->pend_prop[d];
      bool accept;
      if (lsw_sub > lsw_top) accept = true;
      else accept = rng_uniform(c.key, iter, RNG_TOP_ACCEPT, 0, (uint32_t)d) < exp(lsw_sub - lsw_top);
      if (accept) {
        ts->tt[TT_SSIDE] = (double)side; ts->tt[TT_SSLOT] = (double)prop; ts->tt[TT_SLP] = ts->q_lp[prop]; ts->tt[TT_SH] = ts->q_h[prop];
        ts->tw_keep = prop;
      }
      ts->tt[TT_LSW] = d_lse(lsw_top, lsw_sub);
      ts->tt[TT_DEPTH] = (double)(d + 1);
      ts->tt[TT_RHOSIDE] = (double)side;
    }
    ts->tt[TT_STOP] = (double)stop;
    ts->tw_seq = d + 1; ts->tw_over = stop;
  }
  __syncthreads();
  // 3. publish: the state after combine d, and the STOP word if the trajectory is over
  if (tid < 64) {
    const double val = ts->tt[tid < TT_N ? tid : 0];
    tw_st(t, tid < TT_N ? 16u * (unsigned)(TWB_TOP + 16 * ((d + 1) & 1) + tid) : PT_OOB, val, t.ittag | (unsigned)(d + 2));
    tw_st(t, (stop && tid == 0) ? 16u * (unsigned)TWB_STOP : PT_OOB, (double)(d + 1), t.ittag);
  }
  __syncthreads();
}

// New sample -> this side's chain position (from its own pool or from the other side's QC), draws array (side 0),
// adaptation (both sides, same inputs, same bits), rendezvous with the other side.  Follows cl_cold_twin_end below.
__device__ __noinline__ void cold_twin1_end(const DevModel *Mg, const RunParams *Rg, int chain_, int side_, unsigned launch_, uint32_t iter_) {
  CMp M = (CMp)uni_ptr(Mg);
  CRp R = (CRp)uni_ptr(Rg);
  const int chain = (int)uni32((unsigned)chain_), side = (int)uni32((unsigned)side_), it = (int)uni32(iter_);
  Chain c = make_chain(M, R, chain, side);
  c.pst = model_load_static(M);
  const Xch x = tw1_watch(R, chain + side * R->chains, launch_);
  const Twin t = make_twin(R, chain, side, x.launch, (uint32_t)it);
  ltp ts = c.ts;
  const int tid = c.tid;
  __syncthreads();
  if (!uni_i(ts->tw_over)) {                          // this side ran out of doublings of its own: wait for the end
    if (tid < 64) tw_catch_up(x, t, ts, -1);
    __syncthreads();
  }
  const int owner = uni_i((int)ts->tt[TT_SSIDE]), slot = uni_i((int)ts->tt[TT_SSLOT]);
  if (tid == 0) {
    ts->n_leap = (int)ts->tt[TT_NLEAP]; ts->divergent = (int)ts->tt[TT_DIV]; ts->depth = (int)ts->tt[TT_DEPTH];
    ts->accept_stat = ts->tt[TT_METRO] / ts->tt[TT_NLEAP];
    ts->out_lp = ts->tt[TT_SLP]; ts->out_h = ts->tt[TT_SH];
    c.sc->total_leapfrogs += ts->n_leap;
    c.sc->n_divergent += ts->divergent;
    c.sc->spec_limit = (int)(((unsigned)c.sc->spec_limit << 8) | (unsigned)ts->tw_seq);   // doublings of the last four transitions, a byte each
  }
  __syncthreads();
  const bool warm = it < R->num_warmup;
  const bool save = (!warm || R->save_warmup) && side == 0;
  gdp row = as_g(R->draws) + ((size_t)chain * R->n_save_max + c.sc->saved) * R->row;
  if (save && tid == 0) {
    row[0] = ts->out_lp; row[1] = ts->accept_stat; row[2] = ts->eps; row[3] = ts->depth; row[4] = ts->n_leap;
    row[5] = ts->divergent; row[6] = ts->out_h;
  }
  const unsigned sQ = c.soff(V_QC);
  if (owner == 1 - side) {
    if (tid < 64) { double dummy; tw_wait(x, t, TWB_QC, 1, t.ittag, dummy); }
    __syncthreads();
  }
  {
    // The side that holds the sample publishes it in V_SCR1 (write-through; free until the next transition's merges, which
    // start after the rendezvous below) and keeps QC itself a plainly stored vector, as in the one-workgroup sampler: what
    // reads QC later (Welford update, the model pass of the next transition) uses plain loads.
    const bool remote = owner == 1 - side;
    const unsigned s_src = remote ? c.soff(V_SCR1) : c.soff(V_POOLQ + slot), s_pub = c.soff(V_SCR1);
    for (int i = tid; i < c.D; i += PT_THREADS) {
      const double v = remote ? bld_s(t.ost, 8u * i, s_src) : bld(c.st, 8u * i, s_src);
      bst(c.st, 8u * i, sQ, v);
      if (owner == side) bst_s(c.st, 8u * i, s_pub, v);
      if (save) row[POTUS_N_SAMPLER_COLS + i] = v;
    }
  }
  drain_vmem();
  __syncthreads();
  if (owner == side && tid < 64) tw_st(t, tid == 0 ? 16u * (unsigned)TWB_QC : PT_OOB, 1.0, t.ittag);
  if (tid == 0) {
    c.sc->lp_cur = ts->out_lp;
    if (!warm || R->save_warmup) c.sc->saved += 1;
  }
  if (warm) adapt_after_transition(c, (uint32_t)it, c.vec(V_QC));
  __syncthreads();
  if (tid == 0) c.sc->iter = it + 1;
  // Rendezvous: nothing of this transition is read from the other side's memory after this point, and the other side
  // must have reached the same point before this side's next transition overwrites what it may still be reading.
  drain_vmem();
  __syncthreads();
  if (tid < 64) {
    tw_st(t, tid == 0 ? 16u * (unsigned)(TWB_DONE + side) : PT_OOB, 1.0, t.ittag);
    tw_wait_ge(x, t, TWB_DONE + (1 - side), t.ittag);
  }
  __syncthreads();
}

// grid = 2 * chains; block b works for chain b % chains on side b / chains.  Both sides must be resident together.
__global__ __launch_bounds__(PT_THREADS) void k_run_twin(const DevModel *Mg, const RunParams *Rg, int n_iter, unsigned launch) {
  CMp M = (CMp)Mg;
  CRp R = (CRp)Rg;
  const int chain = blockIdx.x % R->chains, side = blockIdx.x / R->chains;
  Chain c = make_chain(M, R, chain, side);
  if (c.sc->status != 0) return;
  if (R->debug_drop_member == side + 1) return;    // test hook: a side that never shows up
  if (threadIdx.x == 0) cl_dead = 0;
  c.pst = model_setup_lds(M, c.lds);
  const Xch x = tw1_watch(R, chain + side * R->chains, launch);
  const Tw1Args ta{Mg, Rg, chain, side, launch};
  const int total = R->num_warmup + R->num_samples;
  for (int k = 0; k < n_iter; k++) {
    const int it = c.sc->iter;
    if (it >= total || uni_i(cl_dead)) break;
    cold_transition_begin(Mg, Rg, chain, (uint32_t)it, side);
    transition_tree_twin(c, x, ta, (uint32_t)it);
    if (uni_i(cl_dead)) break;
    cold_twin1_end(Mg, Rg, chain, side, launch, (uint32_t)it);
  }
  if (uni_i(cl_dead) && (c.tid & 63) == 0) c.sc->status = POTUS_ERR_WATCHDOG;   // every wave that is still alive
}

// ------------------------------------------------------------------------ cluster kernels (potus_cluster.hpp)
// grid = chains * K; block b works for chain b % chains as member b / chains.
// (twin mode: side s of a chain works in block chain + s * chains of state, scalars and exchange buffers)
__device__ __forceinline__ ClChain make_clchain(CMp M, CCp CL, CRp R, int chain, int m, unsigned launch, int side = 0) {
  ldp lds = (ldp)lds_dyn;
  ClChain c;
  const int Dpad = R->Dpad, K = CL->K;
  const int blk = chain + side * R->chains;
  double *state = R->state + (size_t)blk * V_COUNT * Dpad;
  c.M = M; c.CL = CL; c.part = (cip)(CL->part + m * CP_N); c.lds = lds; c.ts = (ltp)(lds + CL->lds_doubles);
  c.st = make_rsrc(state, (unsigned)V_COUNT * (unsigned)Dpad * 8u);
  c.sc = (gsc)(R->scal + (size_t)blk * K + m);
  c.key = RngKey{R->seed_lo, R->seed_hi, (uint32_t)(R->chain_id_offset + chain + 1)};
  c.perm = as_g(CL->perm);
  c.x.xb = make_rsrc(R->xbuf + (size_t)blk * (8 * K * CL->XW + 16), 4u * (unsigned)K * (unsigned)CL->XW * 16u + 16u);   // + the watchdog word
  c.x.epoch = 0; c.x.launch = launch; c.x.x1e = 0; c.x.K = K; c.x.m = m; c.x.XW = CL->XW; c.x.local = 0;
  c.D = M->D; c.Dpad = Dpad; c.tid = (int)threadIdx.x;
  c.e0 = c.part[CP_E0]; c.e1 = c.e0 + c.part[CP_NE];
  c.max_depth = R->max_depth; c.num_warmup = R->num_warmup; c.init_buffer = R->init_buffer; c.term_buffer = R->term_buffer;
  c.delta = R->delta; c.gamma = R->gamma; c.kappa = R->kappa; c.t0 = R->t0;
#ifdef POTUS_PROF
  c.prof = lds + CL->l_prof;
#endif
  return c;
}

// Parity hook on one cluster (grid = K): q, grad in Stan order; scratch = [2][Dpad] in internal order.
template <int CL_DW>
__global__ __launch_bounds__(PT_THREADS) void k_cl_logprob_grad(const DevModel *Mg, const ClModel *CLg, const RunParams *Rg, const double *q,
                                                                double *lp, double *grad, int n, double *scratch, unsigned launch) {
  CMp M = (CMp)Mg;
  CCp CL = (CCp)CLg;
  CRp R = (CRp)Rg;
  ClChain c = make_clchain(M, CL, R, 0, blockIdx.x, launch);
  c.cst = cl_setup_lds(M, CL, c.part, c.lds);
  const int D = M->D, Dpad = R->Dpad;
  const rsrc_t rs = make_rsrc(scratch, 2u * (unsigned)Dpad * 8u);
  for (int b = 0; b < n; b++) {
    for (int i = c.e0 + c.tid; i < c.e1; i += PT_THREADS) { const int si = c.perm[i]; bst_s(rs, 8u * i, 0, si >= 0 ? as_g(q)[(size_t)b * D + si] : 0.0); }
    cl_sync(c.x, c.red());
    ClPlainPolicy pol{rs, rs, 0u, (unsigned)Dpad * 8u, {0}};
    const double v = cl_pass<CL_DW>(M, CL, c.part, c.lds, c.cst, c.x, pol);
    drain_vmem();
    __syncthreads();
    for (int i = c.e0 + c.tid; i < c.e1; i += PT_THREADS) { const int si = c.perm[i]; if (si >= 0) as_g(grad)[(size_t)b * D + si] = bld(rs, 8u * i, (unsigned)Dpad * 8u); }
    if (blockIdx.x == 0 && c.tid == 0) lp[b] = v;
    cl_sync(c.x, c.red());
  }
}

template <int CL_DW>
__global__ __launch_bounds__(PT_THREADS) void k_cl_init(const DevModel *Mg, const ClModel *CLg, const RunParams *Rg, const double *q0, unsigned launch) {
  CMp M = (CMp)Mg;
  CCp CL = (CCp)CLg;
  CRp R = (CRp)Rg;
  const int chain = blockIdx.x % R->chains, r = blockIdx.x / R->chains, m = r % CL->K, side = r / CL->K;   // side 1: twin mode only
  ClChain c = make_clchain(M, CL, R, chain, m, launch, side);
  c.cst = cl_setup_lds(M, CL, c.part, c.lds);
  const int tid = c.tid;
  const unsigned sQ = c.soff(V_QC), sG = c.soff(V_GC);
  for (int i = c.e0 + tid; i < c.e1; i += PT_THREADS) { bst(c.st, 8u * i, c.soff(V_MINV), 1.0); bst(c.st, 8u * i, c.soff(V_WMEAN), 0.0); bst(c.st, 8u * i, c.soff(V_WM2), 0.0); }
  if (tid == 0) {
    gsc sc = c.sc;
    sc->nom_eps = R->stepsize; sc->mu = log(10.0 * R->stepsize); sc->s_bar = 0; sc->x_bar = 0; sc->ad_counter = 0;
    sc->wf_n = 0; sc->total_leapfrogs = 0; sc->iter = 0; sc->status = 0; sc->n_divergent = 0; sc->saved = 0; sc->leaves_run = 0; sc->spec_limit = 0x7f7f7f7f;
    sc->win_counter = 0; sc->win_size = R->window; sc->win_next = R->init_buffer + R->window - 1;
  }
  __syncthreads();
  bool ok = false;
  const double radius = R->init_radius;
  for (uint32_t attempt = 0; attempt < 100 && !ok; attempt++) {
    for (int i = c.e0 + tid; i < c.e1; i += PT_THREADS) {
      const int si = c.perm[i];                      // < 0: padding element, stays 0
      if (si >= 0) bst_s(c.st, 8u * i, sQ, q0 ? as_g(q0)[(size_t)chain * c.D + si] : radius * (2.0 * rng_uniform(c.key, PT_ITER_PRE, RNG_INITS, attempt, (uint32_t)si) - 1.0));
    }
    cl_sync(c.x, c.red());
    ClPlainPolicy pol{c.st, c.st, sQ, sG, {0}};
    const double lp = cl_pass<CL_DW>(M, CL, c.part, c.lds, c.cst, c.x, pol);
    drain_vmem();
    __syncthreads();
    double bad[1] = {0.0};
    for (int i = c.e0 + tid; i < c.e1; i += PT_THREADS) bad[0] += isfinite(bld(c.st, 8u * i, sG)) ? 0.0 : 1.0;
    cl_allreduce(bad, c.red(), c.x, tid);
    ok = isfinite(lp) && bad[0] == 0.0;
    if (tid == 0 && ok) c.sc->lp_cur = lp;
    __syncthreads();
    if (q0) break;
  }
  if (!ok) { if (tid == 0) c.sc->status = POTUS_ERR_INIT; return; }   // every member takes this branch together
  cl_init_stepsize<CL_DW>(c, PT_ITER_PRE);
  if (tid == 0 && R->num_warmup == 0) c.sc->nom_eps = exp(c.sc->x_bar);
}

// New sample -> chain position and draws array (Stan order); warmup adaptation.
template <int CL_DW>
__device__ __forceinline__ void cl_transition_end(ClChain &c, CRp R, int chain, int it) {
  ltp ts = c.ts;
  const int tid = c.tid;
  if (tid == 0) {
    ts->accept_stat = ts->sum_metro / (double)ts->n_leap;
    ts->out_lp = ts->q_lp[ts->sample_qid];
    ts->out_h = ts->q_h[ts->sample_qid];
    c.sc->total_leapfrogs += ts->n_leap;
    c.sc->n_divergent += ts->divergent;
  }
  __syncthreads();
  const bool warm = it < R->num_warmup;
  const bool save = !warm || R->save_warmup;
  gdp row = as_g(R->draws) + ((size_t)chain * R->n_save_max + c.sc->saved) * R->row;
  if (save && tid == 0 && c.x.m == 0) {
    row[0] = ts->out_lp; row[1] = ts->accept_stat; row[2] = ts->eps; row[3] = ts->depth; row[4] = ts->n_leap;
    row[5] = ts->divergent; row[6] = ts->out_h;
  }
  const unsigned s_src = c.soff(V_POOLQ + ts->sample_qid), sQ = c.soff(V_QC);
  for (int i = c.e0 + tid; i < c.e1; i += PT_THREADS) {
    const double v = bld_s(c.st, 8u * i, s_src);   // positions are read with sc1 loads (see potus_cluster.hpp)
    bst_s(c.st, 8u * i, sQ, v);
    const int si = c.perm[i];
    if (save && si >= 0) row[POTUS_N_SAMPLER_COLS + si] = v;
  }
  cl_sync(c.x, c.red());
  if (tid == 0) {
    c.sc->lp_cur = ts->out_lp;
    if (save) c.sc->saved += 1;
  }
  if (warm) cl_adapt_after_transition<CL_DW>(c, (uint32_t)it);
  __syncthreads();
  if (tid == 0) c.sc->iter = it + 1;
  __syncthreads();
}

// The once-per-transition parts are separate functions, as in the one-workgroup sampler (cold_transition_*): the
// kernel proper then holds one copy of the model pass (the leaf of the tree) instead of four, and the only state that
// crosses a call is the exchange counter.  Arguments of a non-inlined function arrive in vector registers; everything
// wave-uniform is rebuilt from them here.
template <int CL_DW>
__device__ __noinline__ unsigned cl_cold_transition_begin(const DevModel *Mg, const ClModel *CLg, const RunParams *Rg, int chain_, int m_, int side_,
                                                          unsigned launch_, unsigned epoch_, uint32_t iter_) {
  CMp M = (CMp)uni_ptr(Mg);
  CCp CL = (CCp)uni_ptr(CLg);
  CRp R = (CRp)uni_ptr(Rg);
  ClChain c = make_clchain(M, CL, R, (int)uni32((unsigned)chain_), (int)uni32((unsigned)m_), uni32(launch_), (int)uni32((unsigned)side_));
  c.x.epoch = uni32(epoch_); c.x.local = uni_i(cl_local);   // (what k_cl_run found out about this launch's placement)
  c.cst = cl_load_static(CL, c.part);
  CPROF_START(c);
  cl_transition_begin<CL_DW>(c, uni32(iter_));
  CPROF_MARK(c, PF_INITCOPY);
  return c.x.epoch;
}
template <int CL_DW>
__device__ __noinline__ unsigned cl_cold_transition_end(const DevModel *Mg, const ClModel *CLg, const RunParams *Rg, int chain_, int m_,
                                                        unsigned launch_, unsigned epoch_, uint32_t iter_) {
  CMp M = (CMp)uni_ptr(Mg);
  CCp CL = (CCp)uni_ptr(CLg);
  CRp R = (CRp)uni_ptr(Rg);
  const int chain = (int)uni32((unsigned)chain_);
  ClChain c = make_clchain(M, CL, R, chain, (int)uni32((unsigned)m_), uni32(launch_));
  c.x.epoch = uni32(epoch_); c.x.local = uni_i(cl_local);   // (what k_cl_run found out about this launch's placement)
  c.cst = cl_load_static(CL, c.part);
  CPROF_START(c);
  cl_transition_end<CL_DW>(c, R, chain, (int)uni32(iter_));
  CPROF_MARK(c, PF_SAVE);
  return c.x.epoch;
}

// ------------------------------------------------------------------------ twin mode (potus_cluster.hpp): the cold parts
// The combine of doubling d by the side that built its subtree (valid: the subtree completed without an internal U-turn
// or divergence; leaf: momentum slot of its last leaf).  Sets ts->tw_over when the trajectory is over.
template <int CL_DW>
__device__ __noinline__ unsigned cl_cold_twin_combine(const DevModel *Mg, const ClModel *CLg, const RunParams *Rg, int chain_, int m_, int side_,
                                                      unsigned launch_, unsigned epoch_, uint32_t iter_, int depth_, int valid_, int leaf_) {
  CMp M = (CMp)uni_ptr(Mg);
  CCp CL = (CCp)uni_ptr(CLg);
  CRp R = (CRp)uni_ptr(Rg);
  const int chain = (int)uni32((unsigned)chain_), side = (int)uni32((unsigned)side_), d = (int)uni32((unsigned)depth_);
  const int valid = (int)uni32((unsigned)valid_), leaf = (int)uni32((unsigned)leaf_);
  const uint32_t iter = uni32(iter_);
  ClChain c = make_clchain(M, CL, R, chain, (int)uni32((unsigned)m_), uni32(launch_), side);
  c.x.epoch = uni32(epoch_); c.x.local = uni_i(cl_local);   // (what k_cl_run found out about this launch's placement)
  const Twin t = make_twin(R, chain, side, c.x.launch, iter);
  ltp ts = c.ts;
  const int tid = c.tid;
  __syncthreads();
  if (tid == 0) c.sc->leaves_run += ts->n_leap;       // whether the subtree ends up in the trajectory or not
  // 1. the state this combine starts from: published by the combine of doubling d - 1 (or the end of the trajectory)
  if (uni_i(ts->tw_seq) < d) {
    if (tid < 64) tw_catch_up(c.x, t, ts, d);
    __syncthreads();
  }
  if (uni_i((int)ts->tt[TT_STOP])) {                 // the trajectory ended before this doubling: the subtree is dropped
    if (tid == 0) ts->tw_over = 1;
    __syncthreads();
    return c.x.epoch;
  }
  const int rho_side = uni_i((int)ts->tt[TT_RHOSIDE]), dirs = uni_i(ts->tw_dirs);
  int stop = 1;
  if (valid) {
    // 2. rho_top += rho_subtree and the three checks of transition() across the whole trajectory; the other end's momentum
    //    and (if the last combine was the other side's) rho_top come from the other side's state block, where they were
    //    stored write-through before that combine was published.  The subtree's last leaf becomes this side's end point.
    const bool other_moved = ((side ? ~dirs : dirs) & ((1 << d) - 1)) != 0;
    const rsrc_t rA = other_moved ? t.ost : c.st, rR = rho_side == 1 - side ? t.ost : c.st;
    const unsigned sM = c.soff(V_MINV), a_beg = c.soff(V_PF0 + (1 - side)), a_end = c.soff(V_PNEAR), s_rt = c.soff(V_RHOTOP);
    const unsigned b_beg = c.soff(V_POOLP + uni_i(ts->pend_beg[d])), b_end = c.soff(V_POOLP + leaf);
    const unsigned b_rho = d == 0 ? c.soff(V_POOLP + leaf) : c.soff(V_RHOLEV + d), s_pf = c.soff(V_PF0 + side);
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int base = cl_first(c); base < c.e1; base += CL_UNR * PT_THREADS) {
      double mi[CL_UNR], ab[CL_UNR], ae[CL_UNR], ar[CL_UNR], bb[CL_UNR], be[CL_UNR], br[CL_UNR];
#pragma unroll
      for (int k = 0; k < CL_UNR; k++) {
        const int i = base + k * PT_THREADS;
        const unsigned o = i < c.e1 ? 8u * i : PT_OOB;   // masked elements read zeros and add nothing
        mi[k] = bld(c.st, o, sM); ab[k] = bld_s(rA, o, a_beg); ae[k] = bld(c.st, o, a_end); ar[k] = bld_s(rR, o, s_rt);
        bb[k] = bld(c.st, o, b_beg); be[k] = bld(c.st, o, b_end); br[k] = bld(c.st, o, b_rho);
      }
#pragma unroll
      for (int k = 0; k < CL_UNR; k++) {
        const int i = base + k * PT_THREADS;
        const unsigned o = i < c.e1 ? 8u * i : PT_OOB;
        const double rs = ar[k] + br[k];
        bst_s(c.st, o, s_rt, rs);
        bst_s(c.st, o, s_pf, be[k]);
        const double sab = mi[k] * ab[k], sbe = mi[k] * be[k];
        v[0] += sab * rs;                 // p#_beg . rho
        v[1] += sbe * rs;                 // p#_end . rho
        const double e1 = ar[k] + bb[k];  // rho_old + p of the subtree's first leaf
        v[2] += sab * e1;
        v[3] += mi[k] * bb[k] * e1;
        const double e2 = br[k] + ae[k];  // rho_subtree + p of the old end
        v[4] += mi[k] * ae[k] * e2;
        v[5] += sbe * e2;
      }
    }
    cl_allreduce(v, c.red(), c.x, tid);   // also: every member's stores above are complete before anything is published
    const bool persist = v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0 && v[4] > 0 && v[5] > 0;
    stop = (!persist || d + 1 >= c.max_depth) ? 1 : 0;
  }
  __syncthreads();
  if (tid == 0) {
    // the subtree's leaves count whether it is valid or not (base_nuts: n_leapfrog_, sum_metro_prob, divergent_)
    ts->tt[TT_METRO] += ts->sum_metro; ts->tt[TT_NLEAP] += (double)ts->n_leap;
    if (ts->divergent) ts->tt[TT_DIV] = 1.0;
    if (valid) {
      const double lsw_top = ts->tt[TT_LSW], lsw_sub = ts->pend_lsw[d];
      const int prop = ts->pend_prop[d];
      bool accept;
      if (lsw_sub > lsw_top) accept = true;
      else accept = rng_uniform(c.key, iter, RNG_TOP_ACCEPT, 0, (uint32_t)d) < exp(lsw_sub - lsw_top);
      if (accept) {
        ts->tt[TT_SSIDE] = (double)side; ts->tt[TT_SSLOT] = (double)prop; ts->tt[TT_SLP] = ts->q_lp[prop]; ts->tt[TT_SH] = ts->q_h[prop];
        ts->tw_keep = prop;
      }
      ts->tt[TT_LSW] = d_lse(lsw_top, lsw_sub);
      ts->tt[TT_DEPTH] = (double)(d + 1);
      ts->tt[TT_RHOSIDE] = (double)side;
    }
    ts->tt[TT_STOP] = (double)stop;
    ts->tw_seq = d + 1; ts->tw_over = stop;
  }
  __syncthreads();
  // 3. publish (member 0): the state after combine d, and the STOP word if the trajectory is over
  if (c.x.m == 0 && tid < 64) {
    const double val = ts->tt[tid < TT_N ? tid : 0];
    tw_st(t, tid < TT_N ? 16u * (unsigned)(TWB_TOP + 16 * ((d + 1) & 1) + tid) : PT_OOB, val, t.ittag | (unsigned)(d + 2));
    tw_st(t, (stop && tid == 0) ? 16u * (unsigned)TWB_STOP : PT_OOB, (double)(d + 1), t.ittag);
  }
  __syncthreads();
  return c.x.epoch;
}

// New sample -> this side's chain position (from its own pool or from the other side's QC), draws array (side 0),
// adaptation (both sides, same inputs), rendezvous with the other side.
template <int CL_DW>
__device__ __noinline__ unsigned cl_cold_twin_end(const DevModel *Mg, const ClModel *CLg, const RunParams *Rg, int chain_, int m_, int side_,
                                                  unsigned launch_, unsigned epoch_, uint32_t iter_) {
  CMp M = (CMp)uni_ptr(Mg);
  CCp CL = (CCp)uni_ptr(CLg);
  CRp R = (CRp)uni_ptr(Rg);
  const int chain = (int)uni32((unsigned)chain_), side = (int)uni32((unsigned)side_), it = (int)uni32(iter_);
  ClChain c = make_clchain(M, CL, R, chain, (int)uni32((unsigned)m_), uni32(launch_), side);
  c.x.epoch = uni32(epoch_); c.x.local = uni_i(cl_local);   // (what k_cl_run found out about this launch's placement)
  c.cst = cl_load_static(CL, c.part);
  const Twin t = make_twin(R, chain, side, c.x.launch, (uint32_t)it);
  ltp ts = c.ts;
  const int tid = c.tid;
  CPROF_START(c);
  __syncthreads();
  if (!uni_i(ts->tw_over)) {                          // this side ran out of doublings of its own: wait for the end
    if (tid < 64) tw_catch_up(c.x, t, ts, -1);
    __syncthreads();
  }
  const int owner = uni_i((int)ts->tt[TT_SSIDE]), slot = uni_i((int)ts->tt[TT_SSLOT]);
  if (tid == 0) {
    ts->n_leap = (int)ts->tt[TT_NLEAP]; ts->divergent = (int)ts->tt[TT_DIV]; ts->depth = (int)ts->tt[TT_DEPTH];
    ts->accept_stat = ts->tt[TT_METRO] / ts->tt[TT_NLEAP];
    ts->out_lp = ts->tt[TT_SLP]; ts->out_h = ts->tt[TT_SH];
    c.sc->total_leapfrogs += ts->n_leap;
    c.sc->n_divergent += ts->divergent;
    c.sc->spec_limit = (int)(((unsigned)c.sc->spec_limit << 8) | (unsigned)ts->tw_seq);   // doublings of the last four transitions, a byte each
  }
  __syncthreads();
  const bool warm = it < R->num_warmup;
  const bool save = (!warm || R->save_warmup) && side == 0;
  gdp row = as_g(R->draws) + ((size_t)chain * R->n_save_max + c.sc->saved) * R->row;
  if (save && tid == 0 && c.x.m == 0) {
    row[0] = ts->out_lp; row[1] = ts->accept_stat; row[2] = ts->eps; row[3] = ts->depth; row[4] = ts->n_leap;
    row[5] = ts->divergent; row[6] = ts->out_h;
  }
  const unsigned sQ = c.soff(V_QC);
  if (owner == 1 - side) {
    if (tid < 64) { double dummy; tw_wait(c.x, t, TWB_QC, 1, t.ittag, dummy); }
    __syncthreads();
  }
  {
    const rsrc_t rsrc = owner == 1 - side ? t.ost : c.st;
    const unsigned s_src = owner == 1 - side ? sQ : c.soff(V_POOLQ + slot);
    for (int i = c.e0 + tid; i < c.e1; i += PT_THREADS) {
      const double v = bld_s(rsrc, 8u * i, s_src);
      bst_s(c.st, 8u * i, sQ, v);
      const int si = c.perm[i];
      if (save && si >= 0) row[POTUS_N_SAMPLER_COLS + si] = v;
    }
  }
  cl_sync(c.x, c.red());
  if (owner == side && c.x.m == 0 && tid < 64) tw_st(t, tid == 0 ? 16u * (unsigned)TWB_QC : PT_OOB, 1.0, t.ittag);
  if (tid == 0) {
    c.sc->lp_cur = ts->out_lp;
    if (!warm || R->save_warmup) c.sc->saved += 1;
  }
  if (warm) cl_adapt_after_transition<CL_DW>(c, (uint32_t)it);
  __syncthreads();
  if (tid == 0) c.sc->iter = it + 1;
  // Rendezvous: nothing of this transition is read from the other side's memory after this point, and the other side
  // must have reached the same point before this side's next transition overwrites what it may still be reading.
  cl_sync(c.x, c.red());
  if (c.x.m == 0 && tid < 64) tw_st(t, tid == 0 ? 16u * (unsigned)(TWB_DONE + side) : PT_OOB, 1.0, t.ittag);
  if (tid < 64) tw_wait_ge(c.x, t, TWB_DONE + (1 - side), t.ittag);
  __syncthreads();
  CPROF_MARK(c, PF_SAVE);
  return c.x.epoch;
}

template <int CL_DW, bool TWIN>
__global__ __launch_bounds__(PT_THREADS) void k_cl_run(const DevModel *Mg, const ClModel *CLg, const RunParams *Rg, int n_iter, unsigned launch) {
  CMp M = (CMp)Mg;
  CCp CL = (CCp)CLg;
  CRp R = (CRp)Rg;
  // Which cluster a block works for.  Blocks are dealt to the eight XCDs round-robin: whenever the launch has a multiple of eight clusters, the blocks of one residue
  // b % 8 -- one XCD -- form whole clusters (cluster x + 8 q, member (b / 8) % K), so that the members of a cluster share an L2 and publish their exchange words with plain
  // stores (cl_find_local checks, per launch).  With a multiple of eight chains this is the mapping of rounds 1-6 (chain = b % chains, both clusters of a chain on one XCD);
  // with four chains x two clusters the clusters of a chain sit on XCDs c and c + 4.  Otherwise: chain = b % chains, members wherever they fall, write-through stores.
  int chain, m, side;
  {
    const int C = R->chains, K = CL->K, b = (int)blockIdx.x, sides = TWIN ? 2 : 1, ncl = C * sides;
    if (ncl % 8 == 0) {
      const int x = b & 7, i = b >> 3, q = i / K, j = x + 8 * q;                      // cluster j of ncl
      m = i % K;
      if (C % 8 == 0) { chain = x + 8 * (q / sides); side = q % sides; }
      else { chain = j % C; side = j / C; }
    } else {
      const int r = b / C;
      chain = b % C; m = TWIN ? r % K : r; side = TWIN ? r / K : 0;
    }
  }
  ClChain c = make_clchain(M, CL, R, chain, m, launch, side);
  if (c.sc->status != 0) return;
  if (R->debug_drop_member == m + 1) return;       // test hook: a member that never shows up
  c.cst = cl_setup_lds(M, CL, c.part, c.lds);
  cl_find_local(c.x, c.red());                      // (one all-reduce per launch: exchange words go out as plain stores when the cluster sits on one XCD)
  if (R->debug_drop_member == -1) {                 // test hook: the write-through path whatever the placement
    if (c.tid == 0) cl_local = 0;
    __syncthreads();
    c.x.local = 0;
  }
  if (c.tid == 0) c.sc->xcd_local = c.x.local;
  const ClTwinArgs ta{Mg, CLg, Rg, chain, m, side, launch};
  const int total = R->num_warmup + R->num_samples;
  for (int k = 0; k < n_iter; k++) {
    const int it = c.sc->iter;
    if (it >= total || uni_i(cl_dead)) break;
    c.x.epoch = uni32(cl_cold_transition_begin<CL_DW>(Mg, CLg, Rg, chain, m, side, launch, c.x.epoch, (uint32_t)it));
    cl_transition_tree<CL_DW, TWIN>(c, (uint32_t)it, ta);
    if (uni_i(cl_dead)) break;
    if (TWIN) c.x.epoch = uni32(cl_cold_twin_end<CL_DW>(Mg, CLg, Rg, chain, m, side, launch, c.x.epoch, (uint32_t)it));
    else c.x.epoch = uni32(cl_cold_transition_end<CL_DW>(Mg, CLg, Rg, chain, m, launch, c.x.epoch, (uint32_t)it));
  }
  if (uni_i(cl_dead) && (c.tid & 63) == 0) c.sc->status = POTUS_ERR_WATCHDOG;   // every wave that is still alive
#ifdef POTUS_PROF
  if (R->prof) for (int i = c.tid; i < PT_NPROF; i += PT_THREADS) as_g(R->prof)[((size_t)(chain + side * R->chains) * CL->K + m) * PT_NPROF + i] += c.prof[i];
#endif
}

// ------------------------------------------------------------------------ gradient of a dense-metric round (potus_dense.hpp)
// one workgroup per chain: the position and the slot receiving the gradient come from the chain's round descriptor
__global__ __launch_bounds__(PT_THREADS) void k_dn_grad1(const DevModel *Mg, const DnParams P) {
  const int chain = blockIdx.x;
  const DnRound &rd = P.rd[chain];
  if (!rd.active) return;
  CMp M = (CMp)Mg;
  ldp lds = (ldp)lds_dyn;
  const PassStatic pst = model_setup_lds(M, lds);
  const int D = M->D;
  PlainPolicy pol{make_rsrc(dn_vec(P, chain, rd.qin), 8u * D), make_rsrc(dn_vec(P, chain, rd.gout), 8u * D), 0u, 0u, {0}};
  const double v = model_pass(M, lds, pst, pol);
  if (threadIdx.x == 0) P.lpbuf[chain] = v;
}
// one cluster per chain (models beyond one workgroup): the position goes into the cluster's internal element order
// (scratch vectors of the chain's state block), the gradient comes back in Stan's order
template <int CL_DW>
__global__ __launch_bounds__(PT_THREADS) void k_dn_gradK(const DevModel *Mg, const ClModel *CLg, const RunParams *Rg, const DnParams P, unsigned launch) {
  CMp M = (CMp)Mg;
  CCp CL = (CCp)CLg;
  CRp R = (CRp)Rg;
  const int chain = blockIdx.x % R->chains, m = blockIdx.x / R->chains;
  const DnRound &rd = P.rd[chain];
  if (!rd.active) return;                            // the whole cluster
  ClChain c = make_clchain(M, CL, R, chain, m, launch);
  c.cst = cl_setup_lds(M, CL, c.part, c.lds);
  const double *q = dn_vec(P, chain, rd.qin);
  double *grad = dn_vec(P, chain, rd.gout);
  const unsigned sQ = c.soff(V_QA1), sG = c.soff(V_GC);
  for (int i = c.e0 + c.tid; i < c.e1; i += PT_THREADS) { const int si = c.perm[i]; bst_s(c.st, 8u * i, sQ, si >= 0 ? as_g(q)[si] : 0.0); }
  cl_sync(c.x, c.red());
  ClPlainPolicy pol{c.st, c.st, sQ, sG, {0}};
  const double v = cl_pass<CL_DW>(M, CL, c.part, c.lds, c.cst, c.x, pol);
  drain_vmem();
  __syncthreads();
  for (int i = c.e0 + c.tid; i < c.e1; i += PT_THREADS) { const int si = c.perm[i]; if (si >= 0) as_g(grad)[si] = bld(c.st, 8u * i, sG); }
  if (m == 0 && c.tid == 0) P.lpbuf[chain] = v;
}
// the point found by k_init / k_cl_init (diagonal state block) -> the dense state block, Stan's order
__global__ void k_dn_import_q(const RunParams *Rg, const DnParams P, const int *perm, int Dint) {
  const int chain = blockIdx.y;
  const double *src = Rg->state + ((size_t)chain * V_COUNT + V_QC) * Rg->Dpad;
  double *dst = dn_vec(P, chain, DV_QC);
  const int n = perm ? Dint : P.D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int si = perm ? perm[i] : i;
    if (si >= 0) dst[si] = src[i];
  }
}

// write_array (stan:70-113 transformed parameters, stan:134-140 generated quantities) for saved
// draws; one 256-thread workgroup per draw builds the full CmdStan row in scratch and copies
// the requested column range.  out layout: [iter][chain][col_end - col_begin].
struct WAParams {
  const double *draws; // [chains][n_save_max][row]
  int chains, n_save_max, n_saved, row, ncols, col_begin, col_end;
  double *scratch;     // [gridDim.x][ncols]
  double *out;         // [n_saved * chains][out_stride], the first col_end - col_begin entries of a row are written
  double sigma_ns, sigma_nn;
  int out_stride;
};
__global__ __launch_bounds__(256) void k_write_array(const DevModel *Mg, WAParams W) {
  const DevModel M = *Mg;
  __shared__ double s_bT[64], s_pb[64], s_misc[4];
  const int tid = threadIdx.x, S = M.S, T = M.T, D = M.D;
  double *row = W.scratch + (size_t)blockIdx.x * W.ncols;
  const int o_par = POTUS_N_SAMPLER_COLS;
  const int o_mub = o_par + D, o_muc = o_mub + S * T;
  const int o_mum = o_muc + M.P, o_mupop = o_mum + (M.full ? M.M : 0), o_eb = o_mupop + (M.full ? M.Pop : 0);
  const int o_pb = o_eb + (M.full ? T : 0), o_nat = o_pb + S, o_natpb = o_nat + T, o_srho = o_natpb + 1;
  const int o_etas = o_srho + (M.full ? 1 : 0), o_etan = o_etas + M.Ns, o_gq = o_etan + M.Nn;
  for (int d = blockIdx.x; d < W.n_saved * W.chains; d += gridDim.x) {
    const int iter = d / W.chains, chain = d % W.chains;
    const double *src = W.draws + ((size_t)chain * W.n_save_max + iter) * W.row;
    const double *q = src + POTUS_N_SAMPLER_COLS;
    for (int i = tid; i < POTUS_N_SAMPLER_COLS + D; i += 256) row[i] = src[i];
    __syncthreads();
    double *Ct = row + o_gq; // suffix sums, staged where predicted_score will go
    if (tid < S) {
      double run = 0.0;
      Ct[tid + S * (T - 1)] = 0.0;
      for (int t = T - 2; t >= 0; t--) { run += q[M.o_Z + tid + S * t]; Ct[tid + S * t] = run; }
      double bT = M.mat[M.m_prior + tid], pb = 0.0;
      for (int k = 0; k <= tid; k++) { bT += M.mat[M.m_LT + tid * S + k] * q[M.o_zT + k]; pb += M.mat[M.m_LB + tid * S + k] * q[M.o_zb + k]; }
      s_bT[tid] = bT; s_pb[tid] = pb; row[o_pb + tid] = pb;
    }
    if (tid == 64 && M.full) {
      const double mue = 0.02 * q[M.o_mue], rho = d_inv_logit(q[M.o_rho]), srho = sqrt(1.0 - rho * rho) * M.sigma_e;
      row[o_par + M.o_mue] = mue; row[o_par + M.o_rho] = rho; row[o_srho] = srho;
      double e = q[M.o_ze] * M.sigma_e;
      row[o_eb] = e;
      for (int t = 1; t < T; t++) { e = mue + rho * (e - mue) + q[M.o_ze + t] * srho; row[o_eb + t] = e; }
    }
    for (int i = tid; i < M.P; i += 256) row[o_muc + i] = q[M.o_c + i] * M.sigma_c;
    if (M.full) {
      for (int i = tid; i < M.M; i += 256) row[o_mum + i] = q[M.o_m + i] * M.sigma_m;
      for (int i = tid; i < M.Pop; i += 256) row[o_mupop + i] = q[M.o_pop + i] * M.sigma_pop;
    }
    __syncthreads();
    for (int idx = tid; idx < S * T; idx += 256) {
      const int s = idx % S, t = idx / S;
      double a = s_bT[s];
      const double *Lrow = M.mat + s * M.SP, *Cc = Ct + S * t;
      for (int k = 0; k <= s; k++) a += Lrow[k] * Cc[k];
      row[o_mub + idx] = a;
    }
    if (tid == 0) { double a = 0.0; for (int s = 0; s < S; s++) a += s_pb[s] * M.mat[M.m_w + s]; s_misc[0] = a; row[o_natpb] = a; }
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
      double a = 0.0;
      for (int s = 0; s < S; s++) a += row[o_mub + s + S * t] * M.mat[M.m_w + s];
      row[o_nat + t] = a;
    }
    __syncthreads();
    for (int i = tid; i < M.Npoll; i += 256) {
      const int s = M.pi[i], t = M.pi[M.Npad + i], qi = M.pi[5 * M.Npad + i];
      const bool nat = s == S;
      double eta = (nat ? row[o_nat + t] : row[o_mub + s + S * t]) + row[o_muc + M.pi[2 * M.Npad + i]];
      if (M.full) eta += row[o_mum + M.pi[3 * M.Npad + i]] + row[o_mupop + M.pi[4 * M.Npad + i]] + M.pd[2 * M.Npad + i] * row[o_eb + t];
      eta += q[qi] * M.pd[3 * M.Npad + i] + (nat ? s_misc[0] : s_pb[s]);
      if (nat) row[o_etan + (qi - M.o_nn)] = eta; else row[o_etas + (qi - M.o_ns)] = eta;
    }
    __syncthreads();
    for (int idx = tid; idx < S * T; idx += 256) { // predicted_score[t,s] = inv_logit(mu_b[s,t]) (stan:135-139)
      const int t = idx % T, s = idx / T;
      row[o_gq + idx] = d_inv_logit(row[o_mub + s + S * t]);
    }
    __syncthreads();
    const int nsel = W.col_end - W.col_begin;
    double *dst = W.out + (size_t)d * W.out_stride;
    for (int i = tid; i < nsel; i += 256) dst[i] = row[W.col_begin + i];
    __syncthreads();
  }
}

// ======================================================================== host
namespace {

thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(POTUS_ERR_DEVICE, "%s failed: %s", #x, hipGetErrorString(e_)); } while (0)

struct Layout { // column blocks of one output row after the 7 sampler columns
  int D, ncols;
  int o_zT, o_Z, o_c, o_m, o_pop, o_mue, o_rho, o_ze, o_nn, o_ns, o_zb;
};
Layout make_layout(const potus_data *d) {
  Layout L{};
  const bool full = d->variant == POTUS_VARIANT_FULL;
  int o = 0;
  L.o_zT = o; o += d->S;
  L.o_Z = o; o += d->S * d->T;
  L.o_c = o; o += d->P;
  if (full) { L.o_m = o; o += d->M; L.o_pop = o; o += d->Pop; L.o_mue = o; o += 1; L.o_rho = o; o += 1; L.o_ze = o; o += d->T; }
  else L.o_m = L.o_pop = L.o_mue = L.o_rho = L.o_ze = -1;
  L.o_nn = o; o += d->N_national_polls;
  L.o_ns = o; o += d->N_state_polls;
  L.o_zb = o; o += d->S;
  L.D = o;
  const int tp = d->S * d->T + d->P + (full ? d->M + d->Pop + d->T : 0) + d->S + d->T + 1 + (full ? 1 : 0)
                 + d->N_state_polls + d->N_national_polls;
  L.ncols = POTUS_N_SAMPLER_COLS + L.D + tp + d->T * d->S;
  return L;
}

int validate(const potus_data *d) {
  if (!d) return fail(POTUS_ERR_ARG, "null data");
  if (d->variant != POTUS_VARIANT_FULL && d->variant != POTUS_VARIANT_NO_MODE) return fail(POTUS_ERR_ARG, "unknown variant %d", d->variant);
  const bool full = d->variant == POTUS_VARIANT_FULL;
  if (d->S < 1 || d->P < 1 || d->N_state_polls < 0 || d->N_national_polls < 0)
    return fail(POTUS_ERR_ARG, "bad sizes: S = %d, P = %d, N_state_polls = %d, N_national_polls = %d", d->S, d->P, d->N_state_polls, d->N_national_polls);
  if (d->T < 2) return fail(POTUS_ERR_ARG, "T = %d: the model is a random walk over days, at least two are needed", d->T);
  if (full && (d->M < 1 || d->Pop < 1)) return fail(POTUS_ERR_ARG, "bad sizes M/Pop");
  auto range = [&](const int32_t *v, int n, int lo, int hi, const char *name) {
    if (n > 0 && !v) return fail(POTUS_ERR_ARG, "%s is null", name);
    for (int i = 0; i < n; i++) if (v[i] < lo || v[i] > hi) return fail(POTUS_ERR_ARG, "%s[%d] = %d outside [%d,%d]", name, i + 1, v[i], lo, hi);
    return 0;
  };
  int rc;
  // declared bounds stan:9-17; state = S+1 is declared legal but indexes out of range at stan:97
  if ((rc = range(d->state, d->N_state_polls, 1, d->S, "state"))) return rc;
  if ((rc = range(d->day_state, d->N_state_polls, 1, d->T, "day_state"))) return rc;
  if ((rc = range(d->day_national, d->N_national_polls, 1, d->T, "day_national"))) return rc;
  if ((rc = range(d->poll_state, d->N_state_polls, 1, d->P, "poll_state"))) return rc;
  if ((rc = range(d->poll_national, d->N_national_polls, 1, d->P, "poll_national"))) return rc;
  if (full) {
    if ((rc = range(d->poll_mode_state, d->N_state_polls, 1, d->M, "poll_mode_state"))) return rc;
    if ((rc = range(d->poll_mode_national, d->N_national_polls, 1, d->M, "poll_mode_national"))) return rc;
    if ((rc = range(d->poll_pop_state, d->N_state_polls, 1, d->Pop, "poll_pop_state"))) return rc;
    if ((rc = range(d->poll_pop_national, d->N_national_polls, 1, d->Pop, "poll_pop_national"))) return rc;
    if ((d->N_state_polls && !d->unadjusted_state) || (d->N_national_polls && !d->unadjusted_national)) return fail(POTUS_ERR_ARG, "unadjusted_* is null");
    for (int i = 0; i < d->N_state_polls; i++) if (!(d->unadjusted_state[i] >= 0 && d->unadjusted_state[i] <= 1)) return fail(POTUS_ERR_ARG, "unadjusted_state outside [0,1]");
    for (int i = 0; i < d->N_national_polls; i++) if (!(d->unadjusted_national[i] >= 0 && d->unadjusted_national[i] <= 1)) return fail(POTUS_ERR_ARG, "unadjusted_national outside [0,1]");
  }
  if ((d->N_state_polls && (!d->n_democrat_state || !d->n_two_share_state)) || (d->N_national_polls && (!d->n_democrat_national || !d->n_two_share_national)))
    return fail(POTUS_ERR_ARG, "poll counts are null");
  for (int i = 0; i < d->N_state_polls; i++) if (d->n_democrat_state[i] < 0 || d->n_democrat_state[i] > d->n_two_share_state[i]) return fail(POTUS_ERR_ARG, "n_democrat_state outside [0,N]");
  for (int i = 0; i < d->N_national_polls; i++) if (d->n_democrat_national[i] < 0 || d->n_democrat_national[i] > d->n_two_share_national[i]) return fail(POTUS_ERR_ARG, "n_democrat_national outside [0,N]");
  if (!d->mu_b_prior || !d->state_weights || !d->state_covariance_0) return fail(POTUS_ERR_ARG, "null prior/weights/covariance");
  const int S = d->S;
  for (int i = 0; i < S; i++) for (int j = 0; j < i; j++) {
    const double a = d->state_covariance_0[i + (size_t)j * S], b = d->state_covariance_0[j + (size_t)i * S];
    if (std::fabs(a - b) > 1e-8 * std::max(1.0, std::max(std::fabs(a), std::fabs(b)))) return fail(POTUS_ERR_ARG, "state_covariance_0 is not symmetric");
  }
  return 0;
}

// FNV-1a over everything the posterior depends on (sizes, index vectors, counts, flags, prior, weights, covariance, scales)
unsigned long long hash_data(const potus_data *d) {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; p && i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
  const bool full = d->variant == POTUS_VARIANT_FULL;
  const int32_t dims[8] = {d->N_national_polls, d->N_state_polls, d->T, d->S, d->P, full ? d->M : 0, full ? d->Pop : 0, d->variant};
  mix(dims, sizeof dims);
  const size_t ns = (size_t)d->N_state_polls * 4, nn = (size_t)d->N_national_polls * 4;
  mix(d->state, ns); mix(d->day_state, ns); mix(d->day_national, nn); mix(d->poll_state, ns); mix(d->poll_national, nn);
  if (full) { mix(d->poll_mode_state, ns); mix(d->poll_mode_national, nn); mix(d->poll_pop_state, ns); mix(d->poll_pop_national, nn);
              mix(d->unadjusted_state, ns * 2); mix(d->unadjusted_national, nn * 2); }
  mix(d->n_democrat_national, nn); mix(d->n_two_share_national, nn); mix(d->n_democrat_state, ns); mix(d->n_two_share_state, ns);
  mix(d->mu_b_prior, (size_t)d->S * 8); mix(d->state_weights, (size_t)d->S * 8); mix(d->state_covariance_0, (size_t)d->S * d->S * 8);
  const double sc[9] = {d->sigma_c, full ? d->sigma_m : 0, full ? d->sigma_pop : 0, d->sigma_measure_noise_national, d->sigma_measure_noise_state,
                        full ? d->sigma_e_bias : 0, d->random_walk_scale, d->mu_b_T_scale, d->polling_bias_scale};
  mix(sc, sizeof sc);
  return h;
}

// lower Cholesky factor, column-major (cholesky_decompose, stan:52-54)
bool chol(const std::vector<double> &A, std::vector<double> &L, int n) {
  L.assign((size_t)n * n, 0.0);
  for (int j = 0; j < n; j++) {
    double s = A[j + (size_t)j * n];
    for (int k = 0; k < j; k++) s -= L[j + (size_t)k * n] * L[j + (size_t)k * n];
    if (!(s > 0)) return false;
    const double ljj = std::sqrt(s);
    L[j + (size_t)j * n] = ljj;
    for (int i = j + 1; i < n; i++) {
      double t = A[i + (size_t)j * n];
      for (int k = 0; k < j; k++) t -= L[i + (size_t)k * n] * L[j + (size_t)k * n];
      L[i + (size_t)j * n] = t / ljj;
    }
  }
  return true;
}

struct Sampler {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  DevModel M{};
  DevModel *dM = nullptr; // device copy read by the kernels through scalar loads
  RunParams *dR = nullptr;
  RunParams R{};
  potus_opts opts{};
  Layout L{};
  double sigma_ns = 0, sigma_nn = 0;
  std::vector<void *> allocs;
  size_t lds_bytes = 0;
  std::string k1_unsupported;   // why the one-workgroup kernels cannot run this model (empty: they can); cluster mode may still
  bool inited = false;
  double last_ms = 0;
  long long last_leapfrogs = 0;
  double warm_ms = 0, samp_ms = 0;
  std::vector<double> LB, LT, LW; // column-major host copies (transformed data)
  // cluster mode (K > 1)
  int K = 1, cl_dw = 8;
  int twin = 0;             // two clusters per chain (one per end of the trajectory): state, scalars and exchange buffers hold 2 x chains blocks
  int sides() const { return twin ? 2 : 1; }
  bool coop() const { return K > 1 || twin; }   // workgroups that wait for each other: the whole grid must be resident
  unsigned launch_id = 0;   // tags the exchange words of each launch
  ClModel CL{};
  ClModel *dCL = nullptr;
  size_t cl_lds_bytes = 0;
  std::vector<int> h_ps, h_pt, h_pp, h_pm, h_ppop, h_pq, h_dayptr, h_perm;   // day-sorted polls (host copies)
  std::vector<double> h_w;    // state_weights
  std::vector<double> h_pu;
  unsigned long long data_hash = 0;   // FNV-1a over the data block: handles that pool their draws must hold the same posterior
  // dense metric (potus_dense.hpp)
  bool dense = false;
  DnParams dn{};
  int *h_active = nullptr;                 // pinned host copy of the per-chain activity flags
  int dn_win_counter = 0, dn_win_next = 0, dn_win_size = 0, dn_wf_n = 0;   // host mirror of the warm-up window schedule
  hipEvent_t mv0 = nullptr, mv1 = nullptr;
  static constexpr int DN_AHEAD = 4;       // leaf rounds the host keeps queued beyond the one whose activity flags it has seen
  hipEvent_t rv0[DN_AHEAD] = {}, rv1[DN_AHEAD] = {}, rdone[DN_AHEAD] = {};   // per queued round: around its matrix pass, after its flag copy
  bool dn_pending = false;                 // pooled_metric = 2: a window end whose moments wait for the host (potus_dense_pool_window / _finish)
  int dn_pending_n = 0; unsigned dn_pending_iter = 0; double dn_pending_cov_ms = 0;
  int rv_launches[DN_AHEAD] = {1, 1, 1, 1};   // ... and the launches of that pass (pooled metric: one per DNP_RMAX right-hand sides)
  double mv_ms = 0;                        // time spent in the matrix passes (k_dn_symv + finish, events), their number and the bytes they loaded
  int mv_launches_pending = 0;             // matrix passes between the event pair of the last timed dense_matvec (the first pass of a transition: two)
  long long mv_calls = 0, mv_bytes = 0, dn_rounds = 0, dn_pass_bytes[2] = {0, 0};   // bytes per pass at DN_RB rows per workgroup / at the handle's DnParams::rb
  double we_cov_ms = 0, we_chol_ms = 0, we_eps_ms = 0;   // window ends: covariance, factorisation, init_stepsize (host clock around synchronised sections)
  int we_count = 0;
};

std::mutex g_mu;
std::vector<Sampler *> g_handles;
// Cluster kernels need all their workgroups co-resident (they wait for each other): whatever launches work on a device --
// sampler runs, the parity hook, write_array, the posterior summaries, the CSV writer's row kernels -- holds that device's
// mutex from launch to completion, so host threads driving one handle per GPU run side by side while two calls on one GPU take
// turns.  potus_run_many locks the devices of its handles in ascending order.
constexpr int POTUS_MAX_DEVICES = 64;
std::mutex g_device_mu[POTUS_MAX_DEVICES];
struct DeviceLocks {
  std::vector<std::unique_lock<std::mutex>> held;
  explicit DeviceLocks(std::vector<int> devs) {
    std::sort(devs.begin(), devs.end());
    devs.erase(std::unique(devs.begin(), devs.end()), devs.end());
    for (int d : devs) held.emplace_back(g_device_mu[std::min(std::max(d, 0), POTUS_MAX_DEVICES - 1)]);
  }
  explicit DeviceLocks(int dev) : DeviceLocks(std::vector<int>{dev}) {}
};
// the caller's HIP device is put back on every return path (the library works on the handles' devices)
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

Sampler *get(int h) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (h < 0 || h >= (int)g_handles.size()) return nullptr;
  return g_handles[h];
}

// One owner for every temporary device buffer of a call: freed on every return path.
struct DevBufs {
  std::vector<void *> v;
  ~DevBufs() { for (void *p : v) (void)hipFree(p); }
  template <class T> hipError_t alloc(T **p, size_t bytes) {
    void *q = nullptr;
    const hipError_t e = hipMalloc(&q, std::max<size_t>(bytes, 8));
    if (e == hipSuccess) v.push_back(q);
    *p = static_cast<T *>(q);
    return e;
  }
};

template <class T>
int upload(Sampler *s, const std::vector<T> &v, const T **dst) {
  void *p = nullptr;
  const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  HIP_TRY(hipMalloc(&p, bytes));
  s->allocs.push_back(p);
  if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *dst = static_cast<const T *>(p);
  return 0;
}

int build_model(Sampler *sp, const potus_data *d) {
  const bool full = d->variant == POTUS_VARIANT_FULL;
  const int S = d->S, T = d->T, Ns = d->N_state_polls, Nn = d->N_national_polls;
  DevModel &M = sp->M;
  const Layout &L = sp->L;
  if (S + 1 > 64) return fail(POTUS_ERR_UNSUPPORTED, "S = %d: this kernel maps states to the 64 lanes of a wave (S <= 63)", S);
  auto k1_fail = [&](const char *fmt, auto... args) { char buf[256]; std::snprintf(buf, sizeof buf, fmt, args...); if (sp->k1_unsupported.empty()) sp->k1_unsupported = buf; };
  if (T > PT_NW * PT_CH) k1_fail("T = %d: one workgroup per chain handles T <= %d days", T, PT_NW * PT_CH);
  M.S = S; M.T = T; M.P = d->P; M.M = full ? d->M : 0; M.Pop = full ? d->Pop : 0; M.Ns = Ns; M.Nn = Nn; M.Npoll = Ns + Nn;
  M.D = L.D; M.full = full;
  M.SE = S + 1; M.SP = S | 1; M.TP = T | 1;
  M.o_zT = L.o_zT; M.o_Z = L.o_Z; M.o_c = L.o_c; M.o_m = L.o_m; M.o_pop = L.o_pop; M.o_mue = L.o_mue; M.o_rho = L.o_rho;
  M.o_ze = L.o_ze; M.o_nn = L.o_nn; M.o_ns = L.o_ns; M.o_zb = L.o_zb; M.nmid = L.o_nn - L.o_c;
  M.sigma_c = d->sigma_c; M.sigma_m = d->sigma_m; M.sigma_pop = d->sigma_pop; M.sigma_e = d->sigma_e_bias;
  sp->sigma_ns = d->sigma_measure_noise_state; sp->sigma_nn = d->sigma_measure_noise_national;
  M.sigma_ns = d->sigma_measure_noise_state; M.sigma_nn = d->sigma_measure_noise_national;

  // transformed data (stan:42-55)
  std::vector<double> w(d->state_weights, d->state_weights + S), cov(d->state_covariance_0, d->state_covariance_0 + (size_t)S * S);
  double nsd2 = 0;
  for (int i = 0; i < S; i++) for (int j = 0; j < S; j++) nsd2 += w[i] * cov[i + (size_t)j * S] * w[j];
  const double nsd = std::sqrt(nsd2);
  auto scaled_chol = [&](double scale, std::vector<double> &Lout) {
    std::vector<double> A(cov);
    const double f = (scale / nsd) * (scale / nsd);
    for (auto &x : A) x *= f;
    return chol(A, Lout, S);
  };
  if (!scaled_chol(d->polling_bias_scale, sp->LB) || !scaled_chol(d->mu_b_T_scale, sp->LT) || !scaled_chol(d->random_walk_scale, sp->LW))
    return fail(POTUS_ERR_ARG, "state_covariance_0 is not positive definite");
  auto at = [&](const std::vector<double> &Lm, int i, int j) { return Lm[i + (size_t)j * S]; };

  std::vector<double> Lw_ext((size_t)M.SE * M.SP, 0.0), LT_t((size_t)S * S), LB_t((size_t)S * S), LTr((size_t)S * S), LBr((size_t)S * S);
  for (int s = 0; s < S; s++) for (int k = 0; k < S; k++) {
    Lw_ext[(size_t)s * M.SP + k] = at(sp->LW, s, k);
    LT_t[(size_t)k * S + s] = at(sp->LT, s, k); LTr[(size_t)s * S + k] = at(sp->LT, s, k);
    LB_t[(size_t)k * S + s] = at(sp->LB, s, k); LBr[(size_t)s * S + k] = at(sp->LB, s, k);
  }
  for (int k = 0; k < S; k++) { double v = 0; for (int s = k; s < S; s++) v += at(sp->LW, s, k) * w[s]; Lw_ext[(size_t)S * M.SP + k] = v; }

  // polls merged and sorted by day (then state); national polls carry the pseudo-state S
  struct Poll { int s, t, p, m, pop, qidx; double y, n, unadj, sig; };
  std::vector<Poll> polls;
  for (int i = 0; i < Ns; i++)
    polls.push_back({d->state[i] - 1, d->day_state[i] - 1, d->poll_state[i] - 1, full ? d->poll_mode_state[i] - 1 : 0,
                     full ? d->poll_pop_state[i] - 1 : 0, L.o_ns + i, (double)d->n_democrat_state[i], (double)d->n_two_share_state[i],
                     full ? d->unadjusted_state[i] : 0.0, d->sigma_measure_noise_state});
  for (int j = 0; j < Nn; j++)
    polls.push_back({S, d->day_national[j] - 1, d->poll_national[j] - 1, full ? d->poll_mode_national[j] - 1 : 0,
                     full ? d->poll_pop_national[j] - 1 : 0, L.o_nn + j, (double)d->n_democrat_national[j],
                     (double)d->n_two_share_national[j], full ? d->unadjusted_national[j] : 0.0, d->sigma_measure_noise_national});
  std::stable_sort(polls.begin(), polls.end(), [](const Poll &a, const Poll &b) { return a.t != b.t ? a.t < b.t : a.s < b.s; });
  const int Np = (int)polls.size();
  std::vector<int> ps(Np), pt(Np), pp(Np), pm(Np), ppop(Np), pq(Np), day_ptr(T + 1, 0);
  std::vector<double> py(Np), pn(Np), pu(Np), psig(Np);
  for (int i = 0; i < Np; i++) {
    ps[i] = polls[i].s; pt[i] = polls[i].t; pp[i] = polls[i].p; pm[i] = polls[i].m; ppop[i] = polls[i].pop; pq[i] = polls[i].qidx;
    py[i] = polls[i].y; pn[i] = polls[i].n; pu[i] = polls[i].unadj; psig[i] = polls[i].sig;
    day_ptr[polls[i].t + 1]++;
  }
  for (int t = 0; t < T; t++) day_ptr[t + 1] += day_ptr[t];

  // per-day gathers: days balanced over the 16 waves (longest-processing-time first); each wave gets
  // the polls of its days as one contiguous entry list (index, day, state)
  std::vector<int> order, load(PT_NW, 0), npolls_w(PT_NW, 0);
  for (int t = 0; t < T; t++) if (day_ptr[t + 1] > day_ptr[t]) order.push_back(t);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return day_ptr[a + 1] - day_ptr[a] > day_ptr[b + 1] - day_ptr[b]; });
  std::vector<std::vector<int>> wt(PT_NW);
  for (int t : order) {
    int wmin = -1;   // least-loaded wave with a free day slot (64) and room for the polls (256 packed states)
    const int nt = day_ptr[t + 1] - day_ptr[t];
    for (int wv = 0; wv < PT_NW; wv++)
      if (wt[wv].size() < 64 && npolls_w[wv] + nt <= 256 && (wmin < 0 || load[wv] < load[wmin])) wmin = wv;
    if (wmin < 0) { k1_fail("poll schedule of one workgroup per chain does not fit: more than %d polls or %d polled days", 256 * PT_NW, 64 * PT_NW); break; }
    npolls_w[wmin] += nt;
    wt[wmin].push_back(t);
    load[wmin] += day_ptr[t + 1] - day_ptr[t] + 2;
  }
  std::vector<int> wd_t(PT_NW * 64, 0), wd_a(PT_NW * 64, 0), wd_b(PT_NW * 64, 0), wpk(PT_NW * 64, 0), daymask(PT_NW, 0);
  for (int wv = 0; wv < PT_NW; wv++) {
    int e = 0;
    for (size_t j = 0; j < wt[wv].size(); j++) {
      const int t = wt[wv][j];
      wd_t[wv * 64 + j] = t; wd_a[wv * 64 + j] = day_ptr[t]; wd_b[wv * 64 + j] = day_ptr[t + 1];
      for (int i = day_ptr[t]; i < day_ptr[t + 1]; i++, e++) if (e < 256) wpk[wv * 64 + e / 4] |= ps[i] << (8 * (e % 4));
    }
  }
  for (int t = 0; t < T && t < PT_NW * PT_CH; t++) if (day_ptr[t + 1] > day_ptr[t]) daymask[t / PT_CH] |= (int)(1u << (t % PT_CH));

  // two-level segment sums: level-1 tasks of <= PT_SUBLEN polls (padded with the zero slot Npoll),
  // level-2 one thread per segment
  std::vector<int> sub16, seg_ptr{0}, seg_kind, seg_index;
  std::vector<double> seg_scale, sub_wt16;
  int nsub = 0;
  auto add_group = [&](int nseg, int kind, int index0, double scale, bool weighted, auto key) {
    std::vector<std::vector<int>> lists(nseg);
    for (int i = 0; i < Np; i++) { const int k = key(i); if (k >= 0) lists[k].push_back(i); }
    for (int sgi = 0; sgi < nseg; sgi++) {
      const auto &l = lists[sgi];
      for (size_t a = 0; a < l.size(); a += PT_SUBLEN) {
        for (size_t j = a; j < a + PT_SUBLEN; j++) {
          sub16.push_back(j < l.size() ? l[j] : Np);
          if (weighted) sub_wt16.push_back(j < l.size() ? pu[l[j]] : 0.0);
        }
        nsub++;
      }
      seg_ptr.push_back(nsub);
      seg_kind.push_back(kind); seg_index.push_back(index0 + sgi); seg_scale.push_back(scale);
    }
  };
  add_group(d->P, 0, L.o_c, d->sigma_c, false, [&](int i) { return pp[i]; });
  if (full) {
    add_group(d->M, 0, L.o_m, d->sigma_m, false, [&](int i) { return pm[i]; });
    add_group(d->Pop, 0, L.o_pop, d->sigma_pop, false, [&](int i) { return ppop[i]; });
  }
  add_group(S + 1, 1, 0, 1.0, false, [&](int i) { return ps[i]; });
  M.sub_weighted_begin = nsub;
  if (full) add_group(T, 2, 0, 1.0, true, [&](int i) { return pt[i]; });
  M.nsub = nsub;
  M.nseg = (int)seg_kind.size();
  seg_ptr.push_back(nsub); // so that seg_ptr[seg + 1] is loadable for every seg

  // LDS layout (doubles)
  int o = 0;
  auto take = [&](int n) { const int a = o; o += (n + 1) & ~1; return a; };
  M.l_C = take(std::max(S * M.TP, 2 * PT_NW * M.SE));
  M.l_Lw = take(M.SE * M.SP);
  M.l_X = take(std::max(2 * PT_NW * M.SE, Np + 2));
  M.l_Y = take(std::max(PT_NW * M.SE, M.nsub));
  M.l_zT = take(S + 1); M.l_zb = take(S + 1); M.l_mid = take(M.nmid + 1);
  M.l_bT = take(M.SE); M.l_pb = take(M.SE); M.l_e = take(T); M.l_gs = take(M.SE); M.l_ge = take(T);
  M.l_scal = take(SC_N); M.l_red = take(PT_NW * PT_NRED); M.l_st = take((Np + 8 + 7) / 8);
#ifdef POTUS_PROF
  M.l_prof = take(PT_NPROF);
#else
  M.l_prof = 0;
#endif
  M.lds_doubles = o;
  sp->lds_bytes = (size_t)o * 8 + sizeof(TS) + 16;
  if (sp->lds_bytes > 160 * 1024) k1_fail("one workgroup per chain needs %zu bytes of LDS (> 160 KiB): T=%d, polls=%d", sp->lds_bytes, T, Np);

  int rc;
  // pack: mat = Lw_ext | LT_t | LB_t | LT | LB | prior | w
  std::vector<double> mat(Lw_ext);
  auto app = [&](const std::vector<double> &v) { const int off = (int)mat.size(); mat.insert(mat.end(), v.begin(), v.end()); return off; };
  M.m_LTt = app(LT_t); M.m_LBt = app(LB_t); M.m_LT = app(LTr); M.m_LB = app(LBr);
  M.m_prior = app(std::vector<double>(d->mu_b_prior, d->mu_b_prior + S)); M.m_w = app(w);
  {
    std::vector<double> px(d->mu_b_prior, d->mu_b_prior + S);
    double nat = 0; for (int s = 0; s < S; s++) nat += w[s] * px[s];   // stan:87 applied to the prior part of mu_b
    px.push_back(nat); px.push_back(0.0);
    M.m_priorx = app(px);
  }
  // (stan:50-52 uses square(scale): the sign of a scale does not reach the factors, chol(c^2 A) = |c| chol(A))
  M.aT = std::fabs(d->mu_b_T_scale / d->random_walk_scale); M.aB = std::fabs(d->polling_bias_scale / d->random_walk_scale);
  M.Npad = (Np + 15) & ~15;
  std::vector<int> pi((size_t)6 * M.Npad, 0);
  std::vector<double> pdv((size_t)4 * M.Npad, 0.0);
  for (int i = 0; i < Np; i++) {
    pi[i] = ps[i]; pi[M.Npad + i] = pt[i]; pi[2 * M.Npad + i] = pp[i]; pi[3 * M.Npad + i] = pm[i]; pi[4 * M.Npad + i] = ppop[i];
    pi[5 * M.Npad + i] = pq[i];
    pdv[i] = py[i]; pdv[M.Npad + i] = pn[i]; pdv[2 * M.Npad + i] = pu[i]; pdv[3 * M.Npad + i] = psig[i];
  }
  std::vector<int> sched(wd_t);
  auto appi = [&](const std::vector<int> &v) {
    while (sched.size() % 4) sched.push_back(0);   // 16-byte aligned blocks (sub16 is read with 128-bit loads)
    const int off = (int)sched.size();
    sched.insert(sched.end(), v.begin(), v.end());
    return off;
  };
  M.c_wda = appi(wd_a); M.c_wdb = appi(wd_b); M.c_wpk = appi(wpk); M.c_mask = appi(daymask);
  M.c_sub16 = appi(sub16); M.c_segptr = appi(seg_ptr); M.c_segkind = appi(seg_kind); M.c_segidx = appi(seg_index);
  for (int k = 0; k < 8; k++) sched.push_back(0);
  if (sub_wt16.empty()) sub_wt16.assign(2, 0.0);
  if ((rc = upload(sp, mat, &M.mat)) || (rc = upload(sp, pi, &M.pi)) || (rc = upload(sp, pdv, &M.pd)) ||
      (rc = upload(sp, sched, &M.sched)) || (rc = upload(sp, seg_scale, &M.seg_scale)) || (rc = upload(sp, sub_wt16, &M.sub_wt16)))
    return rc;
  sp->h_w = w;   // weighted.mean() normalises: keep them summing to one
  { double sw = 0; for (double x : sp->h_w) sw += x; for (double &x : sp->h_w) x /= sw; }
  sp->h_ps = ps; sp->h_pt = pt; sp->h_pp = pp; sp->h_pm = pm; sp->h_ppop = ppop; sp->h_pq = pq; sp->h_dayptr = day_ptr; sp->h_pu = pu;
  void *pdm = nullptr;
  HIP_TRY(hipMalloc(&pdm, sizeof(DevModel)));
  sp->allocs.push_back(pdm);
  HIP_TRY(hipMemcpy(pdm, &M, sizeof(DevModel), hipMemcpyHostToDevice));
  sp->dM = (DevModel *)pdm;
  return 0;
}

int set_lds_attr(Sampler *sp) {
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_logprob_grad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp->lds_bytes));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_init), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp->lds_bytes));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_run), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp->lds_bytes));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_run_twin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp->lds_bytes));
  return 0;
}

// Cluster mode: split the days over K members (contiguous ranges balanced by days + polls), give every
// parameter one owner, lay the owners' elements out contiguously and build each member's schedule.
int build_cluster(Sampler *sp, const potus_data *d, int K) {
  const DevModel &M = sp->M;
  const Layout &L = sp->L;
  const bool full = M.full;
  const int S = M.S, T = M.T, Np = M.Npoll, P = M.P;
  if (K < 2 || K > CL_MAXK) return fail(POTUS_ERR_ARG, "cus_per_chain must be 1 or in [2,%d]", CL_MAXK);
  ClModel &C = sp->CL;
  C.K = K;
  C.NR = 2 * S + P + (full ? M.M + M.Pop + 2 : 0);
  C.NREP = 2 * S + M.nmid;
  // days per wave: 4 (at most 32 days per member) when that leaves room to balance the members by polls, else 8
  // (up to 26 days per member on average: measured with 12 chains on two clusters of 10, profiles/r02_cl_cluster_sizes.txt)
  const int dw4_max = getenv("POTUS_CL_DW4_MAXAVG") ? atoi(getenv("POTUS_CL_DW4_MAXAVG")) : CL_DW4_MAXAVG;   // development: sweeps
  const int DW = (T + K - 1) / K <= dw4_max ? 4 : 8;
  // (development: POTUS_CL_MAXDAYS caps the days of a member below what its waves hold -- partition sweeps, profiles/r05_cl_partition.txt)
  const int maxdays = getenv("POTUS_CL_MAXDAYS") ? std::max(1, std::min(PT_NW * DW, atoi(getenv("POTUS_CL_MAXDAYS")))) : PT_NW * DW;
  sp->cl_dw = DW;   // (12 once the adjoint goes to the matrix cores, below)
  C.XW = (std::max(XP_P + C.NR, XQ0 + C.NREP) + 7) & ~7;
  if (P > 65535 || M.M > 255 || M.Pop > 255) return fail(POTUS_ERR_UNSUPPORTED, "cluster mode packs pollster/mode/population indices in 16/8/8 bits");
  if (C.NREP > 2 * PT_THREADS) return fail(POTUS_ERR_UNSUPPORTED, "cluster mode: %d small parameters (> %d)", C.NREP, 2 * PT_THREADS);
  if ((C.NR + K - 1) / K > PT_THREADS - 128 - 3) return fail(POTUS_ERR_UNSUPPORTED, "cluster mode: too many pollsters for K = %d", K);
  const std::vector<int> &dp = sp->h_dayptr;

  // contiguous day ranges: minimise the largest cost (a day = S elements of vector work, a poll = a
  // 51-term dot + a gather) with at most CL_MAXDAYS days each
  // (weights measured on the 2016 posterior, scripts/micro/r02_sweep.sh: a poll costs a member about as much as a day --
  //  dot, binomial term, adjoint gather -- 20.1 us per leapfrog against 21.1 with the earlier 51 : 10)
  const int cw_day = getenv("POTUS_CW_DAY") ? atoi(getenv("POTUS_CW_DAY")) : 30, cw_poll = getenv("POTUS_CW_POLL") ? atoi(getenv("POTUS_CW_POLL")) : 45;   // (round 4, without the 51 x 51 products: 30 : 45 / 30 : 60 / 20 : 30 13.9 us against 14.1 with 30 : 30)
  auto groups_for = [&](int B, std::vector<int> *cut) {
    int g = 0, t = 0;
    if (cut) cut->assign(1, 0);
    while (t < T) {
      int nd = 0, cost = 0;
      while (t < T && nd < maxdays) {
        const int c1 = cw_day + cw_poll * (dp[t + 1] - dp[t]);
        if (nd > 0 && cost + c1 > B) break;
        cost += c1; nd++; t++;
      }
      g++;
      if (cut) cut->push_back(t);
    }
    return g;
  };
  int lo = 1, hi = cw_day * T + cw_poll * Np;
  while (lo < hi) { const int mid = (lo + hi) / 2; if (groups_for(mid, nullptr) <= K) hi = mid; else lo = mid + 1; }
  std::vector<int> cut;
  if (groups_for(lo, &cut) > K) return fail(POTUS_ERR_UNSUPPORTED, "T = %d days do not fit %d members of at most %d days", T, K, maxdays);
  while ((int)cut.size() < K + 1) cut.push_back(T);   // members without days still own a share of the small vectors

  // the adjoint product on the fp64 matrix cores (potus_cluster.hpp, cl_pass_partial): G[pseudo-state][local day], row stride
  // = 16 mod 32 doubles (the four rows a wave reads per MFMA step fall on disjoint LDS banks)
  // -- for poll-dense posteriors (more than 8 polls per day: the dense product then beats the walk over the polls, measured in
  // profiles/r03_cl_mfma_adjoint.txt; the reference's own posteriors have 4-6 and keep the walk); POTUS_CL_MFMA = 0 / 1 overrides
  const bool mfma = DW == 4 && (getenv("POTUS_CL_MFMA") ? atoi(getenv("POTUS_CL_MFMA")) != 0 : Np > 8 * T);
  const int GS = 48, GROWS = 4 * ((M.SE + 3) / 4);
  if (mfma) sp->cl_dw = 12;
  std::vector<int> part((size_t)K * CP_N, 0), sched, perm;   // perm: internal index -> Stan index, -1 for padding
  std::vector<double> wts;
  int e = 0, npmax = 0, nsubmax = 0;
  for (int m = 0; m < K; m++) {
    int *pt_ = &part[(size_t)m * CP_N];
    const int d0 = cut[m], nd = cut[m + 1] - cut[m], p0 = dp[d0], np = dp[d0 + nd] - p0;
    const int r0 = (int)((long long)C.NR * m / K), nr = (int)((long long)C.NR * (m + 1) / K) - r0;
    pt_[CP_D0] = d0; pt_[CP_ND] = nd; pt_[CP_P0] = p0; pt_[CP_NP] = np; pt_[CP_E0] = e; pt_[CP_R0] = r0; pt_[CP_NR] = nr;
    if (getenv("POTUS_CL_VERBOSE")) fprintf(stderr, "cluster member %2d: days [%3d, %3d) = %2d, polls %4d, small-vector slots %3d\n", m, d0, d0 + nd, nd, np, nr);
    npmax = std::max(npmax, np);
    // internal order: days of raw_mu_b | noise of own polls | pad | own days of raw_e_bias | share of the small
    // vectors | pad.  The two blocks start on 128-byte lines: no cache line has two writers (members sit on
    // different compute units, possibly different XCDs) or mixes plain stores (first block: read only by the owner)
    // with write-through stores (second block: read by every member).
    auto pad16 = [&]() { while (perm.size() % 16) perm.push_back(-1); };
    for (int tl = 0; tl < nd; tl++) for (int k = 0; k < S; k++) perm.push_back(L.o_Z + k + S * (d0 + tl));
    for (int il = 0; il < np; il++) perm.push_back(sp->h_pq[p0 + il]);
    pad16();
    pt_[CP_E_SH] = (int)perm.size();
    if (full) for (int tl = 0; tl < nd; tl++) perm.push_back(L.o_ze + d0 + tl);
    for (int j = 0; j < nr; j++) { const int r = r0 + j; perm.push_back(r < S ? L.o_zT + r : r < 2 * S ? L.o_zb + (r - S) : L.o_c + (r - 2 * S)); }
    pad16();
    e = (int)perm.size();
    pt_[CP_NE] = e - pt_[CP_E0];

    // the member's days are dealt to the waves as contiguous ranges of at most CL_DW days, balanced by
    // polls (a wave gathers the adjoint of the days it owns): smallest feasible bound on polls per wave
    std::vector<int> wd0(PT_NW, nd), wnd(PT_NW, 0);
    {
      auto assign = [&](int B, bool commit) {
        int tl = 0;
        for (int wv = 0; wv < PT_NW; wv++) {
          int n = 0, polls = 0;
          const int start = tl;
          while (tl < nd && n < DW) {
            const int c1 = dp[d0 + tl + 1] - dp[d0 + tl];
            // leave enough room in the remaining waves for the remaining days
            if (n > 0 && polls + c1 > B && (nd - tl) <= (PT_NW - 1 - wv) * DW) break;
            polls += c1; n++; tl++;
          }
          if (commit) { wd0[wv] = start; wnd[wv] = n; }
        }
        return tl == nd;
      };
      int lo2 = 0, hi2 = np + 1;
      auto maxpolls = [&](int B) {   // feasibility: all days placed and no wave above the bound unless forced by one day
        int tl = 0, worst = 0;
        for (int wv = 0; wv < PT_NW; wv++) {
          int n = 0, polls = 0;
          while (tl < nd && n < DW) {
            const int c1 = dp[d0 + tl + 1] - dp[d0 + tl];
            if (n > 0 && polls + c1 > B && (nd - tl) <= (PT_NW - 1 - wv) * DW) break;
            polls += c1; n++; tl++;
          }
          worst = std::max(worst, polls);
        }
        return tl == nd ? worst : 1 << 30;
      };
      int best = hi2, bestB = hi2;
      for (int B = lo2; B <= hi2; B++) { const int w_ = maxpolls(B); if (w_ < best) { best = w_; bestB = B; } }
      if (!assign(bestB, true)) return fail(POTUS_ERR_STATE, "internal: member %d: %d days do not fit %d waves", m, nd, PT_NW);
    }
    // adjoint gather: equal chunks of the member's polls per wave; program word per poll; what the owner of
    // each day reads back (last polled day <= it, chunk of that day's last poll)
    std::vector<int> ca(PT_NW), cb(PT_NW), tab(np), s2info(PT_THREADS, 0);
    for (int wv = 0; wv < PT_NW; wv++) { ca[wv] = (int)((long long)np * std::min(wv, CL_NCHUNK) / CL_NCHUNK); cb[wv] = (int)((long long)np * std::min(wv + 1, CL_NCHUNK) / CL_NCHUNK); }
    auto chunk_of = [&](int il) { int c = 0; while (c < CL_NCHUNK - 1 && il >= cb[c]) c++; return c; };
    for (int il = 0; il < np; il++) {
      const int g = p0 + il, tl = sp->h_pt[g] - d0;
      const bool dayend = il == np - 1 || sp->h_pt[g + 1] != sp->h_pt[g];
      tab[il] = sp->h_ps[g] | (tl << 8) | ((dayend ? 1 : 0) << 16);
    }
    for (int wv = 0; wv < PT_NW; wv++)
      for (int j = 0; j < wnd[wv]; j++) {
        int tl = wd0[wv] + j;
        while (tl >= 0 && dp[d0 + tl + 1] == dp[d0 + tl]) tl--;   // last day with polls at or before this one
        s2info[wv * 64 + j] = tl < 0 ? 0 : ((tl + 1) | (chunk_of(dp[d0 + tl + 1] - 1 - p0) << 8));
      }
    // adjoint on the matrix cores (4-days-per-wave build): the (state, day) cells of the member's polls -- runs of its
    // day-then-state sorted polls -- one per thread: first poll | polls << 10 | offset in G << 16
    std::vector<int> cellw((size_t)CL_CELLS_PER_THREAD * PT_THREADS, 0);
    int ncell = 0;
    if (mfma) {
      const int dump = GROWS * GS;
      if (np >= 1024) return fail(POTUS_ERR_UNSUPPORTED, "cluster mode: member %d has %d polls (the adjoint scatter addresses 1023)", m, np);
      for (auto &cw : cellw) cw = np | (1 << 10) | (dump << 16);
      for (int il = 0; il < np;) {
        int j = il + 1;
        while (j < np && sp->h_pt[p0 + j] == sp->h_pt[p0 + il] && sp->h_ps[p0 + j] == sp->h_ps[p0 + il]) j++;
        {
          if (j - il > 63) return fail(POTUS_ERR_UNSUPPORTED, "cluster mode: %d polls of one state on one day (the adjoint scatter takes 63)", j - il);
          if (ncell >= CL_CELLS_PER_THREAD * PT_THREADS) return fail(POTUS_ERR_UNSUPPORTED, "cluster mode: member %d has more than %d polled (state, day) cells", m, CL_CELLS_PER_THREAD * PT_THREADS);
          cellw[ncell++] = il | ((j - il) << 10) | ((sp->h_ps[p0 + il] * GS + (sp->h_pt[p0 + il] - d0)) << 16);
        }
        il = j;
      }
    }
    pt_[CP_NCELL] = ncell;
    std::vector<int> wdays(wd0);
    wdays.insert(wdays.end(), wnd.begin(), wnd.end());
    wdays.insert(wdays.end(), ca.begin(), ca.end());
    wdays.insert(wdays.end(), cb.begin(), cb.end());

    // two-level segment sums over the member's polls (local poll indices, padded with the zero slot np)
    std::vector<int> sub16, seg_ptr{0}, seg_kind, seg_index;
    std::vector<double> sub_wt;
    int nsub = 0;
    auto add_group = [&](int nseg, int kind, int index0, bool weighted, auto key) {
      std::vector<std::vector<int>> lists(nseg);
      for (int il = 0; il < np; il++) lists[key(p0 + il)].push_back(il);
      for (int sgi = 0; sgi < nseg; sgi++) {
        const auto &l = lists[sgi];
        if (l.empty()) continue;
        for (size_t a = 0; a < l.size(); a += PT_SUBLEN) {
          for (size_t j = a; j < a + PT_SUBLEN; j++) {
            sub16.push_back(j < l.size() ? l[j] : np);
            if (weighted) sub_wt.push_back(j < l.size() ? sp->h_pu[p0 + l[j]] : 0.0);
          }
          nsub++;
        }
        seg_ptr.push_back(nsub);
        seg_kind.push_back(kind); seg_index.push_back(index0 + sgi);
      }
    };
    add_group(P, 0, 2 * S, false, [&](int i) { return sp->h_pp[i]; });
    if (full) {
      add_group(M.M, 0, 2 * S + P, false, [&](int i) { return sp->h_pm[i]; });
      add_group(M.Pop, 0, 2 * S + P + M.M, false, [&](int i) { return sp->h_ppop[i]; });
    }
    const int wb = nsub;   // tasks from here on sum unadjusted * residual
    if (full) add_group(nd, 2, 0, false, [&](int i) { return sp->h_pt[i] - d0; });
    seg_ptr.push_back(nsub);
    if ((int)seg_kind.size() > PT_THREADS) return fail(POTUS_ERR_UNSUPPORTED, "cluster mode: member %d has %zu segments (> %d)", m, seg_kind.size(), PT_THREADS);
    pt_[CP_NSUB] = nsub; pt_[CP_NSEG] = (int)seg_kind.size(); pt_[CP_WB] = wb;
    nsubmax = std::max(nsubmax, nsub);
    auto appi = [&](const std::vector<int> &v) {
      while (sched.size() % 4) sched.push_back(0);
      const int off = (int)sched.size();
      sched.insert(sched.end(), v.begin(), v.end());
      return off;
    };
    pt_[CP_O_WD] = appi(s2info); appi(wdays);
    pt_[CP_O_MASK] = appi(tab); pt_[CP_O_SUB] = appi(sub16); pt_[CP_O_SEGPTR] = appi(seg_ptr);
    pt_[CP_O_SEGKIND] = appi(seg_kind); pt_[CP_O_SEGIDX] = appi(seg_index);
    pt_[CP_O_CELL] = appi(cellw);
    while (wts.size() % 2) wts.push_back(0.0);
    pt_[CP_O_WT] = (int)wts.size();
    wts.insert(wts.end(), sub_wt.begin(), sub_wt.end());
  }
  const int Dint = e;   // internal length of a vector (parameters + padding)
  C.Dint = Dint;
  for (int k = 0; k < 16; k++) { sched.push_back(0); wts.push_back(0.0); }
  C.npmax = npmax; C.nsubmax = nsubmax;

  std::vector<int> iperm(L.D, -1);
  for (int i = 0; i < Dint; i++) { if (perm[i] >= L.D) return fail(POTUS_ERR_STATE, "internal: bad permutation"); if (perm[i] >= 0) iperm[perm[i]] = i; }
  for (int i = 0; i < L.D; i++) if (iperm[i] < 0) return fail(POTUS_ERR_STATE, "internal: cluster layout misses parameter %d", i);
  std::vector<int> rep_pos(C.NREP), rep_owner(C.NREP);
  for (int j = 0; j < C.NREP; j++) {
    rep_pos[j] = iperm[j < S ? L.o_zT + j : j < 2 * S ? L.o_zb + (j - S) : L.o_c + (j - 2 * S)];
    int mo = 0;
    while (mo < K - 1 && rep_pos[j] >= part[(size_t)mo * CP_N + CP_E0] + part[(size_t)mo * CP_N + CP_NE]) mo++;   // owner = member whose element range holds it
    rep_owner[j] = mo;
  }
  std::vector<double> rep_scale(C.NR, 1.0);
  for (int i = 0; i < S; i++) { rep_scale[i] = M.aT; rep_scale[S + i] = M.aB; }   // L_T = aT L_W, L_B = aB L_W (DevModel::aT)
  for (int i = 0; i < P; i++) rep_scale[2 * S + i] = d->sigma_c;
  if (full) {
    for (int i = 0; i < M.M; i++) rep_scale[2 * S + P + i] = d->sigma_m;
    for (int i = 0; i < M.Pop; i++) rep_scale[2 * S + P + M.M + i] = d->sigma_pop;
  }
  sp->h_perm = perm;

  int ndmax = 1, nemax = 0;
  for (int m = 0; m < K; m++) { ndmax = std::max(ndmax, part[(size_t)m * CP_N + CP_ND]); nemax = std::max(nemax, part[(size_t)m * CP_N + CP_NE]); }
  C.NDP = ndmax | 1;   // odd row stride of the member's C[state][local day] block
  C.GS = GS; C.GROWS = GROWS;
  ClLay lay = cl_layout(S, M.SE, M.SP, C.NDP, npmax, nsubmax, C.NREP, C.NR, T, mfma ? GROWS * GS + CL_G_PAD : 0);
  // The builds with the layout fixed at compile time (tags 16 and 17 = full model / no_mode_adjustment variant, ClFixed): 51 states, members of
  // at most 32 days / 256 polls / 384 level-1 tasks, the walk for the adjoint.  The layout is then the same function of ClFixed's capacities -- the numbers the
  // kernel uses as immediates.
  {
    const bool fits = DW == 4 && !mfma && K <= ClFixed::KMAX && std::max(XP_P + C.NR, XQ0 + C.NREP) <= ClFixed::XW && S == ClFixed::S && M.SE == ClFixed::SE && M.SP == ClFixed::SP && ndmax <= ClFixed::NDP - 1 && npmax <= ClFixed::NPCAP &&
                      nsubmax <= ClFixed::NSUBCAP && C.NREP <= ClFixed::NREPCAP && C.NR <= ClFixed::NRCAP && T <= ClFixed::TCAP && (int)SC_N <= 8 &&
                      nemax <= ClFixed::NECAP;                // the member's resident share of a vector (ClLeapPolicyRes)
    if (fits && !(getenv("POTUS_CL_DYNAMIC") && atoi(getenv("POTUS_CL_DYNAMIC")))) {
      C.NDP = ClFixed::NDP; C.XW = ClFixed::XW;
      lay = ClFixed::L;
      sp->cl_dw = full ? 16 : 17;       // the variant is a constant of the build as well (ClTag::FULL)
    }
  }
  C.l_C = lay.l_C; C.l_G = lay.l_G; C.l_Lw = lay.l_Lw; C.l_prior = lay.l_prior; C.l_pm = lay.l_pm; C.l_py = lay.l_py; C.l_pun = lay.l_pun;
  C.l_sub = lay.l_sub; C.l_tab = lay.l_tab; C.l_ru = lay.l_ru; C.l_wide = lay.l_wide; C.l_wout = lay.l_wout; C.l_X = lay.l_X; C.l_Y = lay.l_Y;
  C.l_r = lay.l_r; C.l_rep = lay.l_rep; C.l_bT = lay.l_bT; C.l_e = lay.l_e; C.l_c1 = lay.l_c1; C.l_c2 = lay.l_c2; C.l_c3 = lay.l_c3; C.l_ge = lay.l_ge;
  C.l_P = lay.l_P; C.l_scal = lay.l_scal; C.l_red = lay.l_red; C.l_st = lay.l_st; C.l_prof = lay.l_prof; C.lds_doubles = lay.total;
  const int o = lay.total;
  sp->cl_lds_bytes = (size_t)o * 8 + sizeof(TS) + 16;
  if (sp->cl_lds_bytes > 160 * 1024 - 64) return fail(POTUS_ERR_UNSUPPORTED, "cluster mode needs %zu bytes of LDS per workgroup", sp->cl_lds_bytes);

  int rc;
  if ((rc = upload(sp, part, &C.part)) || (rc = upload(sp, sched, &C.sched)) || (rc = upload(sp, wts, &C.wt)) ||
      (rc = upload(sp, rep_pos, &C.rep_pos)) || (rc = upload(sp, rep_owner, &C.rep_owner)) || (rc = upload(sp, rep_scale, &C.rep_scale)) || (rc = upload(sp, perm, &C.perm)))
    return rc;
  void *pc = nullptr;
  HIP_TRY(hipMalloc(&pc, sizeof(ClModel)));
  sp->allocs.push_back(pc);
  HIP_TRY(hipMemcpy(pc, &C, sizeof(ClModel), hipMemcpyHostToDevice));
  sp->dCL = (ClModel *)pc;
  for (const void *f : {reinterpret_cast<const void *>(k_cl_logprob_grad<4>), reinterpret_cast<const void *>(k_cl_logprob_grad<8>), reinterpret_cast<const void *>(k_cl_logprob_grad<12>),
                        reinterpret_cast<const void *>(k_cl_logprob_grad<16>), reinterpret_cast<const void *>(k_cl_init<16>),
                        reinterpret_cast<const void *>(k_cl_run<16, false>), reinterpret_cast<const void *>(k_cl_run<16, true>),
                        reinterpret_cast<const void *>(k_cl_logprob_grad<17>), reinterpret_cast<const void *>(k_cl_init<17>),
                        reinterpret_cast<const void *>(k_cl_run<17, false>), reinterpret_cast<const void *>(k_cl_run<17, true>),
                        reinterpret_cast<const void *>(k_cl_init<4>), reinterpret_cast<const void *>(k_cl_init<8>), reinterpret_cast<const void *>(k_cl_init<12>),
                        reinterpret_cast<const void *>(k_cl_run<4, false>), reinterpret_cast<const void *>(k_cl_run<8, false>), reinterpret_cast<const void *>(k_cl_run<12, false>),
                        reinterpret_cast<const void *>(k_cl_run<4, true>), reinterpret_cast<const void *>(k_cl_run<8, true>), reinterpret_cast<const void *>(k_cl_run<12, true>)})
    HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp->cl_lds_bytes));
  sp->K = K;
  return 0;
}

// The builds of the cluster pass (template tag of potus_cluster.hpp): 4 = four days per wave, 8 = eight days per wave, 12 = four
// days per wave with the adjoint product on the fp64 matrix cores (poll-dense posteriors), 16 = four days per wave, 51 states, LDS
// layout fixed at compile time (ClFixed: the reference's posteriors on clusters of 16).
#define CL_DISPATCH(tag, CALL) do { if ((tag) == 16) { CALL(16); } else if ((tag) == 17) { CALL(17); } else if ((tag) == 4) { CALL(4); } else if ((tag) == 12) { CALL(12); } else { CALL(8); } } while (0)

// replica 0 of every chain's scalars
int read_scalars(Sampler *sp, std::vector<ChainScalars> &sc) {
  std::vector<ChainScalars> all((size_t)sp->R.chains * sp->K);
  HIP_TRY(hipMemcpy(all.data(), sp->R.scal, sizeof(ChainScalars) * all.size(), hipMemcpyDeviceToHost));
  sc.resize(sp->R.chains);
  for (int c = 0; c < sp->R.chains; c++) sc[c] = all[(size_t)c * sp->K];
  return 0;
}

// Anything that makes the chains of a handle unusable: a chain whose step-size search ran away (Stan throws there),
// a cluster launch that gave up waiting for a member (watchdog word raised by the device).
int check_chains(Sampler *sp) {
  std::vector<ChainScalars> sc;
  { const int rc_ = read_scalars(sp, sc); if (rc_) return rc_; }
  if (sp->coop()) {
    const size_t per_chain = 4 * (size_t)sp->K * sp->CL.XW + 8;   // 16-byte words: exchange slots + the watchdog line (K = 1: XW = 0)
    for (int b = 0; b < sp->R.chains * sp->sides(); b++) {
      const int c = b % sp->R.chains;
      unsigned wd = 0, wt = 0;
      HIP_TRY(hipMemcpy(&wd, (const char *)sp->R.xbuf + ((size_t)b * per_chain + 4 * (size_t)sp->K * sp->CL.XW) * 16, 4, hipMemcpyDeviceToHost));
      if (sp->twin) HIP_TRY(hipMemcpy(&wt, (const char *)sp->R.twbuf + ((size_t)c * TWB_WORDS + TWB_WD) * 16, 4, hipMemcpyDeviceToHost));
      wd |= wt;
      if (wd) return fail(POTUS_ERR_WATCHDOG, "chain %d: the %d workgroups working for it were not running together (is another process using GPU %d?); "
                                              "the launch was abandoned and this handle is no longer usable", c + 1, sp->K * sp->sides(), sp->device);
    }
  }
  for (int c = 0; c < sp->R.chains; c++) {
    if (sc[c].status == POTUS_ERR_INIT) return fail(POTUS_ERR_INIT, "chain %d: no finite initial log density/gradient after 100 attempts", c + 1);
    if (sc[c].status == POTUS_ERR_STEPSIZE) return fail(POTUS_ERR_STEPSIZE, "chain %d: the step-size search left (0, 1e7) -- the posterior is improper or the gradient is wrong (Stan throws here)", c + 1);
    if (sc[c].status == POTUS_ERR_WATCHDOG) return fail(POTUS_ERR_WATCHDOG, "chain %d: cluster launch abandoned (watchdog)", c + 1);
    if (sc[c].status != 0) return fail(POTUS_ERR_STATE, "chain %d: status %d", c + 1, sc[c].status);
  }
  return 0;
}


// ======================================================================== dense metric: host side (potus_dense.hpp)
// windowed_adaptation's schedule, shared by every chain: the longest window = capacity of the draw buffer
int dense_window_capacity(int nw, int ib, int tb, int bw) {
  if (nw < 20) return 1;
  int cap = 1, n = 0, counter = 0, next = ib + bw - 1, size = bw;
  for (int it = 0; it < nw; it++) {
    const bool in = counter >= ib && counter < nw - tb && counter != nw, end = counter == next && counter != nw;
    if (in) n++;
    cap = std::max(cap, n);
    if (end) {
      const int last = nw - tb - 1;
      if (next != last) { size *= 2; next = counter + size; if (next != last && next + 2 * size >= nw - tb) next = last; }
      n = 0;
    }
    counter++;
  }
  return cap;
}

// Blocked Cholesky factorisation in place on the lower triangles of the chains' matrices, in panels of DN_PANEL block columns:
// inside a panel potrf / trsm / narrow update per block column, behind it one wide update (k_dn_syrk_wide).
void dense_cholesky_launch(hipStream_t st, const DnParams &P, int chains) {
  const int nb = (P.D + DN_NB - 1) / DN_NB;
  for (int pb = 0; pb < nb; pb += DN_PANEL) {
    const int nk = std::min(DN_PANEL, nb - pb), last = pb + nk - 1;
    for (int kb = pb; kb <= last; kb++) {
      hipLaunchKernelGGL(k_dn_potrf, dim3(chains), dim3(256), 0, st, P, kb);
      const int rem = nb - kb - 1;
      if (rem > 0) {
        hipLaunchKernelGGL(k_dn_trsm, dim3(rem, chains), dim3(256), 0, st, P, kb);
        if (last > kb) hipLaunchKernelGGL(k_dn_syrk, dim3(rem, last - kb, chains), dim3(256), 0, st, P, kb, last);
      }
    }
    const int behind = P.D - (pb + nk) * DN_NB;
    if (behind > 0) {
      const int nt = (behind + DN_WT - 1) / DN_WT;
      hipLaunchKernelGGL(k_dn_syrk_wide, dim3(nt, nt, chains), dim3(256), 0, st, P, pb, nk);
    }
  }
}

// Shape of one launch of the symmetric product: rows a workgroup takes at a time (256 for large matrices: fewer tiles per
// byte) and, by the number of chains that take part, workgroups per pair of row blocks (enough to keep 256 compute units
// busy when the active chains are few; the results do not depend on it).
void dense_launch_shape(DnParams &P, int active) {
  active = std::max(active, 1);
  P.rb = P.D >= 32 * DN_RB_MAX ? DN_RB_MAX : P.D >= 32 * DN_RB_BIG ? DN_RB_BIG : DN_RB;   // by the size of the matrix alone: a chain's numbers must not depend on its companions
  P.ntile = (P.D + DN_CT - 1) / DN_CT;
  const int pairs = ((P.D + P.rb - 1) / P.rb + 1) / 2;
  // One workgroup per compute unit at a time (DN_SYMV_LDS(2) = 90 KB of LDS with DN_RB_MAX = 512): the launch runs in waves of 256 workgroups.  Few, large workgroups are
  // best (measured: profiles/r02_dense_active_sweep.txt), so: the smallest split whose last wave is at least 90 % full.
  const int base = active * pairs;
  P.split = 1;
  if (const char *e = getenv("POTUS_DENSE_SPLIT")) { P.split = std::max(1, std::min(DN_SPLIT_MAX, atoi(e))); return; }   // development: sweeps of scripts/micro/dense_probe.py
  double best = 0.0;
  for (int s = 1; s <= DN_SPLIT_MAX; s++) {
    const int w = base * s;
    const double fill = (double)w / (((w + 255) / 256) * 256);
    if (fill > best + 1e-9) { best = fill; P.split = s; }
    if (fill >= 0.9) { P.split = s; break; }
  }
}

// bytes one pass of the symmetric product loads per chain: the tiles right of (and on) each row block's diagonal
long long dense_pass_bytes(int D, int LD, int rb) {
  long long n = 0;
  const int tile_end = std::min(LD, ((D + DN_CT - 1) / DN_CT) * DN_CT);
  for (int r0 = 0; r0 < D; r0 += rb) n += (long long)std::min(rb, D - r0) * (tile_end - (r0 / DN_CT) * DN_CT);
  return n * 8;
}

int dense_alloc(Sampler *sp) {
  DnParams &P = sp->dn;
  const int D = sp->L.D, chains = sp->R.chains;
  P.chains = chains; P.D = D; P.LD = (D + 7) & ~7; P.npart = (D + DN_FIN / 4 - 1) / (DN_FIN / 4); P.nblk = (D + DN_RB - 1) / DN_RB; P.sc_stride = sp->K; P.identity = 1;
  P.win_cap = dense_window_capacity(sp->R.num_warmup, sp->R.init_buffer, sp->R.term_buffer, sp->R.window);
  P.f32 = sp->opts.metric_storage == POTUS_STORAGE_F32 ? 1 : 0;
  P.pooled = sp->opts.pooled_metric ? 1 : 0;
  dense_launch_shape(P, chains);
  // pooled metric: ONE full symmetric matrix + ONE factor for the handle, and the partial products of the row splits.  The number of splits is a
  // function of D alone (the order in which a product is summed must not depend on who else is active): the largest one (up to DNP_SPLIT_MAX) that keeps the launch
  // within two workgroups of 256 columns per compute unit -- all resident at once, no tail --, in whole batches of DNP_KB rows.
  P.pool_split = 1; P.pool_rows = D;
  if (P.pooled) {
    const int panels = (D + DNP_COLS_OF(P.f32) - 1) / DNP_COLS_OF(P.f32);
    int s = std::max(1, std::min(DNP_SPLIT_MAX, (2 * 256) / panels));   // at most two workgroups per compute unit, all resident at once (measured: profiles/r06_dense_pooled.txt)
    if (const char *e = getenv("POTUS_POOL_SPLIT")) s = std::max(1, std::min(DNP_SPLIT_MAX, atoi(e)));   // development: sweeps
    P.pool_rows = (((D + s - 1) / s + DNP_KB - 1) / DNP_KB) * DNP_KB;
    P.pool_split = (D + P.pool_rows - 1) / P.pool_rows;
  }
  const int mchains = P.pooled ? 1 : chains;
  const size_t mat = (size_t)mchains * D * P.LD * 8 * (P.pooled ? 2 : 1) + (P.pooled ? (size_t)P.pool_split * DNP_RMAX * P.LD * 8 : 0) + (P.pooled && P.f32 ? (size_t)D * P.LD * 4 : 0), vec = (size_t)chains * DV_COUNT * P.LD * 8, win = (size_t)chains * P.win_cap * P.LD * 8;
  const size_t tp = (size_t)chains * (P.nblk + P.ntile) * 3 * P.LD * 8;   // column sums per row block + row sums per column tile
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  if (mat + vec + win + tp + (64u << 20) > free_b)
    return fail(POTUS_ERR_UNSUPPORTED, "dense metric: %d chains x (D^2 + %d window draws + %d vectors + %d partial vectors) x 8 bytes = %.1f GB, %.1f GB free on GPU %d "
                                       "(D = %d: %.2f GB per matrix; potus_opts.pooled_metric keeps one matrix and one factor for all chains of the handle)", chains, P.win_cap, DV_COUNT, 3 * P.nblk, (mat + vec + win + tp) / 1e9, free_b / 1e9, sp->device, D, (double)D * P.LD * 8 / 1e9);
  auto get = [&](void **q, size_t bytes) {
    if (hipMalloc(q, std::max<size_t>(bytes, 8)) != hipSuccess) return fail(POTUS_ERR_DEVICE, "hipMalloc(%zu) for the dense metric failed", bytes);
    sp->allocs.push_back(*q);
    // (on the sampler's own stream: it is a non-blocking stream, which a memset on the null stream would not order with)
    return hipMemsetAsync(*q, 0, std::max<size_t>(bytes, 8), sp->stream) == hipSuccess ? 0 : fail(POTUS_ERR_DEVICE, "hipMemset failed");
  };
  int rc;
  if ((rc = get((void **)&P.state, vec)) || (rc = get((void **)&P.A, (size_t)mchains * D * P.LD * 8)) || (rc = get((void **)&P.dg, (size_t)mchains * P.LD * 8)) || (rc = get((void **)&P.win, win)) ||
      (P.pooled && ((rc = get((void **)&P.Lf, (size_t)D * P.LD * 8)) || (rc = get((void **)&P.ypool, (size_t)P.pool_split * DNP_RMAX * P.LD * 8)) ||
                    (rc = get((void **)&P.pmean, (size_t)P.LD * 8)) || (P.f32 && (rc = get((void **)&P.A32, (size_t)D * P.LD * 4))))) ||
      (rc = get((void **)&P.tpart, (size_t)chains * P.nblk * 3 * P.LD * 8)) || (rc = get((void **)&P.srow, (size_t)chains * 3 * P.ntile * P.LD * 8)) ||
      (rc = get((void **)&P.partial, (size_t)chains * P.npart * 8)) || (rc = get((void **)&P.lpbuf, (size_t)chains * 8)) ||
      (rc = get((void **)&P.ts, (size_t)chains * sizeof(TS))) || (rc = get((void **)&P.rd, (size_t)chains * sizeof(DnRound))) ||
      (rc = get((void **)&P.active, (size_t)chains * 4)) || (rc = get((void **)&P.fail, 4)) || (rc = get((void **)&P.act_passes, 8)))
    return rc;
  HIP_TRY(hipHostMalloc((void **)&sp->h_active, (size_t)chains * sizeof(int)));
  HIP_TRY(hipEventCreate(&sp->mv0)); HIP_TRY(hipEventCreate(&sp->mv1));
  for (int k = 0; k < Sampler::DN_AHEAD; k++) { HIP_TRY(hipEventCreate(&sp->rv0[k])); HIP_TRY(hipEventCreate(&sp->rv1[k])); HIP_TRY(hipEventCreate(&sp->rdone[k])); }
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dn_symv<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DN_SYMV_LDS(1)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dn_symv<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DN_SYMV_LDS(2)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dn_symv<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DN_SYMV_LDS(1)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dn_symv<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DN_SYMV_LDS(2)));
  if (sp->K == 1) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dn_grad1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp->lds_bytes));
  else for (const void *f : {reinterpret_cast<const void *>(k_dn_gradK<4>), reinterpret_cast<const void *>(k_dn_gradK<8>), reinterpret_cast<const void *>(k_dn_gradK<12>), reinterpret_cast<const void *>(k_dn_gradK<16>), reinterpret_cast<const void *>(k_dn_gradK<17>)})
    HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp->cl_lds_bytes));
  for (const void *f : {reinterpret_cast<const void *>(k_dn_pool_mm<1>), reinterpret_cast<const void *>(k_dn_pool_mm<2>), reinterpret_cast<const void *>(k_dn_pool_mm<3>),
                        reinterpret_cast<const void *>(k_dn_pool_mm<1, true>), reinterpret_cast<const void *>(k_dn_pool_mm<2, true>), reinterpret_cast<const void *>(k_dn_pool_mm<3, true>)})
    HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DNP_LDS(3)));
  hipLaunchKernelGGL(k_dn_identity, dim3((D + 255) / 256, mchains), dim3(256), 0, sp->stream, P);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(sp->stream));
  sp->dn_win_counter = 0; sp->dn_win_size = sp->R.window; sp->dn_win_next = sp->R.init_buffer + sp->R.window - 1; sp->dn_wf_n = 0;
  sp->dn_pass_bytes[0] = dense_pass_bytes(D, P.LD, DN_RB) / (P.f32 ? 2 : 1); sp->dn_pass_bytes[1] = dense_pass_bytes(D, P.LD, P.rb) / (P.f32 ? 2 : 1);   // ([1]: at the handle's own block size)
  if (P.pooled) sp->dn_pass_bytes[0] = sp->dn_pass_bytes[1] = (long long)D * P.LD * (P.f32 ? 4 : 8);   // the whole symmetric matrix, once per pass whatever the number of chains
  return 0;
}

// M^-1 times the round's right-hand sides for the active chains: one pass over the upper triangles (two launches when
// there are three right-hand sides) and the finishing kernel
void dense_symv_launch(hipStream_t st, const DnParams &P, const DnActive &act, int nrhs) {
  if (P.pooled) {
    // one pass over the handle's ONE matrix for every right-hand side of every chain that takes part (potus_dense_pool.hpp); more than
    // DNP_RMAX of them (over sixteen chains with three each) go in several launches, whole chains at a time
    const int nact = act.n ? act.n : P.chains, per = DNP_RMAX / nrhs;
    for (int c0 = 0; c0 < nact; c0 += per) {
      DnPoolRhs R;
      const int nc = std::min(per, nact - c0);
      R.n = nc * nrhs; R.nrhs = nrhs; R.job0 = 0;
      for (int r = 0; r < DNP_RMAX; r++) R.chain[r] = (unsigned char)(r < R.n ? (act.n ? act.idx[c0 + r / nrhs] : c0 + r / nrhs) : 0);
      const dim3 grid((unsigned)((P.D + DNP_COLS_OF(P.f32) - 1) / DNP_COLS_OF(P.f32)), (unsigned)P.pool_split);
      const int nt = (R.n + 15) / 16;
      if (P.f32) {
        if (nt == 1) hipLaunchKernelGGL((k_dn_pool_mm<1, true>), grid, dim3(DNP_THREADS), DNP_LDS(1), st, P, R, P.pool_rows);
        else if (nt == 2) hipLaunchKernelGGL((k_dn_pool_mm<2, true>), grid, dim3(DNP_THREADS), DNP_LDS(2), st, P, R, P.pool_rows);
        else hipLaunchKernelGGL((k_dn_pool_mm<3, true>), grid, dim3(DNP_THREADS), DNP_LDS(3), st, P, R, P.pool_rows);
      } else if (nt == 1) hipLaunchKernelGGL(k_dn_pool_mm<1>, grid, dim3(DNP_THREADS), DNP_LDS(1), st, P, R, P.pool_rows);
      else if (nt == 2) hipLaunchKernelGGL(k_dn_pool_mm<2>, grid, dim3(DNP_THREADS), DNP_LDS(2), st, P, R, P.pool_rows);
      else hipLaunchKernelGGL(k_dn_pool_mm<3>, grid, dim3(DNP_THREADS), DNP_LDS(3), st, P, R, P.pool_rows);
      hipLaunchKernelGGL(k_dn_pool_finish, dim3((unsigned)P.npart, (unsigned)R.n), dim3(64), 0, st, P, R, P.pool_split);
    }
    return;
  }
  const int nblk = (P.D + P.rb - 1) / P.rb;
  const unsigned ny = (unsigned)(act.n ? act.n : P.chains);
  const dim3 grid((unsigned)(((nblk + 1) / 2) * P.split), ny), fin((unsigned)P.npart, ny);
  auto one = [&](int job0) {
    if (P.f32) hipLaunchKernelGGL((k_dn_symv<1, true>), grid, dim3(DN_THREADS), DN_SYMV_LDS(1), st, P, act, job0);
    else hipLaunchKernelGGL((k_dn_symv<1, false>), grid, dim3(DN_THREADS), DN_SYMV_LDS(1), st, P, act, job0);
    hipLaunchKernelGGL(k_dn_symv_finish<1>, fin, dim3(DN_FIN), 0, st, P, act, job0);
  };
  if (nrhs == 1) one(0);
  else {
    if (P.f32) hipLaunchKernelGGL((k_dn_symv<2, true>), grid, dim3(DN_THREADS), DN_SYMV_LDS(2), st, P, act, 0);
    else hipLaunchKernelGGL((k_dn_symv<2, false>), grid, dim3(DN_THREADS), DN_SYMV_LDS(2), st, P, act, 0);
    hipLaunchKernelGGL(k_dn_symv_finish<2>, fin, dim3(DN_FIN), 0, st, P, act, 0);
    if (nrhs == 3) one(2);
  }
}

inline dim3 dn_grid(const Sampler *sp) { return dim3((unsigned)std::min((sp->L.D + 255) / 256, 64), (unsigned)sp->R.chains); }

int dense_grad(Sampler *sp) {
  if (sp->K == 1)
    hipLaunchKernelGGL(k_dn_grad1, dim3(sp->R.chains), dim3(PT_THREADS), sp->lds_bytes, sp->stream, (const DevModel *)sp->dM, sp->dn);
  else {
    const unsigned lid = ++sp->launch_id;
#define POTUS_CALL(TAG) hipLaunchKernelGGL(k_dn_gradK<TAG>, dim3(sp->R.chains * sp->K), dim3(PT_THREADS), sp->cl_lds_bytes, sp->stream, (const DevModel *)sp->dM, \
                                           (const ClModel *)sp->dCL, (const RunParams *)sp->dR, sp->dn, lid)
    CL_DISPATCH(sp->cl_dw, POTUS_CALL);
#undef POTUS_CALL
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
// one pass over the matrices of the active chains; the pass is timed with events resolved at the next sync point
int dense_matvec(Sampler *sp, int nrhs, int n_active, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, bool timed = true) {
  DnActive act;
  act.n = 0;                                              // every chain, or (from the flags of the last sync point) the active ones, compacted
  if (n_active < sp->R.chains && sp->R.chains <= DN_ACT_MAX) {
    for (int c = 0; c < sp->R.chains; c++) if (sp->h_active[c]) act.idx[act.n++] = c;
    n_active = act.n;
  }
  dense_launch_shape(sp->dn, n_active);
  sp->dn.count_passes = timed ? 1 : 0;     // potus_dense_timing: bytes and milliseconds over the same set of passes
  // (launches of the matrix pass: two for three right-hand sides per chain; pooled: one per DNP_RMAX right-hand sides)
  const int nact_launch = act.n ? act.n : sp->R.chains;   // (an empty list means everybody: dense_symv_launch)
  sp->mv_launches_pending = !timed ? 0 : sp->dn.pooled ? (nact_launch + DNP_RMAX / nrhs - 1) / (DNP_RMAX / nrhs) : (nrhs == 3 ? 2 : 1);
  // (the bytes the passes stream are counted on the device, DnParams::act_passes: the host's flags may be a few rounds old)
  HIP_TRY(hipEventRecord(e0 ? e0 : sp->mv0, sp->stream));
  dense_symv_launch(sp->stream, sp->dn, act, nrhs);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e1 ? e1 : sp->mv1, sp->stream));
  return 0;
}
// sync point of a round: the activity flags come to the host; returns the number of chains still active
int dense_sync(Sampler *sp, int *n_active, bool timed_matvec) {
  HIP_TRY(hipMemcpyAsync(sp->h_active, sp->dn.active, (size_t)sp->R.chains * sizeof(int), hipMemcpyDeviceToHost, sp->stream));
  HIP_TRY(hipStreamSynchronize(sp->stream));
  int n = 0;
  for (int c = 0; c < sp->R.chains; c++) n += sp->h_active[c] != 0;
  if (timed_matvec) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, sp->mv0, sp->mv1));
    sp->mv_ms += ms; sp->mv_calls += sp->mv_launches_pending;
  }
  *n_active = n;
  return 0;
}
// p ~ N(0, M): standard normals, then L^T p = u (nothing to solve while the metric is the unit matrix)
int dense_sample_p(Sampler *sp, unsigned iter, unsigned purpose) {
  hipLaunchKernelGGL(k_dn_normals, dn_grid(sp), dim3(256), 0, sp->stream, sp->dn, (const RunParams *)sp->dR, iter, purpose);
  HIP_TRY(hipGetLastError());
  if (!sp->dn.identity) {
    const int nb = (sp->L.D + DN_NB - 1) / DN_NB;
    for (int b = nb - 1; b >= 0; b--) {
      hipLaunchKernelGGL(k_dn_trsv_diag, dim3(sp->R.chains), dim3(64), 0, sp->stream, sp->dn, b);
      if (b > 0) hipLaunchKernelGGL(k_dn_trsv_update, dim3((b * DN_NB + 255) / 256, sp->R.chains), dim3(256), 0, sp->stream, sp->dn, b);
    }
    HIP_TRY(hipGetLastError());
  }
  return 0;
}
// base_hmc::init_stepsize for every chain of the handle (the point QC, its log density in the chain scalars)
int dense_init_stepsize(Sampler *sp, unsigned iter) {
  int rc, n_active = 0;
  const int cg = (sp->R.chains + 63) / 64;
  hipLaunchKernelGGL(k_dn_eps_arm, dim3(cg), dim3(64), 0, sp->stream, sp->dn, (const RunParams *)sp->dR);
  if ((rc = dense_grad(sp))) return rc;                                  // hamiltonian.init at the current point: gradient -> GC
  if ((rc = dense_sync(sp, &n_active, false))) return rc;
  for (int attempt = 0; n_active > 0; attempt++) {
    if (attempt > 200) return fail(POTUS_ERR_STATE, "dense init_stepsize did not terminate");
    if ((rc = dense_sample_p(sp, iter, RNG_INIT_EPS))) return rc;
    hipLaunchKernelGGL(k_dn_eps_prekick, dn_grid(sp), dim3(256), 0, sp->stream, sp->dn, (const RunParams *)sp->dR);
    if ((rc = dense_matvec(sp, 2, n_active, nullptr, nullptr, false))) return rc;
    hipLaunchKernelGGL(k_dn_eps_mid, dim3(cg), dim3(64), 0, sp->stream, sp->dn, (const RunParams *)sp->dR);
    if ((rc = dense_grad(sp))) return rc;
    hipLaunchKernelGGL(k_dn_kick, dn_grid(sp), dim3(256), 0, sp->stream, sp->dn);
    if ((rc = dense_matvec(sp, 1, n_active, nullptr, nullptr, false))) return rc;
    hipLaunchKernelGGL(k_dn_eps_step, dim3(cg), dim3(64), 0, sp->stream, sp->dn, (const RunParams *)sp->dR);
    HIP_TRY(hipGetLastError());
    if ((rc = dense_sync(sp, &n_active, false))) return rc;
  }
  return 0;
}
// the rest of a window end once M^-1 stands (pooled: once the pooled M2 stands in P.A, n_total draws behind it): factor, new step size
static double dn_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int dense_window_finish(Sampler *sp, int n, unsigned iter, double n_total, double cov_ms) {
  DnParams &P = sp->dn;
  const int D = sp->L.D, chains = sp->R.chains;
  if (P.pooled) {
    const double ts = dn_now();
    hipLaunchKernelGGL(k_dn_pool_scale, dim3((unsigned)std::min((D + 255) / 256, 64), (unsigned)D), dim3(256), 0, sp->stream, P, n_total);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(sp->stream));
    cov_ms += dn_now() - ts;
  }
  const double tc = dn_now();
  dense_cholesky_launch(sp->stream, P, P.pooled ? 1 : chains);   // in place on the lower triangle; the upper one keeps M^-1 (pooled: in the factor's own buffer)
  HIP_TRY(hipGetLastError());
  int failed = 0;
  HIP_TRY(hipMemcpyAsync(&failed, P.fail, 4, hipMemcpyDeviceToHost, sp->stream));
  HIP_TRY(hipStreamSynchronize(sp->stream));
  if (failed) return fail(POTUS_ERR_STATE, "the adapted covariance of a chain is not positive definite (window of %d draws)", n);
  const double t2 = dn_now();
  P.identity = 0;
  int rc;
  if ((rc = dense_init_stepsize(sp, iter))) return rc;
  sp->we_cov_ms += cov_ms; sp->we_chol_ms += t2 - tc; sp->we_eps_ms += dn_now() - t2; sp->we_count += 1;
  hipLaunchKernelGGL(k_dn_window_done, dim3((chains + 63) / 64), dim3(64), 0, sp->stream, P, (const RunParams *)sp->dR, (int)iter);
  HIP_TRY(hipGetLastError());
  return 0;
}
// covar_adaptation at the end of a window of n draws: covariance -> M^-1, Cholesky factor, new step size
int dense_window_end(Sampler *sp, int n, unsigned iter) {
  DnParams &P = sp->dn;
  const int D = sp->L.D, chains = sp->R.chains, nb = (D + DN_NB - 1) / DN_NB;
  if (n < 2) return fail(POTUS_ERR_STATE, "adaptation window of %d draws", n);
  HIP_TRY(hipStreamSynchronize(sp->stream));
  const double t0 = dn_now();
  if (P.pooled) {   // the handle's mean and M2 over the draws of every chain (potus_dense_pool.hpp)
    hipLaunchKernelGGL(k_dn_pool_center, dim3((unsigned)std::min((D + 255) / 256, 64)), dim3(256), 0, sp->stream, P, n);
    hipLaunchKernelGGL(k_dn_pool_cov, dim3(nb, nb), dim3(256), 0, sp->stream, P, n);
  } else {
    hipLaunchKernelGGL(k_dn_center, dn_grid(sp), dim3(256), 0, sp->stream, P, n);
    hipLaunchKernelGGL(k_dn_cov, dim3(nb, nb, chains), dim3(256), 0, sp->stream, P, n);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(sp->stream));
  const double t1 = dn_now();
  if (sp->opts.pooled_metric == 2) {   // the host pools further (other handles, other ranks) and calls potus_dense_pool_finish
    sp->dn_pending = true; sp->dn_pending_n = n; sp->dn_pending_iter = iter; sp->dn_pending_cov_ms = t1 - t0;
    return 0;
  }
  return dense_window_finish(sp, n, iter, (double)chains * (double)n, t1 - t0);
}
// n_iter transitions of every chain of the handle
int dense_run(Sampler *sp, int n_iter) {
  DnParams &P = sp->dn;
  const int chains = sp->R.chains, total = sp->R.num_warmup + sp->R.num_samples, cg = (chains + 63) / 64;
  const int nw = sp->R.num_warmup, ib = sp->R.init_buffer, tb = sp->R.term_buffer;
  std::vector<ChainScalars> sc;
  int rc;
  if (sp->dn_pending) return fail(POTUS_ERR_STATE, "a pooled window end is pending (pooled_metric = 2): potus_dense_pool_finish comes before the next potus_run");
  if ((rc = read_scalars(sp, sc))) return rc;
  int it = sc[0].iter;
  for (int k = 0; k < n_iter && it < total && !sp->dn_pending; k++, it++) {
    int n_active = 0;
    hipLaunchKernelGGL(k_dn_arm, dim3(cg), dim3(64), 0, sp->stream, P, (const RunParams *)sp->dR, total);
    if ((rc = dense_sample_p(sp, (unsigned)it, RNG_MOMENTUM))) return rc;
    if ((rc = dense_grad(sp))) return rc;                                // hamiltonian.init: gradient at the current point -> GC
    hipLaunchKernelGGL(k_dn_begin, dn_grid(sp), dim3(256), 0, sp->stream, P, (const RunParams *)sp->dR);
    if ((rc = dense_matvec(sp, 3, chains))) return rc;
    hipLaunchKernelGGL(k_dn_step, dim3(chains), dim3(DN_THREADS), 0, sp->stream, P, (const RunParams *)sp->dR, (unsigned)it, (int)DN_MODE_BEGIN);
    HIP_TRY(hipGetLastError());
    if ((rc = dense_sync(sp, &n_active, true))) return rc;
    // Leaf rounds.  The host does not wait for a round before it queues the next: it keeps DN_AHEAD rounds in flight and looks
    // at the activity flags of the round that left the queue (they come back with an asynchronous copy behind every round;
    // flags only ever go from 1 to 0 inside a transition).  Chains that have finished sit rounds out on the device (every
    // kernel returns at once for them), so the rounds queued past the end of the transition cost a few empty launches --
    // against a host round trip per leaf with the GPU idle (13 % of the wall time at D = 15 098 before).
    {
      constexpr int R = Sampler::DN_AHEAD;
      long long q = 0;                                  // rounds queued in this transition
      auto retire = [&](long long r) -> int {           // wait for round r; its pass time; the flags as of then (or later)
        const int k = (int)(r % R);
        HIP_TRY(hipEventSynchronize(sp->rdone[k]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, sp->rv0[k], sp->rv1[k]));
        sp->mv_ms += ms; sp->mv_calls += sp->rv_launches[k];
        int n = 0;
        for (int c = 0; c < sp->R.chains; c++) n += sp->h_active[c] != 0;
        n_active = n;
        return 0;
      };
      long long retired = 0;
      while (n_active > 0) {
        const int k = (int)(q % R);
        if ((rc = dense_grad(sp))) return rc;
        hipLaunchKernelGGL(k_dn_kick, dn_grid(sp), dim3(256), 0, sp->stream, P);
        if ((rc = dense_matvec(sp, 2, n_active, sp->rv0[k], sp->rv1[k]))) return rc;
        sp->rv_launches[k] = sp->mv_launches_pending;
        hipLaunchKernelGGL(k_dn_step, dim3(chains), dim3(DN_THREADS), 0, sp->stream, P, (const RunParams *)sp->dR, (unsigned)it, (int)DN_MODE_LEAF);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(sp->h_active, sp->dn.active, (size_t)sp->R.chains * sizeof(int), hipMemcpyDeviceToHost, sp->stream));
        HIP_TRY(hipEventRecord(sp->rdone[k], sp->stream));
        sp->dn_rounds += 1;
        q++;
        if (q - retired >= R) { if ((rc = retire(retired))) return rc; retired++; }
      }
      while (retired < q) { if ((rc = retire(retired))) return rc; retired++; }
    }
    // adapt_dense_e_nuts::transition: the warm-up schedule is the same for every chain, so the host keeps its own copy
    int flags = 0;
    const bool warm = it < nw;
    if (warm && nw >= 20) {
      const int wc = sp->dn_win_counter;
      if (wc >= ib && wc < nw - tb && wc != nw) flags |= 1;
      if (wc == sp->dn_win_next && wc != nw) flags |= 2;
    }
    hipLaunchKernelGGL(k_dn_end, dim3(chains), dim3(DN_THREADS), 0, sp->stream, P, (const RunParams *)sp->dR, it, flags, sp->dn_wf_n, total);
    HIP_TRY(hipGetLastError());
    if (flags & 1) sp->dn_wf_n += 1;
    if (flags & 2) {
      if ((rc = dense_window_end(sp, sp->dn_wf_n, (unsigned)it))) return rc;
      const int last = nw - tb - 1;                                       // windowed_adaptation::compute_next_window
      if (sp->dn_win_next != last) {
        sp->dn_win_size *= 2;
        sp->dn_win_next = sp->dn_win_counter + sp->dn_win_size;
        if (sp->dn_win_next != last && sp->dn_win_next + 2 * sp->dn_win_size >= nw - tb) sp->dn_win_next = last;
      }
      sp->dn_wf_n = 0;
    }
    if (warm && nw >= 20) sp->dn_win_counter += 1;
  }
  HIP_TRY(hipStreamSynchronize(sp->stream));
  return 0;
}
// potus_init of a dense-metric handle: k_init / k_cl_init have found the initial points and -- the metric being the
// unit matrix, for which the diagonal and the dense sampler coincide -- the initial step sizes; the points move over
int dense_import_init(Sampler *sp) {
  if (!sp->dn.identity || sp->dn_win_counter != 0) {
    // a second potus_init: back to the unit metric and the start of the window schedule (the upper triangle is what the matrix
    // pass reads, so the whole matrix goes, not only the flag)
    DnParams &P = sp->dn;
    HIP_TRY(hipMemsetAsync(P.A, 0, (size_t)(P.pooled ? 1 : P.chains) * P.D * P.LD * 8, sp->stream));
    if (P.pooled) HIP_TRY(hipMemsetAsync(P.Lf, 0, (size_t)P.D * P.LD * 8, sp->stream));
    if (P.pooled && P.f32) HIP_TRY(hipMemsetAsync(P.A32, 0, (size_t)P.D * P.LD * 4, sp->stream));
    HIP_TRY(hipMemsetAsync(P.fail, 0, 4, sp->stream));
    hipLaunchKernelGGL(k_dn_identity, dim3((P.D + 255) / 256, P.pooled ? 1 : P.chains), dim3(256), 0, sp->stream, P);
    HIP_TRY(hipGetLastError());
    P.identity = 1;
  }
  sp->dn_pending = false;
  sp->dn_win_counter = 0; sp->dn_win_size = sp->R.window; sp->dn_win_next = sp->R.init_buffer + sp->R.window - 1; sp->dn_wf_n = 0;
  hipLaunchKernelGGL(k_dn_import_q, dim3(64, sp->R.chains), dim3(256), 0, sp->stream, (const RunParams *)sp->dR, sp->dn,
                     sp->K > 1 ? sp->CL.perm : (const int *)nullptr, sp->K > 1 ? sp->CL.Dint : 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(sp->stream));
  return 0;
}

} // namespace

// ======================================================================== C ABI
extern "C" {

const char *potus_version(void) { return "potus_hmc 0.5 (gfx950)"; }   // 0.5: potus_opts.pooled_metric (round 6);   // 0.4: potus_opts.metric_storage (round 3), potus_diagnostics* (round 4)

int potus_last_error(char *buf, int len) {
  if (buf && len > 0) { std::snprintf(buf, (size_t)len, "%s", g_err.c_str()); }
  return (int)g_err.size();
}

void potus_default_opts(potus_opts *o) {
  std::memset(o, 0, sizeof *o);
  o->chains = 4; o->chain_id_offset = 0; o->num_warmup = 1000; o->num_samples = 1000; o->max_depth = 10;
  o->init_buffer = 75; o->term_buffer = 50; o->window = 25;
  o->delta = 0.8; o->gamma = 0.05; o->kappa = 0.75; o->t0 = 10; o->stepsize = 1.0; o->init_radius = 2.0;
  o->seed = 1843; o->device = 0; o->save_warmup = 0; o->cus_per_chain = 0; o->metric = POTUS_METRIC_DIAG; o->twin = -1; o->pooled_metric = 0; o->reserved_ = 0; o->metric_storage = POTUS_STORAGE_F64;
}

int potus_num_params(const potus_data *d, int *D) {
  if (!d || !D) return fail(POTUS_ERR_ARG, "null argument");
  *D = make_layout(d).D;
  return 0;
}
int potus_num_columns(const potus_data *d, int *n_cols) {
  if (!d || !n_cols) return fail(POTUS_ERR_ARG, "null argument");
  *n_cols = make_layout(d).ncols;
  return 0;
}

int potus_column_name(const potus_data *d, int col, char *buf, int len) {
  if (!d || !buf || len <= 0) return fail(POTUS_ERR_ARG, "null argument");
  static const char *samp[] = {"lp__", "accept_stat__", "stepsize__", "treedepth__", "n_leapfrog__", "divergent__", "energy__"};
  if (col < 0) return fail(POTUS_ERR_ARG, "negative column");
  if (col < POTUS_N_SAMPLER_COLS) { std::snprintf(buf, (size_t)len, "%s", samp[col]); return 0; }
  const bool full = d->variant == POTUS_VARIANT_FULL;
  struct Blk { const char *name; int r, c; }; // r x c column-major; c = 0: vector, r = 0: scalar
  std::vector<Blk> b = {{"raw_mu_b_T", d->S, 0}, {"raw_mu_b", d->S, d->T}, {"raw_mu_c", d->P, 0}};
  if (full) { b.push_back({"raw_mu_m", d->M, 0}); b.push_back({"raw_mu_pop", d->Pop, 0}); b.push_back({"mu_e_bias", 0, 0});
              b.push_back({"rho_e_bias", 0, 0}); b.push_back({"raw_e_bias", d->T, 0}); }
  b.push_back({"raw_measure_noise_national", d->N_national_polls, 0}); b.push_back({"raw_measure_noise_state", d->N_state_polls, 0});
  b.push_back({"raw_polling_bias", d->S, 0});
  b.push_back({"mu_b", d->S, d->T}); b.push_back({"mu_c", d->P, 0});
  if (full) { b.push_back({"mu_m", d->M, 0}); b.push_back({"mu_pop", d->Pop, 0}); b.push_back({"e_bias", d->T, 0}); }
  b.push_back({"polling_bias", d->S, 0}); b.push_back({"national_mu_b_average", d->T, 0}); b.push_back({"national_polling_bias_average", 0, 0});
  if (full) b.push_back({"sigma_rho", 0, 0});
  b.push_back({"logit_pi_democrat_state", d->N_state_polls, 0}); b.push_back({"logit_pi_democrat_national", d->N_national_polls, 0});
  b.push_back({"predicted_score", d->T, d->S});
  int c = col - POTUS_N_SAMPLER_COLS;
  for (const Blk &k : b) {
    const int n = k.r == 0 ? 1 : (k.c == 0 ? k.r : k.r * k.c);
    if (c < n) {
      if (k.r == 0) std::snprintf(buf, (size_t)len, "%s", k.name);
      else if (k.c == 0) std::snprintf(buf, (size_t)len, "%s.%d", k.name, c + 1);
      else std::snprintf(buf, (size_t)len, "%s.%d.%d", k.name, c % k.r + 1, c / k.r + 1);
      return 0;
    }
    c -= n;
  }
  return fail(POTUS_ERR_ARG, "column %d out of range", col);
}

// ---------------------------------------------------------------- how many workgroups work for a chain (pure functions)
// The first plan for cus_per_chain (potus_create may still double it when a member's polls do not fit its LDS):
// 0 = as many compute units per chain as fit the device, out of {16, 8, 4, 1}; 9-12 chains on 256 compute units get 14 / 12 /
// 11 / 10 so that a second cluster per chain fits (*lowered_for_twin = 1: back to 16 if the second cluster is not taken).
int potus_plan_cus_per_chain(int chains, int T, int n_cus, int cus_per_chain, int twin, int metric, int one_workgroup_ok, int *K_out, int *lowered_for_twin) {
  if (!K_out || !lowered_for_twin) return fail(POTUS_ERR_ARG, "potus_plan_cus_per_chain: null output");
  if (chains < 1 || n_cus < 1 || T < 1) return fail(POTUS_ERR_ARG, "potus_plan_cus_per_chain: chains, T and the number of compute units must be positive");
  int K = cus_per_chain;
  *lowered_for_twin = 0;
  if (K < 0 || K > CL_MAXK) return fail(POTUS_ERR_ARG, "cus_per_chain must be in [0,%d]", CL_MAXK);
  if (K == 0) {
    K = chains * 16 <= n_cus ? 16 : chains * 8 <= n_cus ? 8 : chains * 4 <= n_cus ? 4 : 1;
    if (K == 4 && T > 4 * CL_MAXDAYS) K = 1;      // four members hold up to 256 days
    // 9-12 chains: two clusters of 14, 12, 11 or 10 per chain (17-20 us per leapfrog on the 2016 posterior) beat one cluster
    // of 16 (21 us) -- as long as the members keep few enough days for the lighter build of the pass (4 days per wave)
    if (K == 16 && twin != 0 && metric != POTUS_METRIC_DENSE && chains * 32 > n_cus) {
      const int k2 = n_cus / (2 * chains);
      if (k2 >= 10 && (T + k2 - 1) / k2 <= CL_DW4_MAXAVG) { K = k2; *lowered_for_twin = 1; }
    }
    // models beyond the one-workgroup kernels (T > 256, > 2048 polls) need a cluster; long campaigns need more members
    if (K == 1 && !one_workgroup_ok) K = 8;
    while (K > 1 && T > K * CL_MAXDAYS && 2 * K <= CL_MAXK) K *= 2;
    if (K > 1 && chains * K > n_cus)
      return fail(POTUS_ERR_UNSUPPORTED, "%s, and %d chains x %d compute units do not fit the %d of the device",
                  one_workgroup_ok ? "T needs more members per chain" : "the model needs a cluster per chain", chains, K, n_cus);
  } else if (K == 1 && !one_workgroup_ok)
    return fail(POTUS_ERR_UNSUPPORTED, "the model is beyond the one-workgroup kernels; use cus_per_chain = 0 or >= 8");
  if (K > 1 && chains * K > n_cus) return fail(POTUS_ERR_ARG, "chains * cus_per_chain = %d exceeds the %d compute units of the device", chains * K, n_cus);
  *K_out = K;
  return 0;
}

// One or two clusters (K > 1) / workgroups (K = 1) per chain, one per end of the trajectory: two when asked for (twin = 1) or when
// the library chooses the size as well (twin < 0 and cus_per_chain = 0), the metric is diagonal and 2 * chains * K workgroups are
// resident together (resident_per_cu: workgroups of the kernel a compute unit holds).  twin = 1 that does not fit is an error.
int potus_plan_sides(int chains, int K, int n_cus, int resident_per_cu, int cus_per_chain, int twin, int metric, int *sides) {
  if (!sides) return fail(POTUS_ERR_ARG, "potus_plan_sides: null output");
  if (twin < -1 || twin > 1) return fail(POTUS_ERR_ARG, "twin must be -1 (library's choice), 0 or 1");
  const bool fits = metric != POTUS_METRIC_DENSE && resident_per_cu >= 1 &&
                    2ll * chains * K <= (long long)n_cus * (K == 1 ? resident_per_cu : 1);
  if (twin == 1 && !fits) {
    if (K == 1)
      return fail(POTUS_ERR_ARG, "twin = 1: two workgroups per chain need %d resident workgroups (the device holds %d) and the diagonal metric",
                  2 * chains, n_cus * std::max(resident_per_cu, 0));
    return fail(POTUS_ERR_ARG, "twin = 1: two clusters of %d per chain need %d compute units (the device has %d) and the diagonal metric", K, chains * 2 * K, n_cus);
  }
  *sides = ((twin == 1 || (twin < 0 && cus_per_chain == 0)) && fits) ? 2 : 1;
  return 0;
}

int potus_create(const potus_data *d, const potus_opts *o, int *handle) {
  if (!o || !handle) return fail(POTUS_ERR_ARG, "null argument");
  int rc = validate(d);
  if (rc) return rc;
  if (o->chains < 1) return fail(POTUS_ERR_ARG, "chains must be >= 1");
  if (o->max_depth < 1 || o->max_depth > PT_MAXD) return fail(POTUS_ERR_ARG, "max_depth must be in [1,%d]", PT_MAXD);
  if (o->num_warmup < 0 || o->num_samples < 0) return fail(POTUS_ERR_ARG, "negative iteration counts");
  if (o->metric != POTUS_METRIC_DIAG && o->metric != POTUS_METRIC_DENSE) return fail(POTUS_ERR_ARG, "metric must be POTUS_METRIC_DIAG or POTUS_METRIC_DENSE");
  if (o->twin < -1 || o->twin > 1) return fail(POTUS_ERR_ARG, "twin must be -1 (library's choice), 0 or 1");
  if (o->metric_storage != POTUS_STORAGE_F64 && o->metric_storage != POTUS_STORAGE_F32) return fail(POTUS_ERR_ARG, "metric_storage must be POTUS_STORAGE_F64 or POTUS_STORAGE_F32");
  if (o->pooled_metric < 0 || o->pooled_metric > 2) return fail(POTUS_ERR_ARG, "pooled_metric must be 0, 1 or 2");
  if (o->pooled_metric && o->metric != POTUS_METRIC_DENSE) return fail(POTUS_ERR_ARG, "pooled_metric applies to the dense metric (metric = POTUS_METRIC_DENSE)");
  if (o->metric_storage == POTUS_STORAGE_F32 && o->metric != POTUS_METRIC_DENSE) return fail(POTUS_ERR_ARG, "metric_storage = f32 applies to the dense metric only");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(POTUS_ERR_DEVICE, "no HIP device: libpotus_hmc needs an MI355X (gfx950)");
  if (o->device < 0 || o->device >= ndev) return fail(POTUS_ERR_DEVICE, "device %d out of range (have %d)", o->device, ndev);
  HIP_TRY(hipSetDevice(o->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, o->device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(POTUS_ERR_DEVICE, "device is %s; this library is built for gfx950 only", prop.gcnArchName);

  Sampler *sp = new Sampler();
  sp->device = o->device; sp->opts = *o; sp->L = make_layout(d);
  sp->data_hash = hash_data(d);
  auto bail = [&](int code) {
    for (void *p : sp->allocs) (void)hipFree(p);
    if (sp->ev0) (void)hipEventDestroy(sp->ev0);
    if (sp->ev1) (void)hipEventDestroy(sp->ev1);
    if (sp->mv0) (void)hipEventDestroy(sp->mv0);
    if (sp->mv1) (void)hipEventDestroy(sp->mv1);
    for (int k = 0; k < Sampler::DN_AHEAD; k++) { if (sp->rv0[k]) (void)hipEventDestroy(sp->rv0[k]); if (sp->rv1[k]) (void)hipEventDestroy(sp->rv1[k]); if (sp->rdone[k]) (void)hipEventDestroy(sp->rdone[k]); }
    if (sp->h_active) (void)hipHostFree(sp->h_active);
    if (sp->stream) (void)hipStreamDestroy(sp->stream);
    delete sp;
    return code;
  };
  if ((rc = build_model(sp, d))) return bail(rc);
  if (sp->k1_unsupported.empty() && (rc = set_lds_attr(sp))) return bail(rc);
  {
    // workgroups (CUs) per chain and sides: potus_plan_cus_per_chain / potus_plan_sides (pure functions, below; tested without a GPU)
    const int ncu = prop.multiProcessorCount;
    int K = 0, lowered = 0;
    if ((rc = potus_plan_cus_per_chain(o->chains, d->T, ncu, o->cus_per_chain, o->twin, o->metric, sp->k1_unsupported.empty() ? 1 : 0, &K, &lowered))) {
      if (rc == POTUS_ERR_UNSUPPORTED && !sp->k1_unsupported.empty()) {
        const std::string first = g_err;
        return bail(fail(rc, "%s (%s)", sp->k1_unsupported.c_str(), first.c_str()));
      }
      return bail(rc);
    }
    const bool k_lowered_for_twin = lowered != 0;
    if (K == 1) {
      // Two workgroups per chain, one per end of the trajectory (potus_nuts_twin.hpp): both sides of every chain wait for each
      // other, so the whole grid has to be resident at once
      int per_cu = 0, sides = 1;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_run_twin), PT_THREADS, sp->lds_bytes) != hipSuccess) per_cu = 0;
      if ((rc = potus_plan_sides(o->chains, 1, ncu, per_cu, o->cus_per_chain, o->twin, o->metric, &sides))) return bail(rc);
      sp->twin = sides == 2;
    }
    if (K > 1) {
      rc = build_cluster(sp, d, K);
      // chosen by the library: a member whose polls do not fit its LDS gets half of them with twice the members
      while (rc == POTUS_ERR_UNSUPPORTED && o->cus_per_chain == 0 && 2 * K <= CL_MAXK && o->chains * 2 * K <= ncu) { K *= 2; rc = build_cluster(sp, d, K); }
      if (rc) return bail(rc);
      // the members of a cluster wait for each other: the whole grid has to be resident at once
      int per_cu = 0;
      const void *kfn = sp->cl_dw == 16 ? reinterpret_cast<const void *>(k_cl_run<16, false>) : sp->cl_dw == 17 ? reinterpret_cast<const void *>(k_cl_run<17, false>) : sp->cl_dw == 4 ? reinterpret_cast<const void *>(k_cl_run<4, false>)
                        : sp->cl_dw == 12 ? reinterpret_cast<const void *>(k_cl_run<12, false>) : reinterpret_cast<const void *>(k_cl_run<8, false>);
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, PT_THREADS, sp->cl_lds_bytes) != hipSuccess || per_cu < 1)
        return bail(fail(POTUS_ERR_DEVICE, "cluster kernel cannot be resident on this device (occupancy query: %d workgroups per compute unit with %zu bytes of LDS)",
                         per_cu, sp->cl_lds_bytes));
      if ((long long)o->chains * K > (long long)ncu * per_cu)
        return bail(fail(POTUS_ERR_ARG, "chains * cus_per_chain = %d workgroups cannot be resident together (%d compute units x %d)", o->chains * K, ncu, per_cu));
      // Two clusters per chain, one per end of the trajectory (potus_cluster.hpp, twin mode), when the compute units are there:
      // asked for (twin = 1), or chosen with the cluster size (twin < 0 and cus_per_chain = 0).  Diagonal metric only.
      int sides_plan = 1;
      if ((rc = potus_plan_sides(o->chains, K, ncu, 1, o->cus_per_chain, o->twin, o->metric, &sides_plan))) return bail(rc);
      const bool twin_fits = sides_plan == 2;      // asked for or chosen, and the compute units are there
      const void *kft = sp->cl_dw == 16 ? reinterpret_cast<const void *>(k_cl_run<16, true>) : sp->cl_dw == 17 ? reinterpret_cast<const void *>(k_cl_run<17, true>) : sp->cl_dw == 4 ? reinterpret_cast<const void *>(k_cl_run<4, true>)
                        : sp->cl_dw == 12 ? reinterpret_cast<const void *>(k_cl_run<12, true>) : reinterpret_cast<const void *>(k_cl_run<8, true>);
      int per_cu_t = 0;
      if (twin_fits && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_t, kft, PT_THREADS, sp->cl_lds_bytes) == hipSuccess && per_cu_t >= 1)
        sp->twin = 1;
      else if (o->twin == 1) return bail(fail(POTUS_ERR_DEVICE, "twin = 1: the twin kernel cannot be resident on this device"));
      if (!sp->twin && k_lowered_for_twin) {   // the smaller clusters were chosen for the sake of the second one: without it, back to 16
        if ((rc = build_cluster(sp, d, 16))) return bail(rc);
      }
    }
  }
  if (hipStreamCreateWithFlags(&sp->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&sp->ev0) != hipSuccess ||
      hipEventCreate(&sp->ev1) != hipSuccess)
    return bail(fail(POTUS_ERR_DEVICE, "stream/event creation failed"));

  RunParams &R = sp->R;
  R.chains = o->chains; R.chain_id_offset = o->chain_id_offset; R.num_warmup = o->num_warmup; R.num_samples = o->num_samples;
  R.max_depth = o->max_depth; R.save_warmup = o->save_warmup;
  // windowed_adaptation::set_window_params
  int ib = o->init_buffer, tb = o->term_buffer, bw = o->window;
  if (o->num_warmup >= 20 && ib + bw + tb > o->num_warmup) { ib = (int)(0.15 * o->num_warmup); tb = (int)(0.1 * o->num_warmup); bw = o->num_warmup - (ib + tb); }
  R.init_buffer = ib; R.term_buffer = tb; R.window = bw;
  R.delta = o->delta; R.gamma = o->gamma; R.kappa = o->kappa; R.t0 = o->t0; R.stepsize = o->stepsize; R.init_radius = o->init_radius;
  R.seed_lo = (unsigned)(o->seed & 0xFFFFFFFFu); R.seed_hi = (unsigned)(o->seed >> 32);
  R.Dpad = sp->K > 1 ? ((sp->CL.Dint + 15) & ~15) : ((sp->L.D + 7) & ~7);
  R.row = POTUS_N_SAMPLER_COLS + sp->L.D;
  R.n_save_max = o->num_samples + (o->save_warmup ? o->num_warmup : 0);
  void *p = nullptr;
  const size_t nblk = (size_t)o->chains * sp->sides();      // twin mode: a block per side
  const size_t state_bytes = nblk * V_COUNT * R.Dpad * sizeof(double);
  if (hipMalloc(&p, state_bytes) != hipSuccess) return bail(fail(POTUS_ERR_DEVICE, "hipMalloc(%zu) for chain state failed", state_bytes));
  sp->allocs.push_back(p); R.state = (double *)p;
  (void)hipMemset(p, 0, state_bytes);
  if (hipMalloc(&p, sizeof(ChainScalars) * nblk * sp->K) != hipSuccess) return bail(fail(POTUS_ERR_DEVICE, "hipMalloc for chain scalars failed"));
  sp->allocs.push_back(p); R.scal = (ChainScalars *)p;
  (void)hipMemset(p, 0, sizeof(ChainScalars) * nblk * sp->K);
  R.K = sp->K; R.xbuf = nullptr; R.xcnt = nullptr; R.twin = sp->twin; R.twbuf = nullptr;
  R.debug_drop_member = getenv("POTUS_DEBUG_DROP_MEMBER") ? atoi(getenv("POTUS_DEBUG_DROP_MEMBER")) : 0;
  if (sp->coop()) {
    const size_t xb = nblk * (4 * (size_t)sp->K * sp->CL.XW + 8) * 16;   // per chain (and side): 4 exchange slots + a line for the watchdog word
    if (hipMalloc(&p, xb) != hipSuccess) return bail(fail(POTUS_ERR_DEVICE, "hipMalloc for exchange buffers failed"));
    sp->allocs.push_back(p); R.xbuf = (double *)p;
    (void)hipMemset(p, 0, xb);
    if (hipMalloc(&p, (size_t)o->chains * 64 * sizeof(unsigned)) != hipSuccess) return bail(fail(POTUS_ERR_DEVICE, "hipMalloc for exchange counters failed"));
    sp->allocs.push_back(p); R.xcnt = (unsigned *)p;
    (void)hipMemset(p, 0, (size_t)o->chains * 64 * sizeof(unsigned));
    if (sp->twin) {
      const size_t tb = (size_t)o->chains * TWB_WORDS * 16;
      if (hipMalloc(&p, tb) != hipSuccess) return bail(fail(POTUS_ERR_DEVICE, "hipMalloc for the twin mailboxes failed"));
      sp->allocs.push_back(p); R.twbuf = (double *)p;
      (void)hipMemset(p, 0, tb);
    }
  }
  const size_t draw_bytes = std::max<size_t>((size_t)o->chains * R.n_save_max * R.row, 1) * sizeof(double);
  if (hipMalloc(&p, draw_bytes) != hipSuccess) return bail(fail(POTUS_ERR_DEVICE, "hipMalloc(%zu) for draws failed", draw_bytes));
  sp->allocs.push_back(p); R.draws = (double *)p;
  (void)hipMemset(p, 0, draw_bytes);

#ifdef POTUS_PROF
  if (hipMalloc(&p, sizeof(double) * PT_NPROF * nblk * sp->K) != hipSuccess) return bail(fail(POTUS_ERR_DEVICE, "hipMalloc for profile failed"));
  sp->allocs.push_back(p); R.prof = (double *)p;
  (void)hipMemset(p, 0, sizeof(double) * PT_NPROF * nblk * sp->K);
#else
  R.prof = nullptr;
#endif
  if (hipMalloc(&p, sizeof(RunParams)) != hipSuccess) return bail(fail(POTUS_ERR_DEVICE, "hipMalloc for run parameters failed"));
  sp->allocs.push_back(p); sp->dR = (RunParams *)p;
  if (hipMemcpy(p, &R, sizeof(RunParams), hipMemcpyHostToDevice) != hipSuccess) return bail(fail(POTUS_ERR_DEVICE, "upload of run parameters failed"));
  if (o->metric == POTUS_METRIC_DENSE) {
    sp->dense = true;
    if ((rc = dense_alloc(sp))) return bail(rc);
  }
  // the buffers above were cleared on the null stream, the sampler launches on its own non-blocking stream
  if (hipDeviceSynchronize() != hipSuccess) return bail(fail(POTUS_ERR_DEVICE, "device synchronisation after the allocations failed"));
  std::lock_guard<std::mutex> lk(g_mu);
  size_t slot = 0;
  while (slot < g_handles.size() && g_handles[slot]) slot++;     // handles of destroyed samplers are reused
  if (slot == g_handles.size()) g_handles.push_back(sp); else g_handles[slot] = sp;
  *handle = (int)slot;
  return 0;
}

int potus_destroy(int handle) {
  Sampler *sp = get(handle);
  if (!sp) return fail(POTUS_ERR_STATE, "bad handle");
  (void)hipSetDevice(sp->device);
  (void)hipStreamSynchronize(sp->stream);
  for (void *p : sp->allocs) (void)hipFree(p);
  (void)hipEventDestroy(sp->ev0); (void)hipEventDestroy(sp->ev1); (void)hipStreamDestroy(sp->stream);
  if (sp->mv0) (void)hipEventDestroy(sp->mv0);
  if (sp->mv1) (void)hipEventDestroy(sp->mv1);
  for (int k = 0; k < Sampler::DN_AHEAD; k++) { if (sp->rv0[k]) (void)hipEventDestroy(sp->rv0[k]); if (sp->rv1[k]) (void)hipEventDestroy(sp->rv1[k]); if (sp->rdone[k]) (void)hipEventDestroy(sp->rdone[k]); }
  if (sp->h_active) (void)hipHostFree(sp->h_active);
  { std::lock_guard<std::mutex> lk(g_mu); g_handles[handle] = nullptr; }
  delete sp;
  return 0;
}

int potus_cus_per_chain(int handle, int *k) {
  Sampler *sp = get(handle);
  if (!sp || !k) return fail(POTUS_ERR_STATE, "bad handle");
  *k = sp->K;
  return 0;
}

int potus_clusters_per_chain(int handle, int *n) {
  Sampler *sp = get(handle);
  if (!sp || !n) return fail(POTUS_ERR_STATE, "bad handle");
  *n = sp->sides();
  return 0;
}

int potus_log_prob_grad(int handle, const double *q, int n, double *lp, double *grad) {
  Sampler *sp = get(handle);
  if (!sp) return fail(POTUS_ERR_STATE, "bad handle");
  if (n < 0 || (n > 0 && (!q || !lp || !grad))) return fail(POTUS_ERR_ARG, "null argument");
  if (n == 0) return 0;
  DeviceLocks lock(sp->device);
  HIP_TRY(hipSetDevice(sp->device));
  const size_t D = sp->L.D;
  DevBufs tmp;
  double *dq = nullptr, *dlp = nullptr, *dg = nullptr;
  HIP_TRY(tmp.alloc(&dq, n * D * 8)); HIP_TRY(tmp.alloc(&dg, n * D * 8)); HIP_TRY(tmp.alloc(&dlp, (size_t)n * 8));
  HIP_TRY(hipMemcpyAsync(dq, q, n * D * 8, hipMemcpyHostToDevice, sp->stream));
  double *dscr = nullptr;
  if (sp->K > 1) {   // the cluster's own pass, on one cluster
    HIP_TRY(tmp.alloc(&dscr, 2 * (size_t)sp->R.Dpad * 8));
    const unsigned lid = ++sp->launch_id;
#define POTUS_CALL(TAG) hipLaunchKernelGGL(k_cl_logprob_grad<TAG>, dim3(sp->K), dim3(PT_THREADS), sp->cl_lds_bytes, sp->stream, (const DevModel *)sp->dM, \
                                           (const ClModel *)sp->dCL, (const RunParams *)sp->dR, (const double *)dq, dlp, dg, n, dscr, lid)
    CL_DISPATCH(sp->cl_dw, POTUS_CALL);
#undef POTUS_CALL
  } else {
    const int grid = std::min(n, 1024);
    hipLaunchKernelGGL(k_logprob_grad, dim3(grid), dim3(PT_THREADS), sp->lds_bytes, sp->stream, (const DevModel *)sp->dM, (const double *)dq, dlp, dg, n);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(lp, dlp, (size_t)n * 8, hipMemcpyDeviceToHost, sp->stream));
  HIP_TRY(hipMemcpyAsync(grad, dg, n * D * 8, hipMemcpyDeviceToHost, sp->stream));
  HIP_TRY(hipStreamSynchronize(sp->stream));
  return 0;
}

int potus_init(int handle, const double *q0) {
  Sampler *sp = get(handle);
  if (!sp) return fail(POTUS_ERR_STATE, "bad handle");
  DeviceLocks lock(sp->device);
  HIP_TRY(hipSetDevice(sp->device));
  DevBufs tmp;
  double *dq0 = nullptr;
  const size_t bytes = (size_t)sp->R.chains * sp->L.D * 8;
  if (q0) { HIP_TRY(tmp.alloc(&dq0, bytes)); HIP_TRY(hipMemcpyAsync(dq0, q0, bytes, hipMemcpyHostToDevice, sp->stream)); }
  if (sp->K > 1) {
    const unsigned lid = ++sp->launch_id;
#define POTUS_CALL(TAG) hipLaunchKernelGGL(k_cl_init<TAG>, dim3(sp->R.chains * sp->K * sp->sides()), dim3(PT_THREADS), sp->cl_lds_bytes, sp->stream, (const DevModel *)sp->dM, \
                                           (const ClModel *)sp->dCL, (const RunParams *)sp->dR, (const double *)dq0, lid)
    CL_DISPATCH(sp->cl_dw, POTUS_CALL);
#undef POTUS_CALL
  } else
    hipLaunchKernelGGL(k_init, dim3(sp->R.chains * sp->sides()), dim3(PT_THREADS), sp->lds_bytes, sp->stream, (const DevModel *)sp->dM, (const RunParams *)sp->dR, (const double *)dq0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(sp->stream));
  { const int rc_ = check_chains(sp); if (rc_) return rc_; }
  if (sp->dense) { const int rc_ = dense_import_init(sp); if (rc_) return rc_; }
  sp->inited = true;
  return 0;
}

namespace {
// potus_run in two halves, so that several samplers can be in flight at once
struct RunTicket { Sampler *sp; int handle; long long before; int it0; };
int run_launch(RunTicket &t, int n_iter) {
  Sampler *sp = t.sp;
  HIP_TRY(hipSetDevice(sp->device));
  potus_total_leapfrogs(t.handle, &t.before);
  potus_iterations_done(t.handle, &t.it0);
  HIP_TRY(hipEventRecord(sp->ev0, sp->stream));
  if (sp->dense) {   // a sequence of chip-wide launches driven from here (potus_dense.hpp); it ends synchronised
    const int rc = dense_run(sp, n_iter);
    if (rc) return rc;
  } else if (sp->K > 1) {
    const unsigned lid = ++sp->launch_id;
    const dim3 grid((unsigned)(sp->R.chains * sp->K * sp->sides()));
    const DevModel *dM = sp->dM; const ClModel *dCL = sp->dCL; const RunParams *dR = sp->dR;
#define POTUS_CALL(TAG) do { if (sp->twin) hipLaunchKernelGGL((k_cl_run<TAG, true>), grid, dim3(PT_THREADS), sp->cl_lds_bytes, sp->stream, dM, dCL, dR, n_iter, lid); \
                             else hipLaunchKernelGGL((k_cl_run<TAG, false>), grid, dim3(PT_THREADS), sp->cl_lds_bytes, sp->stream, dM, dCL, dR, n_iter, lid); } while (0)
    CL_DISPATCH(sp->cl_dw, POTUS_CALL);
#undef POTUS_CALL
  } else if (sp->twin)
    hipLaunchKernelGGL(k_run_twin, dim3(2 * sp->R.chains), dim3(PT_THREADS), sp->lds_bytes, sp->stream, (const DevModel *)sp->dM, (const RunParams *)sp->dR, n_iter,
                       ++sp->launch_id);
  else
    hipLaunchKernelGGL(k_run, dim3(sp->R.chains), dim3(PT_THREADS), sp->lds_bytes, sp->stream, (const DevModel *)sp->dM, (const RunParams *)sp->dR, n_iter);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(sp->ev1, sp->stream));
  return 0;
}
int run_finish(RunTicket &t) {
  Sampler *sp = t.sp;
  HIP_TRY(hipSetDevice(sp->device));
  HIP_TRY(hipStreamSynchronize(sp->stream));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, sp->ev0, sp->ev1));
  long long after = 0;
  potus_total_leapfrogs(t.handle, &after);
  sp->last_ms = ms; sp->last_leapfrogs = after - t.before;
  int it1 = 0; potus_iterations_done(t.handle, &it1);
  // split elapsed time between warm-up and sampling in proportion to iterations (for the CSV footer)
  const int nw = sp->R.num_warmup;
  const int w_it = std::max(0, std::min(it1, nw) - std::min(t.it0, nw)), tot = std::max(1, it1 - t.it0);
  sp->warm_ms += ms * w_it / tot; sp->samp_ms += ms * (tot - w_it) / tot;
  return check_chains(sp);
}
} // namespace

int potus_run(int handle, int n_iter) {
  Sampler *sp = get(handle);
  if (!sp) return fail(POTUS_ERR_STATE, "bad handle");
  if (!sp->inited) return fail(POTUS_ERR_STATE, "potus_init must be called before potus_run");
  if (n_iter <= 0) return 0;
  // launches are serialised per process: a cluster launch must find its compute units free, also of the
  // workgroups of a one-workgroup-per-chain sampler
  DeviceLocks lock(sp->device);
  RunTicket t{sp, handle, 0, 0};
  int rc = run_launch(t, n_iter);
  if (rc) return rc;
  return run_finish(t);
}

// Several samplers at once (the three backtests of BASELINE configs[3] on one GPU, or the shards of one posterior on
// the GPUs of a node, driven from a single host thread such as R's): all launches of a group are issued before any is
// waited for.  The workgroups of a cluster launch wait for each other, so every cluster launch in flight on a device
// must be resident in full: a group takes cluster samplers only while their workgroups fit the compute units of their
// device, XCD by XCD (workgroups are dealt round-robin to the eight XCDs); the rest runs in the next group.
int potus_run_many(const int *handles, int n_handles, int n_iter) {
  if (!handles || n_handles < 0) return fail(POTUS_ERR_ARG, "potus_run_many: null handle list");
  if (n_iter <= 0 || n_handles == 0) return 0;
  std::vector<RunTicket> todo;
  for (int i = 0; i < n_handles; i++) {
    Sampler *sp = get(handles[i]);
    if (!sp) return fail(POTUS_ERR_STATE, "potus_run_many: bad handle %d", handles[i]);
    if (!sp->inited) return fail(POTUS_ERR_STATE, "potus_init must be called before potus_run_many");
    for (int j = 0; j < i; j++) if (handles[j] == handles[i]) return fail(POTUS_ERR_ARG, "potus_run_many: handle %d listed twice", handles[i]);
    todo.push_back(RunTicket{sp, handles[i], 0, 0});
  }
  std::vector<int> devs;
  for (const RunTicket &t : todo) devs.push_back(t.sp->device);
  DeviceLocks lock(devs);
  std::vector<char> done(todo.size(), 0);
  size_t n_done = 0;
  while (n_done < todo.size()) {
    std::vector<size_t> group;
    struct Use { int device, rows, plain; };   // XCD rows held by clusters; one-workgroup samplers in the group
    std::vector<Use> used;
    for (size_t i = 0; i < todo.size(); i++) {
      if (done[i]) continue;
      Sampler *sp = todo[i].sp;
      auto it = std::find_if(used.begin(), used.end(), [&](const Use &u) { return u.device == sp->device; });
      if (sp->dense) {   // chip-wide launches driven by the host: a group of its own
        if (!group.empty()) continue;
        group.push_back(i);
        break;
      }
      if (sp->coop()) {
        int ncu = 0;
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, sp->device));
        const int rows = (sp->R.chains * sp->K * sp->sides() + 7) / 8, cap = ncu / 8;
        if (it != used.end() && (it->plain > 0 || it->rows + rows > cap)) continue;   // next group (a sampler alone always fits: checked at create)
        if (it == used.end()) used.push_back({sp->device, rows, 0}); else it->rows += rows;
      } else {
        // a one-workgroup-per-chain launch may hold every compute unit for seconds: never beside a cluster
        if (it != used.end() && it->rows > 0) continue;
        if (it == used.end()) used.push_back({sp->device, 0, 1}); else it->plain += 1;
      }
      group.push_back(i);
    }
    int rc = 0;
    size_t launched = 0;
    for (; launched < group.size() && !rc; launched++) rc = run_launch(todo[group[launched]], n_iter);
    for (size_t k = 0; k < launched; k++) { const int r2 = run_finish(todo[group[k]]); if (!rc) rc = r2; }
    if (rc) return rc;
    for (size_t gi : group) { done[gi] = 1; n_done++; }
  }
  return 0;
}

int potus_iterations_done(int handle, int *n) {
  Sampler *sp = get(handle);
  if (!sp || !n) return fail(POTUS_ERR_STATE, "bad handle");
  HIP_TRY(hipSetDevice(sp->device));
  std::vector<ChainScalars> sc;
  { const int rc_ = read_scalars(sp, sc); if (rc_) return rc_; }
  int m = sc[0].iter;
  for (auto &s : sc) m = std::min(m, s.iter);
  *n = m;
  return 0;
}

int potus_total_leapfrogs(int handle, long long *n) {
  Sampler *sp = get(handle);
  if (!sp || !n) return fail(POTUS_ERR_STATE, "bad handle");
  HIP_TRY(hipSetDevice(sp->device));
  std::vector<ChainScalars> sc;
  { const int rc_ = read_scalars(sp, sc); if (rc_) return rc_; }
  long long t = 0;
  for (auto &s : sc) t += s.total_leapfrogs;
  *n = t;
  return 0;
}

int potus_twin_stats(int handle, long long *counted, long long *run_backward, long long *run_forward) {
  Sampler *sp = get(handle);
  if (!sp || !counted || !run_backward || !run_forward) return fail(POTUS_ERR_STATE, "bad handle");
  if (!sp->twin) return fail(POTUS_ERR_STATE, "potus_twin_stats: the handle runs one cluster per chain");
  HIP_TRY(hipSetDevice(sp->device));
  std::vector<ChainScalars> all((size_t)sp->R.chains * sp->K * 2);
  HIP_TRY(hipMemcpy(all.data(), sp->R.scal, sizeof(ChainScalars) * all.size(), hipMemcpyDeviceToHost));
  *counted = *run_backward = *run_forward = 0;
  for (int c = 0; c < sp->R.chains; c++) {
    *counted += all[(size_t)c * sp->K].total_leapfrogs;
    *run_backward += all[(size_t)c * sp->K].leaves_run;
    *run_forward += all[((size_t)sp->R.chains + c) * sp->K].leaves_run;
  }
  return 0;
}

int potus_chain_status(int handle, int *status, int *n_divergent) {
  Sampler *sp = get(handle);
  if (!sp) return fail(POTUS_ERR_STATE, "bad handle");
  HIP_TRY(hipSetDevice(sp->device));
  std::vector<ChainScalars> sc;
  { const int rc_ = read_scalars(sp, sc); if (rc_) return rc_; }
  for (int c = 0; c < sp->R.chains; c++) { if (status) status[c] = sc[c].status; if (n_divergent) n_divergent[c] = sc[c].n_divergent; }
  return 0;
}

int potus_get_adaptation(int handle, double *stepsize, double *inv_metric) {
  Sampler *sp = get(handle);
  if (!sp) return fail(POTUS_ERR_STATE, "bad handle");
  HIP_TRY(hipSetDevice(sp->device));
  std::vector<ChainScalars> sc;
  { const int rc_ = read_scalars(sp, sc); if (rc_) return rc_; }
  for (int c = 0; c < sp->R.chains; c++) {
    if (stepsize) stepsize[c] = sc[c].nom_eps;
    if (inv_metric && sp->dense) {   // the diagonal of the dense inverse metric
      HIP_TRY(hipMemcpy(inv_metric + (size_t)c * sp->L.D, dn_diag(sp->dn, c), (size_t)sp->L.D * 8, hipMemcpyDeviceToHost));   // (pooled: the one diagonal for every chain)
    } else if (inv_metric) {
      double *dst = inv_metric + (size_t)c * sp->L.D;
      if (sp->K == 1) HIP_TRY(hipMemcpy(dst, sp->R.state + ((size_t)c * V_COUNT + V_MINV) * sp->R.Dpad, (size_t)sp->L.D * 8, hipMemcpyDeviceToHost));
      if (sp->K > 1) {   // the cluster keeps its vectors in internal order (with padding)
        std::vector<double> tmp(sp->CL.Dint);
        HIP_TRY(hipMemcpy(tmp.data(), sp->R.state + ((size_t)c * V_COUNT + V_MINV) * sp->R.Dpad, (size_t)sp->CL.Dint * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < sp->CL.Dint; i++) if (sp->h_perm[i] >= 0) dst[sp->h_perm[i]] = tmp[i];
      }
    }
  }
  return 0;
}

int potus_get_dense_metric(int handle, int chain, double *inv_metric) {
  Sampler *sp = get(handle);
  if (!sp || !inv_metric) return fail(POTUS_ERR_STATE, "bad handle or null output");
  if (chain < 0 || chain >= sp->R.chains) return fail(POTUS_ERR_ARG, "chain %d out of range", chain);
  if (!sp->dense) return fail(POTUS_ERR_STATE, "the handle runs the diagonal metric: use potus_get_adaptation");
  HIP_TRY(hipSetDevice(sp->device));
  const size_t D = sp->L.D;
  // the strict upper triangle of the chain's matrix is M^-1's (the lower one holds its Cholesky factor), the diagonal a vector
  if (sp->dn.f32 && !sp->dn.pooled) {   // the upper triangle lives as floats in the second half of each row (dn_f32_row); pooled: A holds the rounded values as doubles
    const size_t LD = sp->dn.LD, rows_at_once = 256;
    std::vector<double> buf(rows_at_once * LD);
    for (size_t r0 = 0; r0 < D; r0 += rows_at_once) {
      const size_t nr = std::min(rows_at_once, D - r0);
      HIP_TRY(hipMemcpy(buf.data(), sp->dn.A + ((size_t)chain * D + r0) * LD, nr * LD * 8, hipMemcpyDeviceToHost));
      for (size_t r = 0; r < nr; r++) {
        const float *f = dn_f32_row(buf.data(), (int)LD, (int)r);
        for (size_t j = r0 + r + 1; j < D; j++) inv_metric[(r0 + r) * D + j] = (double)f[j];
      }
    }
  } else
  HIP_TRY(hipMemcpy2D(inv_metric, D * 8, dn_mat(sp->dn, chain), (size_t)sp->dn.LD * 8, D * 8, D, hipMemcpyDeviceToHost));   // (pooled: every chain's is the handle's one)
  std::vector<double> dg(D);
  HIP_TRY(hipMemcpy(dg.data(), dn_diag(sp->dn, chain), D * 8, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < D; i++) {
    inv_metric[i * D + i] = dg[i];
    for (size_t j = 0; j < i; j++) inv_metric[i * D + j] = inv_metric[j * D + i];
  }
  return 0;
}

// Dense metric: time spent in the matrix passes (k_dn_symv + k_dn_symv_finish, HIP events on the sampler's stream), their number and
// the bytes of matrix they loaded (active chains x the upper-triangle tiles, about 4 D^2 per pass), since the handle was created.
int potus_dense_timing(int handle, double *matvec_ms, long long *passes, long long *bytes, long long *rounds) {
  Sampler *sp = get(handle);
  if (!sp || !sp->dense) return fail(POTUS_ERR_STATE, "bad handle or not a dense-metric sampler");
  if (matvec_ms) *matvec_ms = sp->mv_ms;
  if (passes) *passes = sp->mv_calls;
  if (bytes) {   // (chain, pass) pairs counted by the timed passes themselves x the bytes a pass loads per chain
    unsigned long long np_ = 0;
    DeviceGuard guard;
    DeviceLocks lock(sp->device);
    HIP_TRY(hipSetDevice(sp->device));
    HIP_TRY(hipMemcpy(&np_, sp->dn.act_passes, 8, hipMemcpyDeviceToHost));
    *bytes = (long long)np_ * sp->dn_pass_bytes[sp->dn.rb == DN_RB ? 0 : 1];
  }
  if (rounds) *rounds = sp->dn_rounds;
  return 0;
}

// Dense metric: what the window ends of the warm-up have cost so far -- milliseconds in the covariance, in the blocked Cholesky
// factorisation and in the init_stepsize that follows (host clock around synchronised sections), and their number.
int potus_dense_adapt_timing(int handle, double *cov_ms, double *chol_ms, double *init_stepsize_ms, int *window_ends) {
  Sampler *sp = get(handle);
  if (!sp || !sp->dense) return fail(POTUS_ERR_STATE, "bad handle or not a dense-metric sampler");
  if (cov_ms) *cov_ms = sp->we_cov_ms;
  if (chol_ms) *chol_ms = sp->we_chol_ms;
  if (init_stepsize_ms) *init_stepsize_ms = sp->we_eps_ms;
  if (window_ends) *window_ends = sp->we_count;
  return 0;
}

// pooled_metric = 2: the window end in two halves, so that the host can pool over handles and ranks in between.  potus_run stops after the transition that ends
// a window (potus_iterations_done says where); *pending = 1 then, *count = the draws behind the handle's moments (chains x window length), *mean_dev /
// *m2_dev = DEVICE pointers to the handle's mean [D] and M2 = sum of centred outer products [D rows of *ld doubles, both triangles] -- the host replaces M2 by the
// pooled one (Chan's update: M2 += count (mean - pooled mean)(mean - pooled mean)', then the sum over everybody; us_potus_model_amd/parallel.py) and calls
// potus_dense_pool_finish with the pooled count: M^-1 = N/(N+5) M2/(N-1) + 1e-3 5/(N+5) I, its factor, init_stepsize.  With one handle and nothing in between,
// finish(count) is exactly pooled_metric = 1.
int potus_dense_pool_window(int handle, int *pending, double *count, void **mean_dev, void **m2_dev, long long *ld) {
  Sampler *sp = get(handle);
  if (!sp || !pending) return fail(POTUS_ERR_STATE, "bad handle or null output");
  if (!sp->dense || !sp->dn.pooled) return fail(POTUS_ERR_STATE, "potus_dense_pool_window: the handle does not run a pooled dense metric");
  *pending = sp->dn_pending ? 1 : 0;
  if (count) *count = sp->dn_pending ? (double)sp->R.chains * (double)sp->dn_pending_n : 0.0;
  if (mean_dev) *mean_dev = sp->dn.pmean;
  if (m2_dev) *m2_dev = sp->dn.A;
  if (ld) *ld = sp->dn.LD;
  return 0;
}
int potus_dense_pool_finish(int handle, double n_total) {
  Sampler *sp = get(handle);
  if (!sp) return fail(POTUS_ERR_STATE, "bad handle");
  if (!sp->dense || !sp->dn.pooled || !sp->dn_pending) return fail(POTUS_ERR_STATE, "potus_dense_pool_finish: no pooled window end is pending on this handle");
  if (!(n_total >= (double)sp->R.chains * (double)sp->dn_pending_n)) return fail(POTUS_ERR_ARG, "potus_dense_pool_finish: the pooled count cannot be below the handle's own %d x %d draws", sp->R.chains, sp->dn_pending_n);
  DeviceGuard guard;
  DeviceLocks lock(sp->device);
  HIP_TRY(hipSetDevice(sp->device));
  sp->dn_pending = false;
  const int rc = dense_window_finish(sp, sp->dn_pending_n, sp->dn_pending_iter, n_total, sp->dn_pending_cov_ms);   // (the time between the halves is the host's)
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(sp->stream));
  return check_chains(sp);
}

// Dense metric, verification hook (like potus_log_prob_grad): is the factor in the lower triangle the factor of the metric the
// leapfrog multiplies with?  For n_probe standard-normal vectors x: M^-1 x by the sampler's own matrix pass (upper triangle +
// diagonal, fp64 or fp32 storage) against L (L' x) by plain kernels over the lower triangle; and the momentum draw's own blocked
// back substitution L' p = u, checked by multiplying back.  out[0] = max ||L L' x - M^-1 x|| / ||M^-1 x||, out[1] = ||L' p - u|| / ||u||.
int potus_dense_check(int handle, int chain, int n_probe, double *out) {
  Sampler *sp = get(handle);
  if (!sp || !out) return fail(POTUS_ERR_STATE, "bad handle or null output");
  if (!sp->dense) return fail(POTUS_ERR_STATE, "the handle runs the diagonal metric");
  if (chain < 0 || chain >= sp->R.chains || n_probe < 1) return fail(POTUS_ERR_ARG, "potus_dense_check: bad chain or probe count");
  if (!sp->inited) return fail(POTUS_ERR_STATE, "potus_dense_check: call potus_init first (the check runs the sampler's kernels on the chain's state block)");
  DeviceGuard guard;                                   // the caller's device comes back
  DeviceLocks lock(sp->device);
  HIP_TRY(hipSetDevice(sp->device));
  {
    int failed = 0;
    HIP_TRY(hipMemcpy(&failed, sp->dn.fail, 4, hipMemcpyDeviceToHost));
    if (failed) return fail(POTUS_ERR_STATE, "potus_dense_check: the last window end did not produce a factor (the adapted covariance was not positive definite)");
  }
  DnParams &P = sp->dn;
  P.count_passes = 0;                                  // verification passes are not part of potus_dense_timing's set
  const int D = P.D, chains = P.chains;
  std::vector<DnRound> rds(chains);
  for (int c = 0; c < chains; c++) { std::memset(&rds[c], 0, sizeof(DnRound)); rds[c].active = c == chain; }
  rds[chain].job[0] = DnJob{DV_TMPP, DV_SCR1, -1, -1, -1, 0, 0.0};              // SCR1 = M^-1 TMPP
  HIP_TRY(hipMemcpyAsync(P.rd, rds.data(), rds.size() * sizeof(DnRound), hipMemcpyHostToDevice, sp->stream));
  for (int c = 0; c < chains; c++) sp->h_active[c] = c == chain;
  std::vector<double> a(D), b(D);
  auto fetch = [&](int slot, std::vector<double> &v) {
    return hipMemcpyAsync(v.data(), P.state + ((size_t)chain * DV_COUNT + slot) * P.LD, (size_t)D * 8, hipMemcpyDeviceToHost, sp->stream);
  };
  out[0] = out[1] = 0.0;
  for (int k = 0; k <= n_probe; k++) {
    // standard normals -> P0; the last round solves L' p = u with the sampler's kernels and multiplies back
    const bool solve = k == n_probe;
    rds[chain].aux = k;
    HIP_TRY(hipMemcpyAsync(&P.rd[chain].aux, &rds[chain].aux, sizeof(int), hipMemcpyHostToDevice, sp->stream));
    const int id0 = P.identity;
    if (!solve) P.identity = 1;                                                  // (dense_sample_p: normals only)
    const int rc = dense_sample_p(sp, 0x7ffffff0u, RNG_INIT_EPS);
    P.identity = id0;
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(P.state + ((size_t)chain * DV_COUNT + DV_TMPP) * P.LD, P.state + ((size_t)chain * DV_COUNT + DV_P0) * P.LD, (size_t)D * 8,
                           hipMemcpyDeviceToDevice, sp->stream));
    hipLaunchKernelGGL(k_dn_chk_ltx, dim3((D + 255) / 256), dim3(256), 0, sp->stream, P, chain, (int)DV_TMPP, (int)DV_TMPQ);   // TMPQ = L' TMPP
    if (solve) {
      P.identity = 1;
      const int rc2 = dense_sample_p(sp, 0x7ffffff0u, RNG_INIT_EPS);            // the same normals again: u
      P.identity = id0;
      if (rc2) return rc2;
      HIP_TRY(fetch(DV_TMPQ, a)); HIP_TRY(fetch(DV_P0, b));
    } else {
      hipLaunchKernelGGL(k_dn_chk_lx, dim3(D), dim3(256), 0, sp->stream, P, chain, (int)DV_TMPQ, (int)DV_SCR0);                 // SCR0 = L L' x
      DnActive act; act.n = 1; act.idx[0] = chain;
      dense_launch_shape(P, 1);
      dense_symv_launch(sp->stream, P, act, 1);
      HIP_TRY(fetch(DV_SCR0, a)); HIP_TRY(fetch(DV_SCR1, b));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(sp->stream));
    double num = 0, den = 0;
    for (int i = 0; i < D; i++) { num += (a[i] - b[i]) * (a[i] - b[i]); den += b[i] * b[i]; }
    const double rel = std::sqrt(num / std::max(den, 1e-300));
    if (solve) out[1] = rel; else out[0] = std::max(out[0], rel);
  }
  return 0;
}

static int saved_count(Sampler *sp, int *n_saved) {
  std::vector<ChainScalars> sc;
  { const int rc_ = read_scalars(sp, sc); if (rc_) return rc_; }
  int m = sc[0].saved;
  for (auto &s : sc) m = std::min(m, s.saved);
  *n_saved = m;
  return 0;
}

int potus_get_draws(int handle, double *out, int *n_saved) {
  Sampler *sp = get(handle);
  if (!sp || !n_saved) return fail(POTUS_ERR_STATE, "bad handle");
  HIP_TRY(hipSetDevice(sp->device));
  int rc = saved_count(sp, n_saved);
  if (rc) return rc;
  if (out) {
    for (int c = 0; c < sp->R.chains; c++)
      HIP_TRY(hipMemcpy(out + (size_t)c * *n_saved * sp->R.row, sp->R.draws + (size_t)c * sp->R.n_save_max * sp->R.row,
                        (size_t)*n_saved * sp->R.row * 8, hipMemcpyDeviceToHost));
  }
  return 0;
}

int potus_draws_device_ptr(int handle, void **dptr, long long *n_doubles) {
  Sampler *sp = get(handle);
  if (!sp || !dptr) return fail(POTUS_ERR_STATE, "bad handle");
  *dptr = sp->R.draws;
  if (n_doubles) *n_doubles = (long long)sp->R.chains * sp->R.n_save_max * sp->R.row;
  return 0;
}

// out: host buffer, or (device_out) a device buffer of the sampler's GPU that receives the rows directly
static int write_array_range(Sampler *sp, int n_saved, int col_begin, int col_end, double *out, bool device_out = false, int out_stride = 0) {
  const int nsel = col_end - col_begin;
  if (out_stride <= 0) out_stride = nsel;
  const int ndraw = n_saved * sp->R.chains;
  if (ndraw == 0) return 0;
  const int grid = std::min(ndraw, 512);
  DevBufs tmp;
  double *scratch = nullptr, *dout = nullptr;
  HIP_TRY(tmp.alloc(&scratch, (size_t)grid * sp->L.ncols * 8));
  if (device_out) dout = out;
  else HIP_TRY(tmp.alloc(&dout, (size_t)ndraw * nsel * 8));
  WAParams W{sp->R.draws, sp->R.chains, sp->R.n_save_max, n_saved, sp->R.row, sp->L.ncols, col_begin, col_end, scratch, dout, sp->sigma_ns, sp->sigma_nn, out_stride};
  hipLaunchKernelGGL(k_write_array, dim3(grid), dim3(256), 0, sp->stream, (const DevModel *)sp->dM, W);
  HIP_TRY(hipGetLastError());
  if (!device_out) HIP_TRY(hipMemcpyAsync(out, dout, (size_t)ndraw * nsel * 8, hipMemcpyDeviceToHost, sp->stream));
  HIP_TRY(hipStreamSynchronize(sp->stream));
  return 0;
}

int potus_write_array(int handle, int col_begin, int col_end, double *out) {
  Sampler *sp = get(handle);
  if (!sp || !out) return fail(POTUS_ERR_STATE, "bad handle or null output");
  if (col_begin < 0 || col_end > sp->L.ncols || col_begin >= col_end) return fail(POTUS_ERR_ARG, "bad column range [%d,%d) of %d", col_begin, col_end, sp->L.ncols);
  DeviceLocks lock(sp->device);
  HIP_TRY(hipSetDevice(sp->device));
  int n_saved = 0, rc = saved_count(sp, &n_saved);
  if (rc) return rc;
  return write_array_range(sp, n_saved, col_begin, col_end, out);
}

// rstan::extract(out, pars)[[1]] as R stores it, in ONE pass and in place: `out` is a column-major matrix [rows, col_end - col_begin] whose row
// chain_global * n_saved + iteration holds a draw -- chains merged chain after chain, the handles' chains in the order listed (R/potus_sampling.R deals
// consecutive chain ids to consecutive handles).  *rows_out = the rows the listed handles hold (call with out = NULL to size the result); rows must equal it.
// What R's .Call() wrapper (R/src/potus_call.c) fills an allocVector'ed result with: no second copy in R, long vectors welcome (the .C() path hands R
// [iteration][chain][column] rows, which the shim then permutes -- two more copies of 0.83 GB for predicted_score of 8 x 1000 draws, final_2016.R:708).
__global__ __launch_bounds__(256) void k_rows_to_r_matrix(const double *in, double *out, int n_saved, int chains, int ncol) {
  // in [iteration][chain][column] -> out [column][chain][iteration]; tiles of 32 x 32 through LDS
  __shared__ double tile[32][33];
  const long long nrow = (long long)n_saved * chains;
  const long long r0 = (long long)blockIdx.x * 32;
  const int c0 = (int)blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const long long r = r0 + k;
    tile[k][tx] = (r < nrow && c0 + tx < ncol) ? in[r * ncol + c0 + tx] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const long long r = r0 + tx;                                   // source row = iteration * chains + chain
    if (r < nrow && c0 + k < ncol) {
      const long long it = r / chains, ch = r % chains;
      out[(long long)(c0 + k) * nrow + ch * n_saved + it] = tile[tx][k];
    }
  }
}
int potus_extract_matrix(const int *handles, int n_handles, int col_begin, int col_end, double *out, long long rows, long long *rows_out) {
  if (!handles || n_handles < 1) return fail(POTUS_ERR_ARG, "potus_extract_matrix: null handle list");
  std::vector<Sampler *> sps;
  std::vector<int> saved(n_handles, 0), devs;
  long long total = 0;
  for (int i = 0; i < n_handles; i++) {
    Sampler *sp = get(handles[i]);
    if (!sp) return fail(POTUS_ERR_STATE, "potus_extract_matrix: bad handle %d", handles[i]);
    for (int j = 0; j < i; j++) if (handles[j] == handles[i]) return fail(POTUS_ERR_ARG, "potus_extract_matrix: handle %d listed twice", handles[i]);
    if (i > 0 && sp->L.ncols != sps[0]->L.ncols) return fail(POTUS_ERR_ARG, "potus_extract_matrix: the handles hold different posteriors");
    sps.push_back(sp); devs.push_back(sp->device);
  }
  if (col_begin < 0 || col_end > sps[0]->L.ncols || col_begin >= col_end) return fail(POTUS_ERR_ARG, "bad column range [%d,%d) of %d", col_begin, col_end, sps[0]->L.ncols);
  DeviceGuard guard;
  DeviceLocks lock(devs);
  for (int i = 0; i < n_handles; i++) {
    HIP_TRY(hipSetDevice(sps[i]->device));
    const int rc = saved_count(sps[i], &saved[i]);
    if (rc) return rc;
    if (i > 0 && saved[i] != saved[0]) return fail(POTUS_ERR_STATE, "potus_extract_matrix: the handles have saved different numbers of draws (%d, %d)", saved[0], saved[i]);
    total += (long long)saved[i] * sps[i]->R.chains;
  }
  if (rows_out) *rows_out = total;
  if (!out) return 0;
  if (rows != total) return fail(POTUS_ERR_ARG, "potus_extract_matrix: the result has %lld rows, the handles hold %lld draws", rows, total);
  const int nsel = col_end - col_begin;
  long long off = 0;
  for (int i = 0; i < n_handles; i++) {
    Sampler *sp = sps[i];
    const long long nrow = (long long)saved[i] * sp->R.chains;
    if (nrow == 0) continue;
    HIP_TRY(hipSetDevice(sp->device));
    DevBufs tmp;
    double *rowsd = nullptr, *colsd = nullptr;
    HIP_TRY(tmp.alloc(&rowsd, (size_t)nrow * nsel * 8)); HIP_TRY(tmp.alloc(&colsd, (size_t)nrow * nsel * 8));
    const int rc = write_array_range(sp, saved[i], col_begin, col_end, rowsd, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rows_to_r_matrix, dim3((unsigned)((nrow + 31) / 32), (unsigned)((nsel + 31) / 32)), dim3(256), 0, sp->stream, rowsd, colsd, saved[i], sp->R.chains, nsel);
    HIP_TRY(hipGetLastError());
    // column k of this handle's block -> rows [off, off + nrow) of column k of the result
    HIP_TRY(hipMemcpy2DAsync(out + off, (size_t)total * 8, colsd, (size_t)nrow * 8, (size_t)nrow * 8, (size_t)nsel, hipMemcpyDeviceToHost, sp->stream));
    HIP_TRY(hipStreamSynchronize(sp->stream));
    off += nrow;
  }
  return 0;
}

// The same rows written straight into a DEVICE buffer of the sampler's GPU (e.g. a torch tensor's data_ptr()): what
// the RCCL all-gather of the draws-of-interest sends, without a trip through host memory.
int potus_write_array_device(int handle, int col_begin, int col_end, void *out_device) {
  Sampler *sp = get(handle);
  if (!sp || !out_device) return fail(POTUS_ERR_STATE, "bad handle or null output");
  if (col_begin < 0 || col_end > sp->L.ncols || col_begin >= col_end) return fail(POTUS_ERR_ARG, "bad column range [%d,%d) of %d", col_begin, col_end, sp->L.ncols);
  DeviceLocks lock(sp->device);
  HIP_TRY(hipSetDevice(sp->device));
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, out_device) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != sp->device) {
    (void)hipGetLastError();
    return fail(POTUS_ERR_ARG, "potus_write_array_device: the output is not device memory of GPU %d", sp->device);
  }
  int n_saved = 0, rc = saved_count(sp, &n_saved);
  if (rc) return rc;
  return write_array_range(sp, n_saved, col_begin, col_end, (double *)out_device, true);
}

// Pooled over every saved draw of every listed sampler (the chains of one posterior may sit in several handles, on one
// GPU or several: PotusModel.sample(devices=...), potus_sampling.R gpus=...).  The work runs on the first handle's GPU;
// predicted_score blocks of other GPUs are brought over with a peer copy.  Any number of draws (potus_summary.hpp).
int potus_posterior_summary_many(const int *handles, int n_handles, const double *ev, double *state_out, double *natl_out, double *ev_out) {
  if (!handles || n_handles < 1 || !ev || !state_out || !natl_out || !ev_out) return fail(POTUS_ERR_ARG, "potus_posterior_summary: null argument");
  std::vector<Sampler *> sps;
  std::vector<int> n_saved(n_handles, 0), devs;
  long long nd = 0;
  for (int i = 0; i < n_handles; i++) {
    Sampler *sp = get(handles[i]);
    if (!sp) return fail(POTUS_ERR_STATE, "potus_posterior_summary: bad handle %d", handles[i]);
    devs.push_back(sp->device);
  }
  DeviceGuard guard;                                   // whatever happens below, the caller's device comes back
  DeviceLocks lock(devs);
  for (int i = 0; i < n_handles; i++) {
    Sampler *sp = get(handles[i]);
    for (int j = 0; j < i; j++) if (handles[j] == handles[i]) return fail(POTUS_ERR_ARG, "potus_posterior_summary: handle %d listed twice", handles[i]);
    // pooling is for the chains of ONE posterior: same shape, same variant, same data (its hash) -- the first handle's state
    // weights are the ones the national vote is taken with
    if (i > 0 && (sp->M.S != sps[0]->M.S || sp->M.T != sps[0]->M.T || sp->L.ncols != sps[0]->L.ncols || sp->M.full != sps[0]->M.full ||
                  sp->data_hash != sps[0]->data_hash))
      return fail(POTUS_ERR_ARG, "potus_posterior_summary: handle %d holds another posterior than handle %d (pooled summaries are for the chains of one)", handles[i], handles[0]);
    HIP_TRY(hipSetDevice(sp->device));
    int rc = saved_count(sp, &n_saved[i]);
    if (rc) return rc;
    nd += (long long)n_saved[i] * sp->R.chains;
    sps.push_back(sp);
  }
  Sampler *s0 = sps[0];
  const int S = s0->M.S, T = s0->M.T, TS = S * T, NC = TS + 2 * T;
  if (nd < 2) return fail(POTUS_ERR_STATE, "posterior summaries need at least two saved draws");
  const int col_end = s0->L.ncols, col_begin = col_end - TS;   // predicted_score = the generated-quantities block
  HIP_TRY(hipSetDevice(s0->device));
  DevBufs tmp;
  double *full = nullptr, *cols = nullptr, *dw = nullptr, *dev_ = nullptr, *dout = nullptr, *scr = nullptr;
  HIP_TRY(tmp.alloc(&full, (size_t)nd * NC * 8));
  HIP_TRY(tmp.alloc(&cols, (size_t)nd * NC * 8));
  HIP_TRY(tmp.alloc(&dw, (size_t)S * 8)); HIP_TRY(tmp.alloc(&dev_, (size_t)S * 8));
  HIP_TRY(tmp.alloc(&dout, ((size_t)TS * 4 + (size_t)T * 9) * 8));
  long long row0 = 0;
  for (int i = 0; i < n_handles; i++) {
    Sampler *sp = sps[i];
    const long long rows = (long long)n_saved[i] * sp->R.chains;
    if (rows == 0) continue;
    double *dst = full + (size_t)row0 * NC;
    int rc;
    if (sp->device == s0->device) {
      if ((rc = write_array_range(sp, n_saved[i], col_begin, col_end, dst, true, NC))) return rc;
    } else {
      HIP_TRY(hipSetDevice(sp->device));
      DevBufs far;                                     // freed on its own device before we go back
      double *blk = nullptr;
      HIP_TRY(far.alloc(&blk, (size_t)rows * NC * 8));
      if ((rc = write_array_range(sp, n_saved[i], col_begin, col_end, blk, true, NC))) return rc;
      // (hipMemcpyPeer stages through the host when the two GPUs have no peer access; with it, the copy goes over xGMI)
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, s0->device, sp->device) == hipSuccess && can) { (void)hipSetDevice(s0->device); (void)hipDeviceEnablePeerAccess(sp->device, 0); (void)hipGetLastError(); (void)hipSetDevice(sp->device); }
      HIP_TRY(hipMemcpyPeer(dst, s0->device, blk, sp->device, (size_t)rows * NC * 8));
      HIP_TRY(hipSetDevice(s0->device));
    }
    row0 += rows;
  }
  HIP_TRY(hipSetDevice(s0->device));
  HIP_TRY(hipMemcpyAsync(dw, s0->h_w.data(), (size_t)S * 8, hipMemcpyHostToDevice, s0->stream));
  HIP_TRY(hipMemcpyAsync(dev_, ev, (size_t)S * 8, hipMemcpyHostToDevice, s0->stream));
  hipLaunchKernelGGL(k_ps_derived, dim3((unsigned)std::min<long long>((nd * T + 255) / 256, 65535)), dim3(256), 0, s0->stream, full, nd, T, S,
                     (const double *)dw, (const double *)dev_);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_ps_transpose, dim3((NC + 63) / 64, (unsigned)((nd + 63) / 64)), dim3(256), 0, s0->stream, (const double *)full, cols, nd, NC);
  HIP_TRY(hipGetLastError());
  int npad = 1;
  while (npad < nd && npad < PS_RUN) npad <<= 1;
  const size_t lds = (size_t)npad * 8;
  const int grid = std::min(NC, 1024);
  if (nd > PS_RUN) HIP_TRY(tmp.alloc(&scr, (size_t)grid * (size_t)nd * 8));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_col_summary), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  double *o_state = dout, *o_natl = dout + (size_t)TS * 4, *o_ev = o_natl + (size_t)T * 4;
  hipLaunchKernelGGL(k_col_summary, dim3(grid), dim3(PS_THREADS), lds, s0->stream, (const double *)cols, scr, nd, T, S, o_state, o_natl, o_ev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(state_out, o_state, (size_t)TS * 4 * 8, hipMemcpyDeviceToHost, s0->stream));
  HIP_TRY(hipMemcpyAsync(natl_out, o_natl, (size_t)T * 4 * 8, hipMemcpyDeviceToHost, s0->stream));
  HIP_TRY(hipMemcpyAsync(ev_out, o_ev, (size_t)T * 5 * 8, hipMemcpyDeviceToHost, s0->stream));
  HIP_TRY(hipStreamSynchronize(s0->stream));
  return 0;
}

int potus_posterior_summary(int handle, const double *ev, double *state_out, double *natl_out, double *ev_out) {
  return potus_posterior_summary_many(&handle, 1, ev, state_out, natl_out, ev_out);
}

// ---------------------------------------------------------------------------------------------- diagnostics (potus_diag.hpp)
namespace {
// cols [NC][C][n] on the current device -> rhat / bulk ESS per column (host arrays)
int diagnostics_of_columns(hipStream_t stream, const double *cols, long long n, int C, int NC, double *rhat_out, double *ess_out) {
  if (2 * C > DG_MAXCH) return fail(POTUS_ERR_UNSUPPORTED, "potus_diagnostics: %d chains pooled (at most %d)", C, DG_MAXCH / 2);
  const long long N = 2ll * C * (n / 2);
  DevBufs tmp;
  double *zbuf = nullptr, *dout = nullptr;
  unsigned long long *rkey = nullptr;
  unsigned *ridx = nullptr;
  // a workgroup's scratch: two rank-normalised copies of the column, and the sorted runs when they do not fit LDS; the grid is
  // capped so that the scratch of a call stays within 768 MB whatever the number of pooled draws (a workgroup loops over columns)
  const long long per_wg = std::max<long long>(N, 1) * (16 + (N > DG_RUN ? 12 : 0));
  const int grid = (int)std::max<long long>(1, std::min<long long>(std::min(NC, 1024), (768ll << 20) / per_wg));
  HIP_TRY(tmp.alloc(&zbuf, (size_t)grid * 2 * (size_t)std::max<long long>(N, 1) * 8));
  if (N > DG_RUN) { HIP_TRY(tmp.alloc(&rkey, (size_t)grid * (size_t)N * 8)); HIP_TRY(tmp.alloc(&ridx, (size_t)grid * (size_t)N * 4)); }
  HIP_TRY(tmp.alloc(&dout, (size_t)NC * 2 * 8));
  int npad = 1;
  while (npad < N && npad < DG_RUN) npad <<= 1;
  const size_t lds = (size_t)npad * 12;                // keys + split indices
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dg_column), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  DgParams P{cols, zbuf, rkey, ridx, dout, dout + NC, n, C, NC};
  hipLaunchKernelGGL(k_dg_column, dim3(grid), dim3(DG_THREADS), lds, stream, P);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(rhat_out, dout, (size_t)NC * 8, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(ess_out, dout + NC, (size_t)NC * 8, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  return 0;
}
} // namespace

int potus_diagnostics_device(int device, const void *block, long long n_draws, int n_chains, int n_cols, double *rhat_out, double *ess_bulk_out) {
  if (!block || !rhat_out || !ess_bulk_out || n_draws < 1 || n_chains < 1 || n_cols < 1) return fail(POTUS_ERR_ARG, "potus_diagnostics_device: bad argument");
  if (2 * n_chains > DG_MAXCH) return fail(POTUS_ERR_UNSUPPORTED, "potus_diagnostics_device: %d chains pooled (at most %d)", n_chains, DG_MAXCH / 2);   // before anything is allocated
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(POTUS_ERR_DEVICE, "potus_diagnostics_device: no HIP device %d", device);
  DeviceGuard guard;
  DeviceLocks lock(device);
  HIP_TRY(hipSetDevice(device));
  DevBufs tmp;
  double *cols = nullptr;
  HIP_TRY(tmp.alloc(&cols, (size_t)n_draws * n_chains * n_cols * 8));
  hipLaunchKernelGGL(k_dg_transpose, dim3((n_cols + 63) / 64, (unsigned)((n_draws + 63) / 64), n_chains), dim3(256), 0, 0, (const double *)block, cols, n_draws, n_chains,
                     n_cols, n_chains, 0);
  HIP_TRY(hipGetLastError());
  return diagnostics_of_columns(0, cols, n_draws, n_chains, n_cols, rhat_out, ess_bulk_out);
}

// (have_locks: the caller already holds the device locks of every handle -- potus_check_convergence takes its two column ranges under ONE set of locks, so that
//  no run on another host thread can add draws between them)
static int diagnostics_impl(const int *handles, int n_handles, int col_begin, int col_end, double *rhat_out, double *ess_bulk_out, bool have_locks) {
  if (!handles || n_handles < 1 || !rhat_out || !ess_bulk_out) return fail(POTUS_ERR_ARG, "potus_diagnostics: null argument");
  std::vector<Sampler *> sps;
  std::vector<int> devs;
  for (int i = 0; i < n_handles; i++) {
    Sampler *sp = get(handles[i]);
    if (!sp) return fail(POTUS_ERR_STATE, "bad handle %d", handles[i]);
    for (int j = 0; j < i; j++) if (handles[j] == handles[i]) return fail(POTUS_ERR_ARG, "potus_diagnostics: handle %d listed twice", handles[i]);
    sps.push_back(sp); devs.push_back(sp->device);
  }
  Sampler *s0 = sps[0];
  if (col_begin < 0 || col_end > s0->L.ncols || col_begin >= col_end) return fail(POTUS_ERR_ARG, "potus_diagnostics: columns [%d, %d) of %d", col_begin, col_end, s0->L.ncols);
  DeviceGuard guard;
  DeviceLocks lock(have_locks ? std::vector<int>{} : devs);
  int n_saved = -1, Ctot = 0;
  for (size_t i = 0; i < sps.size(); i++) {
    Sampler *sp = sps[i];
    if (i > 0 && (sp->L.ncols != s0->L.ncols || sp->M.full != s0->M.full || sp->data_hash != s0->data_hash))
      return fail(POTUS_ERR_ARG, "potus_diagnostics: handle %d holds another posterior than handle %d (R-hat / ESS pool the chains of one)", handles[i], handles[0]);
    HIP_TRY(hipSetDevice(sp->device));
    int ns = 0, rc = saved_count(sp, &ns);
    if (rc) return rc;
    if (n_saved >= 0 && ns != n_saved) return fail(POTUS_ERR_STATE, "potus_diagnostics: handle %d has saved %d draws per chain, handle %d has %d", handles[i], ns, handles[0], n_saved);
    n_saved = ns; Ctot += sp->R.chains;
  }
  if (2 * Ctot > DG_MAXCH) return fail(POTUS_ERR_UNSUPPORTED, "potus_diagnostics: %d chains pooled (at most %d)", Ctot, DG_MAXCH / 2);
  // warm-up rows (save_warmup = 1) are not draws from the posterior: rstan::monitor and extract() drop them, and so does this
  const int n_warm_rows = s0->opts.save_warmup ? std::min(n_saved, s0->R.num_warmup) : 0;
  for (Sampler *sp : sps)
    if ((sp->opts.save_warmup ? std::min(n_saved, sp->R.num_warmup) : 0) != n_warm_rows) return fail(POTUS_ERR_ARG, "potus_diagnostics: the handles saved different numbers of warm-up rows");
  const int n_post = n_saved - n_warm_rows;
  if (n_post < 4) return fail(POTUS_ERR_STATE, "R-hat / ESS need at least four saved post-warm-up draws per chain (%d saved, %d of them warm-up)", n_saved, n_warm_rows);
  const int NC = col_end - col_begin;
  HIP_TRY(hipSetDevice(s0->device));
  DevBufs tmp;
  double *cols = nullptr;
  HIP_TRY(tmp.alloc(&cols, (size_t)n_post * Ctot * NC * 8));
  int coff = 0;
  for (size_t i = 0; i < sps.size(); i++) {
    Sampler *sp = sps[i];
    const int C = sp->R.chains;
    DevBufs blkbuf;
    double *blk = nullptr;
    int rc;
    HIP_TRY(hipSetDevice(s0->device));
    HIP_TRY(blkbuf.alloc(&blk, (size_t)n_saved * C * NC * 8));
    if (sp->device == s0->device) {
      if ((rc = write_array_range(sp, n_saved, col_begin, col_end, blk, true, NC))) return rc;
      HIP_TRY(hipStreamSynchronize(sp->stream));
    } else {
      HIP_TRY(hipSetDevice(sp->device));
      DevBufs far;
      double *fb = nullptr;
      HIP_TRY(far.alloc(&fb, (size_t)n_saved * C * NC * 8));
      if ((rc = write_array_range(sp, n_saved, col_begin, col_end, fb, true, NC))) return rc;
      HIP_TRY(hipStreamSynchronize(sp->stream));
      HIP_TRY(hipMemcpyPeer(blk, s0->device, fb, sp->device, (size_t)n_saved * C * NC * 8));
      HIP_TRY(hipSetDevice(s0->device));
    }
    hipLaunchKernelGGL(k_dg_transpose, dim3((NC + 63) / 64, (unsigned)((n_post + 63) / 64), C), dim3(256), 0, s0->stream, (const double *)(blk + (size_t)n_warm_rows * C * NC), cols,
                       (long long)n_post, C, NC, Ctot, coff);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s0->stream));
    coff += C;
  }
  return diagnostics_of_columns(s0->stream, cols, n_post, Ctot, NC, rhat_out, ess_bulk_out);
}

// SURVEY 8(f4), "online R-hat-based early stop": the diagnostics of lp__ and mu_b[:, T] (the columns bench.py's ESS / s is defined on;
// predicted_score[T, :] is their inverse logit) over the post-warm-up draws saved SO FAR by the pooled chains of the handles.  The host loop
// that advances the sampler in chunks of `refresh` transitions may stop once *converged is set; nothing the sampler does depends on it (same
// draws up to that point as an uninterrupted run).  A deviation from Stan, which always runs num_samples iterations: off unless the host asks.
int potus_diagnostics(const int *handles, int n_handles, int col_begin, int col_end, double *rhat_out, double *ess_bulk_out) {
  return diagnostics_impl(handles, n_handles, col_begin, col_end, rhat_out, ess_bulk_out, false);
}

int potus_check_convergence(const int *handles, int n_handles, double rhat_below, double ess_at_least, int *converged, double *rhat_max, double *ess_bulk_min) {
  if (!handles || n_handles < 1 || !converged || !rhat_max || !ess_bulk_min) return fail(POTUS_ERR_ARG, "potus_check_convergence: null argument");
  Sampler *s0 = get(handles[0]);
  if (!s0) return fail(POTUS_ERR_STATE, "bad handle %d", handles[0]);
  std::vector<int> devs;
  for (int i = 0; i < n_handles; i++) { Sampler *sp = get(handles[i]); if (!sp) return fail(POTUS_ERR_STATE, "bad handle %d", handles[i]); devs.push_back(sp->device); }
  DeviceLocks lock(devs);                            // one set of locks for the count and both column ranges
  const int S = s0->M.S, T = s0->M.T, a_mu = POTUS_N_SAMPLER_COLS + s0->L.D + S * (T - 1);
  std::vector<double> rh(1 + S), es(1 + S);
  *converged = 0; *rhat_max = NAN; *ess_bulk_min = NAN;
  int ns = 0;
  { DeviceGuard guard; HIP_TRY(hipSetDevice(s0->device)); const int rc = saved_count(s0, &ns); if (rc) return rc; }
  if (ns - (s0->opts.save_warmup ? std::min(ns, s0->R.num_warmup) : 0) < 4) return 0;      // too early to say anything: not converged, no error
  int rc;
  if ((rc = diagnostics_impl(handles, n_handles, 0, 1, rh.data(), es.data(), true))) return rc;
  if ((rc = diagnostics_impl(handles, n_handles, a_mu, a_mu + S, rh.data() + 1, es.data() + 1, true))) return rc;
  double r = -INFINITY, e = INFINITY;
  bool bad = false;
  for (int i = 0; i <= S; i++) { bad = bad || std::isnan(rh[i]) || std::isnan(es[i]); r = std::max(r, rh[i]); e = std::min(e, es[i]); }
  *rhat_max = bad ? NAN : r; *ess_bulk_min = bad ? NAN : e;
  *converged = (!bad && r < rhat_below && e >= ess_at_least) ? 1 : 0;
  return 0;
}

// The backtest scores of final_2016.R:925-945 (final_2012.R:918-931, final_2008.R:922-935) from the state summaries:
// with p_s = P(score > 0.5) of state s on `day` (1-based; 0 = the last day) and won_s the actual outcome,
//   out[0] = weighted.mean((won - p)^2, ev / sum(ev)), out[1] = mean((won - p)^2), out[2] = sum(round(p) == won)
// (R's round(): half to even).  Pure host arithmetic on the output of potus_posterior_summary.
int potus_backtest_scores(const double *state_out, int T, int S, int day, const double *ev, const int *won, double *out) {
  if (!state_out || !ev || !won || !out || T < 1 || S < 1 || day < 0 || day > T) return fail(POTUS_ERR_ARG, "potus_backtest_scores: bad argument");
  const int t = day == 0 ? T - 1 : day - 1;
  double evsum = 0, a = 0, b = 0;
  int correct = 0;
  for (int s = 0; s < S; s++) evsum += ev[s];
  if (!(evsum > 0)) return fail(POTUS_ERR_ARG, "potus_backtest_scores: electoral votes sum to %g", evsum);
  for (int s = 0; s < S; s++) {
    const double p = state_out[((size_t)t + (size_t)T * s) * 4 + 3], d = (double)won[s] - p;
    a += ev[s] / evsum * d * d;
    b += d * d;
    correct += (int)std::nearbyint(p) == won[s];
  }
  out[0] = a; out[1] = b / S; out[2] = correct;
  return 0;
}

int potus_last_run_timing(int handle, double *ms, long long *leapfrogs) {
  Sampler *sp = get(handle);
  if (!sp) return fail(POTUS_ERR_STATE, "bad handle");
  if (ms) *ms = sp->last_ms;
  if (leapfrogs) *leapfrogs = sp->last_leapfrogs;
  return 0;
}

// CmdStan-format CSV, one file per chain (what rstan::read_stan_csv consumes, final_2016.R:543)
int potus_write_stan_csv(int handle, const char *dir, const char *basename) {
  Sampler *sp = get(handle);
  if (!sp || !dir || !basename) return fail(POTUS_ERR_STATE, "bad handle or null path");
  int n_saved = 0, rc = 0;
  { DeviceLocks lock0(sp->device); HIP_TRY(hipSetDevice(sp->device)); rc = saved_count(sp, &n_saved); }
  if (rc) return rc;
  const int chains = sp->R.chains, ncols = sp->L.ncols, D = sp->L.D;
  std::vector<double> eps(chains), minv((size_t)chains * D);
  if ((rc = potus_get_adaptation(handle, eps.data(), minv.data()))) return rc;
  // reconstruct a potus_data shell for the column names
  potus_data dd{}; dd.S = sp->M.S; dd.T = sp->M.T; dd.P = sp->M.P; dd.M = sp->M.M; dd.Pop = sp->M.Pop;
  dd.N_state_polls = sp->M.Ns; dd.N_national_polls = sp->M.Nn; dd.variant = sp->M.full ? POTUS_VARIANT_FULL : POTUS_VARIANT_NO_MODE;
  std::vector<FILE *> fp(chains, nullptr);
  auto close_all = [&]() { for (FILE *f : fp) if (f) fclose(f); };
  for (int c = 0; c < chains; c++) {
    const std::string path = std::string(dir) + "/" + basename + "-" + std::to_string(sp->R.chain_id_offset + c + 1) + ".csv";
    fp[c] = fopen(path.c_str(), "w");
    if (!fp[c]) { close_all(); return fail(POTUS_ERR_IO, "cannot open %s", path.c_str()); }
    FILE *f = fp[c];
    const potus_opts &o = sp->opts;
    fprintf(f, "# stan_version_major = 2\n# stan_version_minor = 24\n# stan_version_patch = 1\n");
    fprintf(f, "# model = %s\n", sp->M.full ? "poll_model_2020_model" : "poll_model_2020_no_mode_adjustment_model");
    fprintf(f, "# method = sample (Default)\n#   sample\n#     num_samples = %d\n#     num_warmup = %d\n#     save_warmup = %d\n#     thin = 1 (Default)\n",
            o.num_samples, o.num_warmup, o.save_warmup);
    fprintf(f, "#     adapt\n#       engaged = 1 (Default)\n#       gamma = %g\n#       delta = %g\n#       kappa = %g\n#       t0 = %g\n#       init_buffer = %d\n#       term_buffer = %d\n#       window = %d\n",
            o.gamma, o.delta, o.kappa, o.t0, sp->R.init_buffer, sp->R.term_buffer, sp->R.window);
    fprintf(f, "#     algorithm = hmc (Default)\n#       hmc\n#         engine = nuts (Default)\n#           nuts\n#             max_depth = %d\n#         metric = %s\n#         metric_file =  (Default)\n#         stepsize = %g\n#         stepsize_jitter = 0 (Default)\n",
            o.max_depth, sp->dense ? "dense_e" : "diag_e (Default)", o.stepsize);
    fprintf(f, "# id = %d\n# data\n#   file = (in-memory)\n# init = %g\n# random\n#   seed = %llu\n# output\n#   file = %s\n#   diagnostic_file =  (Default)\n#   refresh = 100 (Default)\n",
            sp->R.chain_id_offset + c + 1, o.init_radius, (unsigned long long)o.seed, path.c_str());
    // (not CmdStan's: what the library resolved cus_per_chain = 0 / twin = -1 to -- draws are reproducible bit for bit for a given pair)
    fprintf(f, "# potus_hmc\n#   cus_per_chain = %d\n#   clusters_per_chain = %d\n#   metric_storage = %s\n#   pooled_metric = %d\n", sp->K, sp->sides(), sp->dense && sp->dn.f32 ? "f32" : "f64",
            sp->dense && sp->dn.pooled ? 1 : 0);
    char name[96];
    for (int k = 0; k < ncols; k++) { potus_column_name(&dd, k, name, sizeof name); fprintf(f, k ? ",%s" : "%s", name); }
    fprintf(f, "\n");
  }
  // CmdStan writes the adaptation block when warm-up ends: straight after the header without save_warmup, after the
  // warm-up rows with it.  The text is prepared here, on the calling thread (the dense metric comes off the device), so that
  // the writer threads below only copy it.  dense_e: "# Elements of inverse mass matrix:" and D rows of D values, as CmdStan --
  // up to 2048 parameters; beyond that the matrix is left out TOGETHER WITH its header line (a reader that has seen the
  // header expects D rows; rstan keeps the block as text, potus_get_dense_metric returns the matrix).
  std::vector<std::string> adapt_txt(chains);
  for (int c = 0; c < chains; c++) {
    std::string &t = adapt_txt[c];
    char num[64];
    std::snprintf(num, sizeof num, "# Adaptation terminated\n# Step size = %.6g\n", eps[c]);
    t = num;
    auto put_row = [&](const double *v, int n) {
      t += "# ";
      for (int j = 0; j < n; j++) { std::snprintf(num, sizeof num, j ? ", %.6g" : "%.6g", v[j]); t += num; }
      t += "\n";
    };
    if (sp->dense) {
      if (D <= 2048) {
        std::vector<double> mm((size_t)D * D);
        if ((rc = potus_get_dense_metric(handle, c, mm.data()))) { close_all(); return rc; }
        t += "# Elements of inverse mass matrix:\n";
        for (int i = 0; i < D; i++) put_row(mm.data() + (size_t)i * D, D);
      } else {
        std::snprintf(num, sizeof num, "# (dense inverse metric, %d x %d", D, D);
        t += num;
        t += ": left out of the CSV, see potus_get_dense_metric)\n";
      }
    } else {
      t += "# Diagonal elements of inverse mass matrix:\n";
      put_row(minv.data() + (size_t)c * D, D);
    }
  }
  auto adaptation_block = [&](int c) { fwrite(adapt_txt[c].data(), 1, adapt_txt[c].size(), fp[c]); };
  const int n_warm_rows = sp->opts.save_warmup ? std::min(n_saved, sp->R.num_warmup) : 0;
  if (n_warm_rows == 0) for (int c = 0; c < chains; c++) adaptation_block(c);
  // stream the rows in blocks of draws: rows come back as [iter][chain][ncols]
  const int blk = 32;
  std::vector<double> rows((size_t)blk * chains * ncols);
  Sampler view = *sp; // shallow copy to address a sub-range of iterations
  for (int i0 = 0; i0 < n_saved; i0 += blk) {
    const int nb = std::min(blk, n_saved - i0);
    view.R.draws = sp->R.draws + (size_t)i0 * sp->R.row;
    { DeviceLocks lockb(sp->device); rc = write_array_range(&view, nb, 0, ncols, rows.data()); }
    if (rc) { close_all(); return rc; }
    // one host thread per chain (up to the cores there are) turns its rows into text: "%.6g" through std::to_chars -- the same
    // characters as printf, a third faster, and the 347 M numbers of the 2016 fit (8 x 1000 x 43 360) no longer go through
    // one thread
    const int nthr = std::max(1, std::min(chains, (int)std::thread::hardware_concurrency()));
    std::vector<int> io_err(nthr, 0);
    auto work = [&](int j) {
      std::vector<char> txt((size_t)ncols * 26 + 2);
      for (int c = j; c < chains; c += nthr)
        for (int i = 0; i < nb; i++) {
          const double *r = rows.data() + ((size_t)i * chains + c) * ncols;
          char *q = txt.data();
          for (int k = 0; k < ncols; k++) {
            if (k) *q++ = ',';
            q = std::to_chars(q, q + 25, r[k], std::chars_format::general, 6).ptr;
          }
          *q++ = '\n';
          if (fwrite(txt.data(), 1, (size_t)(q - txt.data()), fp[c]) != (size_t)(q - txt.data())) io_err[j] = 1;
          if (n_warm_rows > 0 && i0 + i == n_warm_rows - 1) adaptation_block(c);
        }
    };
    std::vector<std::thread> pool;
    for (int j = 1; j < nthr; j++) pool.emplace_back(work, j);
    work(0);
    for (auto &t : pool) t.join();
    for (int e : io_err) if (e) { close_all(); return fail(POTUS_ERR_IO, "write to %s/%s-*.csv failed", dir, basename); }
  }
  for (int c = 0; c < chains; c++)
    fprintf(fp[c], "# \n#  Elapsed Time: %.3f seconds (Warm-up)\n#                %.3f seconds (Sampling)\n#                %.3f seconds (Total)\n# \n",
            sp->warm_ms * 1e-3, sp->samp_ms * 1e-3, (sp->warm_ms + sp->samp_ms) * 1e-3);
  close_all();
  return 0;
}

// Development aid (POTUS_PROF builds only; not part of include/potus_hmc.h): per-chain cycle
// counters of the kernel phases.  Returns the number of slots written per chain, 0 otherwise.
int potus_debug_profile(int handle, double *out) {
  Sampler *sp = get(handle);
  if (!sp || !sp->R.prof || !out) return 0;
  (void)hipSetDevice(sp->device);
  if (hipMemcpy(out, sp->R.prof, sizeof(double) * PT_NPROF * sp->R.chains * sp->K * sp->sides(), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return PT_NPROF;
}

// What the last cluster launch found about its placement, per chain and side: 1 = the cluster's members on one XCD (exchange words published with plain stores).
int potus_debug_xcd_local(int handle, int *out) {
  Sampler *sp = get(handle);
  if (!sp || !out || sp->K <= 1) return -1;
  (void)hipSetDevice(sp->device);
  const int blocks = sp->R.chains * sp->sides();
  std::vector<ChainScalars> all((size_t)blocks * sp->K);
  if (hipMemcpy(all.data(), sp->R.scal, sizeof(ChainScalars) * all.size(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (int b = 0; b < blocks; b++) out[b] = all[(size_t)b * sp->K].xcd_local;
  return blocks;
}

// Which build of the cluster pass a handle runs (template tag of potus_cluster.hpp: 4, 8, 12, 16, 17), 0 for one workgroup per chain.
int potus_debug_build_tag(int handle) {
  Sampler *sp = get(handle);
  return sp ? (sp->K > 1 ? sp->cl_dw : 0) : -1;
}

// Development entry points (not in include/potus_hmc.h) that run single pieces of the dense-metric path on caller data,
// so that each can be checked against numpy and timed on its own (tests/test_gpu_dense.py, scripts/micro/dense_probe.py).
namespace {
struct DenseProbe {   // a DnParams with every chain active, owned buffers
  DnParams P{};
  DevBufs bufs;
  int init(int device, int chains, int D, int win_cap, bool pooled = false) {
    HIP_TRY(hipSetDevice(device));
    P.chains = chains; P.D = D; P.LD = (D + 7) & ~7; P.npart = (D + DN_FIN / 4 - 1) / (DN_FIN / 4); P.nblk = (D + DN_RB - 1) / DN_RB; P.sc_stride = 1; P.win_cap = win_cap; P.identity = 0;
    P.pooled = pooled ? 1 : 0;
    const size_t mat = (size_t)(pooled ? 1 : chains) * D * P.LD * 8;
    HIP_TRY(bufs.alloc(&P.state, (size_t)chains * DV_COUNT * P.LD * 8)); HIP_TRY(bufs.alloc(&P.A, mat)); HIP_TRY(bufs.alloc(&P.dg, (size_t)chains * P.LD * 8));
    P.pool_split = 1; P.pool_rows = D;
    // development: POTUS_PROBE_F32 = 1 -- the pooled pieces with metric_storage = f32 (the per-chain pass reads the variable itself, below)
    P.f32 = pooled && getenv("POTUS_PROBE_F32") && atoi(getenv("POTUS_PROBE_F32")) ? 1 : 0;
    if (pooled) {   // (as dense_alloc)
      const int panels = (D + DNP_COLS_OF(P.f32) - 1) / DNP_COLS_OF(P.f32);
      int s = std::max(1, std::min(DNP_SPLIT_MAX, (2 * 256) / panels));   // at most two workgroups per compute unit, all resident at once (measured: profiles/r06_dense_pooled.txt)
      if (const char *e = getenv("POTUS_POOL_SPLIT")) s = std::max(1, std::min(DNP_SPLIT_MAX, atoi(e)));
      P.pool_rows = (((D + s - 1) / s + DNP_KB - 1) / DNP_KB) * DNP_KB;
      P.pool_split = (D + P.pool_rows - 1) / P.pool_rows;
      HIP_TRY(bufs.alloc(&P.Lf, mat)); HIP_TRY(hipMemset(P.Lf, 0, mat));
      HIP_TRY(bufs.alloc(&P.ypool, (size_t)P.pool_split * DNP_RMAX * P.LD * 8)); HIP_TRY(bufs.alloc(&P.pmean, (size_t)P.LD * 8));
      if (P.f32) { HIP_TRY(bufs.alloc(&P.A32, mat / 2)); HIP_TRY(hipMemset(P.A32, 0, mat / 2)); }
      for (const void *f : {reinterpret_cast<const void *>(k_dn_pool_mm<1>), reinterpret_cast<const void *>(k_dn_pool_mm<2>), reinterpret_cast<const void *>(k_dn_pool_mm<3>),
                            reinterpret_cast<const void *>(k_dn_pool_mm<1, true>), reinterpret_cast<const void *>(k_dn_pool_mm<2, true>), reinterpret_cast<const void *>(k_dn_pool_mm<3, true>)})
        HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DNP_LDS(3)));
    }
    dense_launch_shape(P, chains);
    HIP_TRY(bufs.alloc(&P.tpart, (size_t)chains * P.nblk * 3 * P.LD * 8)); HIP_TRY(bufs.alloc(&P.srow, (size_t)chains * 3 * P.ntile * P.LD * 8));
    HIP_TRY(bufs.alloc(&P.win, (size_t)chains * std::max(win_cap, 1) * P.LD * 8)); HIP_TRY(bufs.alloc(&P.partial, (size_t)chains * P.npart * 8));
    HIP_TRY(bufs.alloc(&P.lpbuf, (size_t)chains * 8)); HIP_TRY(bufs.alloc(&P.ts, (size_t)chains * sizeof(TS)));
    HIP_TRY(bufs.alloc(&P.rd, (size_t)chains * sizeof(DnRound))); HIP_TRY(bufs.alloc(&P.active, (size_t)chains * 4)); HIP_TRY(bufs.alloc(&P.fail, 4));
    HIP_TRY(hipMemset(P.state, 0, (size_t)chains * DV_COUNT * P.LD * 8)); HIP_TRY(hipMemset(P.A, 0, mat)); HIP_TRY(hipMemset(P.dg, 0, (size_t)chains * P.LD * 8));
    HIP_TRY(hipMemset(P.fail, 0, 4));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dn_symv<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DN_SYMV_LDS(1)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dn_symv<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DN_SYMV_LDS(2)));
    return 0;
  }
  int set_rounds(const std::vector<DnRound> &r) { HIP_TRY(hipMemcpy(P.rd, r.data(), r.size() * sizeof(DnRound), hipMemcpyHostToDevice)); return 0; }
};
}

// y_r = M^-1 x_r (r < nrhs <= 3) for `chains` dense inverse metrics of size D x D (symmetric; the device keeps the strict
// upper triangle and the diagonal).  Minv_host == NULL: the matrices are generated on the device (k_dn_fill) -- for rates
// at sizes whose matrices would take seconds to upload.  Runs the product `reps` times; x_host / y_host are
// [chains][nrhs][D]; dot_host [chains] receives x_0 . M^-1 x_0; *ms the average time of a pass (HIP events); *pass_bytes
// the bytes of matrix one pass loads per chain.
static int dense_matvec_probe_impl(bool pooled, int device, int chains, int D, int nrhs, const double *Minv_host, const double *x_host, double *y_host, double *dot_host, int reps,
                                   double *ms, long long *pass_bytes) {
  if (chains < 1 || D < 1 || nrhs < 1 || nrhs > 3 || !x_host || !y_host || reps < 1) return fail(POTUS_ERR_ARG, "potus_dense_matvec_probe: bad arguments");
  DenseProbe pr;
  int rc = pr.init(device, chains, D, 1, pooled);
  if (rc) return rc;
  DnParams &P = pr.P;
  if (pooled) {   // ONE full symmetric matrix for every chain (generated: chain 0's of k_dn_fill, mirrored)
    if (Minv_host) {
      HIP_TRY(hipMemcpy2D(P.A, (size_t)P.LD * 8, Minv_host, (size_t)D * 8, (size_t)D * 8, (size_t)D, hipMemcpyHostToDevice));
      if (P.f32) hipLaunchKernelGGL(k_dn_pool_round32, dim3(4096), dim3(256), 0, 0, P);
    } else hipLaunchKernelGGL(k_dn_pool_fill, dim3(4096), dim3(256), 0, 0, P);
  } else if (Minv_host) {
    HIP_TRY(hipMemcpy2D(P.A, (size_t)P.LD * 8, Minv_host, (size_t)D * 8, (size_t)D * 8, (size_t)chains * D, hipMemcpyHostToDevice));   // the lower half is ignored
    std::vector<double> dg((size_t)chains * P.LD, 0.0);
    for (int c = 0; c < chains; c++) for (int i = 0; i < D; i++) dg[(size_t)c * P.LD + i] = Minv_host[((size_t)c * D + i) * D + i];
    HIP_TRY(hipMemcpy(P.dg, dg.data(), dg.size() * 8, hipMemcpyHostToDevice));
  } else {
    // development: POTUS_PROBE_F32 = 1 times the pass over fp32 storage (generated matrices only)
    if (getenv("POTUS_PROBE_F32") && atoi(getenv("POTUS_PROBE_F32"))) {
      P.f32 = 1;
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dn_symv<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DN_SYMV_LDS(1)));
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dn_symv<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DN_SYMV_LDS(2)));
    }
    hipLaunchKernelGGL(k_dn_fill, dim3(4096), dim3(256), 0, 0, P);
  }
  std::vector<DnRound> rds(chains);
  // development: a launch with idle companions (POTUS_PROBE_ACTIVE = n: the first n chains take part; or a list "2,7,11")
  std::vector<int> act_flag(chains, 1);
  int n_act = chains;
  if (const char *e = getenv("POTUS_PROBE_ACTIVE")) {
    std::fill(act_flag.begin(), act_flag.end(), 0);
    if (strchr(e, ',')) { for (const char *q = e; q && *q; q = strchr(q, ',') ? strchr(q, ',') + 1 : nullptr) { const int c = atoi(q); if (c >= 0 && c < chains) act_flag[c] = 1; } }
    else for (int c = 0; c < std::min(chains, std::max(1, atoi(e))); c++) act_flag[c] = 1;
    n_act = (int)std::count(act_flag.begin(), act_flag.end(), 1);
  }
  dense_launch_shape(P, n_act);
  for (int c = 0; c < chains; c++) {
    DnRound &r = rds[c];
    std::memset(&r, 0, sizeof r);
    r.active = act_flag[c];
    for (int k = 0; k < 3; k++) r.job[k] = DnJob{DV_POOLP + k, DV_POOLPS + k, -1, -1, k == 0 ? DV_POOLP : -1, 0, 0.0};
    for (int k = 0; k < nrhs; k++)
      HIP_TRY(hipMemcpy(P.state + ((size_t)c * DV_COUNT + DV_POOLP + k) * P.LD, x_host + ((size_t)c * nrhs + k) * D, (size_t)D * 8, hipMemcpyHostToDevice));
  }
  if ((rc = pr.set_rounds(rds))) return rc;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  DnActive act;
  act.n = 0;
  if (n_act < chains && chains <= DN_ACT_MAX && !getenv("POTUS_PROBE_NO_COMPACTION")) for (int c = 0; c < chains; c++) if (act_flag[c]) act.idx[act.n++] = c;
  dense_symv_launch(0, P, act, nrhs);   // warm-up
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; r++) dense_symv_launch(0, P, act, nrhs);
  (void)hipEventRecord(e1, 0);
  const bool ok = hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess;
  float t = 0; (void)hipEventElapsedTime(&t, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (!ok) return fail(POTUS_ERR_DEVICE, "k_dn_symv failed");
  if (ms) *ms = (double)t / reps;
  if (pass_bytes) *pass_bytes = pooled ? (long long)D * P.LD * (P.f32 ? 4 : 8) * ((n_act * nrhs + DNP_RMAX - 1) / DNP_RMAX)
                                       : dense_pass_bytes(D, P.LD, P.rb) * (nrhs == 3 ? 2 : 1) / (P.f32 ? 2 : 1);
  for (int c = 0; c < chains; c++) {
    for (int k = 0; k < nrhs; k++)
      HIP_TRY(hipMemcpy(y_host + ((size_t)c * nrhs + k) * D, P.state + ((size_t)c * DV_COUNT + DV_POOLPS + k) * P.LD, (size_t)D * 8, hipMemcpyDeviceToHost));
    if (dot_host) {
      std::vector<double> part(P.npart);
      HIP_TRY(hipMemcpy(part.data(), P.partial + (size_t)c * P.npart, (size_t)P.npart * 8, hipMemcpyDeviceToHost));
      double t2 = 0; for (double v : part) t2 += v;
      dot_host[c] = t2;
    }
  }
  return 0;
}

int potus_dense_matvec_probe(int device, int chains, int D, int nrhs, const double *Minv_host, const double *x_host, double *y_host, double *dot_host, int reps,
                             double *ms, long long *pass_bytes) {
  return dense_matvec_probe_impl(false, device, chains, D, nrhs, Minv_host, x_host, y_host, dot_host, reps, ms, pass_bytes);
}
// the same for the POOLED metric (potus_opts.pooled_metric): ONE D x D matrix Minv_host for all chains (NULL: generated), every right-hand side of every
// chain in one pass (k_dn_pool_mm); *pass_bytes = the whole matrix per pass
int potus_dense_pool_matvec_probe(int device, int chains, int D, int nrhs, const double *Minv_host, const double *x_host, double *y_host, double *dot_host, int reps,
                                  double *ms, long long *pass_bytes) {
  return dense_matvec_probe_impl(true, device, chains, D, nrhs, Minv_host, x_host, y_host, dot_host, reps, ms, pass_bytes);
}

// covar_adaptation on caller data: draws [chains][n][D] -> M^-1 (covariance of the window, regularised), its lower
// Cholesky factor (in place, in the lower triangle of the same matrix), and p = L^-T u for u [chains][D] (the momentum
// draw's triangular solve).  Outputs [chains][D][D] (M^-1 rebuilt from the upper triangle and the diagonal vector; L the
// lower triangle, zeros above) / [chains][D]; ms[3] = covariance, factorisation, solve (milliseconds).
static int dense_factor_probe_impl(bool pooled, int device, int chains, int D, int n, const double *draws_host, const double *u_host, double *Minv_host, double *L_host,
                                   double *p_host, double *ms) {
  if (chains < 1 || D < 1 || n < 2 || !draws_host) return fail(POTUS_ERR_ARG, "potus_dense_factor_probe: bad arguments");
  DenseProbe pr;
  int rc = pr.init(device, chains, D, n, pooled);
  if (rc) return rc;
  DnParams &P = pr.P;
  HIP_TRY(hipMemcpy2D(P.win, (size_t)P.LD * 8, draws_host, (size_t)D * 8, (size_t)D * 8, (size_t)chains * n, hipMemcpyHostToDevice));
  std::vector<DnRound> rds(chains);
  for (auto &r : rds) { std::memset(&r, 0, sizeof r); r.active = 1; }
  if ((rc = pr.set_rounds(rds))) return rc;
  hipEvent_t ev[4];
  for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
  const int nb = (D + DN_NB - 1) / DN_NB;
  const dim3 eg((unsigned)std::min((D + 255) / 256, 64), (unsigned)chains);
  (void)hipEventRecord(ev[0], 0);
  if (pooled) {
    hipLaunchKernelGGL(k_dn_pool_center, dim3(eg.x), dim3(256), 0, 0, P, n);
    hipLaunchKernelGGL(k_dn_pool_cov, dim3(nb, nb), dim3(256), 0, 0, P, n);
    hipLaunchKernelGGL(k_dn_pool_scale, dim3(eg.x, (unsigned)D), dim3(256), 0, 0, P, (double)chains * (double)n);
  } else {
    hipLaunchKernelGGL(k_dn_center, eg, dim3(256), 0, 0, P, n);
    hipLaunchKernelGGL(k_dn_cov, dim3(nb, nb, chains), dim3(256), 0, 0, P, n);
  }
  (void)hipEventRecord(ev[1], 0);
  dense_cholesky_launch(0, P, pooled ? 1 : chains);
  (void)hipEventRecord(ev[2], 0);
  if (u_host && p_host) {
    for (int c = 0; c < chains; c++) HIP_TRY(hipMemcpy(P.state + ((size_t)c * DV_COUNT + DV_P0) * P.LD, u_host + (size_t)c * D, (size_t)D * 8, hipMemcpyHostToDevice));
    for (int b = nb - 1; b >= 0; b--) {
      hipLaunchKernelGGL(k_dn_trsv_diag, dim3(chains), dim3(64), 0, 0, P, b);
      if (b > 0) hipLaunchKernelGGL(k_dn_trsv_update, dim3((b * DN_NB + 255) / 256, chains), dim3(256), 0, 0, P, b);
    }
  }
  (void)hipEventRecord(ev[3], 0);
  const bool ok = hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess;
  if (ms) for (int k = 0; k < 3; k++) { float t = 0; (void)hipEventElapsedTime(&t, ev[k], ev[k + 1]); ms[k] = t; }
  for (auto &e : ev) (void)hipEventDestroy(e);
  if (!ok) return fail(POTUS_ERR_DEVICE, "dense factor kernels failed");
  int failed = 0;
  HIP_TRY(hipMemcpy(&failed, P.fail, 4, hipMemcpyDeviceToHost));
  if (failed) return fail(POTUS_ERR_STATE, "covariance not positive definite");
  if (Minv_host || L_host) {
    std::vector<double> buf((size_t)D * D), dg(D), lbuf;
    for (int c = 0; c < (pooled ? 1 : chains); c++) {   // (pooled: ONE matrix and ONE factor come back)
      HIP_TRY(hipMemcpy2D(buf.data(), (size_t)D * 8, dn_mat(P, c), (size_t)P.LD * 8, (size_t)D * 8, (size_t)D, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(dg.data(), dn_diag(P, c), (size_t)D * 8, hipMemcpyDeviceToHost));
      if (pooled) { lbuf.resize((size_t)D * D); HIP_TRY(hipMemcpy2D(lbuf.data(), (size_t)D * 8, P.Lf, (size_t)P.LD * 8, (size_t)D * 8, (size_t)D, hipMemcpyDeviceToHost)); }
      const std::vector<double> &lb = pooled ? lbuf : buf;
      for (size_t i = 0; i < (size_t)D; i++)
        for (size_t j = 0; j < (size_t)D; j++) {
          if (Minv_host) Minv_host[((size_t)c * D + i) * D + j] = pooled ? buf[i * D + j] : (i == j ? dg[i] : (i < j ? buf[i * D + j] : buf[j * D + i]));
          if (L_host) L_host[((size_t)c * D + i) * D + j] = j <= i ? lb[i * D + j] : 0.0;
        }
    }
  }
  if (u_host && p_host)
    for (int c = 0; c < chains; c++) HIP_TRY(hipMemcpy(p_host + (size_t)c * D, P.state + ((size_t)c * DV_COUNT + DV_P0) * P.LD, (size_t)D * 8, hipMemcpyDeviceToHost));
  return 0;
}
int potus_dense_factor_probe(int device, int chains, int D, int n, const double *draws_host, const double *u_host, double *Minv_host, double *L_host,
                             double *p_host, double *ms) {
  return dense_factor_probe_impl(false, device, chains, D, n, draws_host, u_host, Minv_host, L_host, p_host, ms);
}
// the POOLED window end: ONE regularised covariance of all chains x n draws -> Minv_host [D][D] (as the device holds it: the full symmetric matrix),
// ONE factor L_host [D][D], and p = L^-T u for every chain's u
int potus_dense_pool_factor_probe(int device, int chains, int D, int n, const double *draws_host, const double *u_host, double *Minv_host, double *L_host,
                                  double *p_host, double *ms) {
  return dense_factor_probe_impl(true, device, chains, D, n, draws_host, u_host, Minv_host, L_host, p_host, ms);
}

// Development aid: the whole state block [chains][V_COUNT][Dpad] (internal element order) and every replica of the
// chain scalars as raw bytes.  which = 0: sizes only (out[0] = V_COUNT, out[1] = Dpad, out[2] = sizeof scalars * replicas).
int potus_debug_state(int handle, int which, double *out, unsigned char *scal) {
  Sampler *sp = get(handle);
  if (!sp || !out) return 0;
  (void)hipSetDevice(sp->device);
  const size_t nblk = (size_t)sp->R.chains * sp->sides();   // (two clusters / workgroups per chain: side 1's blocks follow side 0's)
  if (which == 0) { out[0] = V_COUNT; out[1] = sp->R.Dpad; out[2] = (double)(sizeof(ChainScalars) * nblk * sp->K); return 1; }
  if (hipMemcpy(out, sp->R.state, sizeof(double) * V_COUNT * sp->R.Dpad * nblk, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  if (scal && hipMemcpy(scal, sp->R.scal, sizeof(ChainScalars) * nblk * sp->K, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return 1;
}

// ---------------------------------------------------------------- .C() wrappers
void potus_R_create(int *dims, int *state, int *day_state, int *day_national, int *poll_state, int *poll_national,
                    int *poll_mode_state, int *poll_mode_national, int *poll_pop_state, int *poll_pop_national,
                    int *n_democrat_national, int *n_two_share_national, int *n_democrat_state, int *n_two_share_state,
                    double *unadjusted_national, double *unadjusted_state, double *mu_b_prior, double *state_weights,
                    double *scalars, double *state_covariance_0, int *iopts, double *dopts, int *handle, int *status) {
  potus_data d{};
  d.N_national_polls = dims[0]; d.N_state_polls = dims[1]; d.T = dims[2]; d.S = dims[3]; d.P = dims[4]; d.M = dims[5]; d.Pop = dims[6];
  d.variant = dims[7];
  d.state = state; d.day_state = day_state; d.day_national = day_national; d.poll_state = poll_state; d.poll_national = poll_national;
  d.poll_mode_state = poll_mode_state; d.poll_mode_national = poll_mode_national; d.poll_pop_state = poll_pop_state;
  d.poll_pop_national = poll_pop_national; d.n_democrat_national = n_democrat_national; d.n_two_share_national = n_two_share_national;
  d.n_democrat_state = n_democrat_state; d.n_two_share_state = n_two_share_state; d.unadjusted_national = unadjusted_national;
  d.unadjusted_state = unadjusted_state; d.mu_b_prior = mu_b_prior; d.state_weights = state_weights;
  d.sigma_c = scalars[0]; d.sigma_m = scalars[1]; d.sigma_pop = scalars[2]; d.sigma_measure_noise_national = scalars[3];
  d.sigma_measure_noise_state = scalars[4]; d.sigma_e_bias = scalars[5]; d.random_walk_scale = scalars[6]; d.mu_b_T_scale = scalars[7];
  d.polling_bias_scale = scalars[8]; d.state_covariance_0 = state_covariance_0;
  potus_opts o; potus_default_opts(&o);
  o.chains = iopts[0]; o.chain_id_offset = iopts[1]; o.num_warmup = iopts[2]; o.num_samples = iopts[3]; o.max_depth = iopts[4];
  o.device = iopts[5]; o.save_warmup = iopts[6]; o.cus_per_chain = iopts[7]; o.metric = iopts[8]; o.twin = iopts[9]; o.metric_storage = iopts[10]; o.pooled_metric = iopts[11];
  o.delta = dopts[0]; o.gamma = dopts[1]; o.kappa = dopts[2]; o.t0 = dopts[3]; o.stepsize = dopts[4]; o.init_radius = dopts[5];
  if (!(dopts[6] >= 0) || dopts[6] > 9007199254740992.0 || dopts[6] != std::floor(dopts[6])) { *status = fail(POTUS_ERR_ARG, "seed must be a non-negative integer below 2^53"); return; }
  o.seed = (uint64_t)dopts[6];   // R integers are 32 bits wide: the seed travels as a double (exact to 2^53)
  *status = potus_create(&d, &o, handle);
}
void potus_R_init(int *handle, int *status) { *status = potus_init(*handle, nullptr); }
void potus_R_run(int *handle, int *n_iter, int *status) { *status = potus_run(*handle, *n_iter); }
void potus_R_run_many(int *handles, int *n_handles, int *n_iter, int *status) { *status = potus_run_many(handles, *n_handles, *n_iter); }
void potus_R_num_columns(int *handle, int *D, int *n_cols, int *status) {
  Sampler *sp = get(*handle);
  if (!sp) { *status = fail(POTUS_ERR_STATE, "bad handle"); return; }
  *D = sp->L.D; *n_cols = sp->L.ncols; *status = 0;
}
void potus_R_write_array(int *handle, int *col_begin, int *col_end, double *out, int *status) { *status = potus_write_array(*handle, *col_begin, *col_end, out); }
void potus_R_write_stan_csv(int *handle, char **dir, char **basename, int *status) { *status = potus_write_stan_csv(*handle, dir[0], basename[0]); }
void potus_R_posterior_summary(int *handles, int *n_handles, double *ev, double *state_out, double *natl_out, double *ev_out, int *status) {
  *status = potus_posterior_summary_many(handles, *n_handles, ev, state_out, natl_out, ev_out);
}
void potus_R_diagnostics(int *handles, int *n_handles, int *cols /*[2]: col_begin, col_end*/, double *rhat_out, double *ess_bulk_out, int *status) {
  *status = potus_diagnostics(handles, *n_handles, cols[0], cols[1], rhat_out, ess_bulk_out);
}
void potus_R_check_convergence(int *handles, int *n_handles, double *limits /*[2]: rhat_below, ess_at_least*/, int *converged, double *out /*[2]: rhat_max, ess_bulk_min*/, int *status) {
  *status = potus_check_convergence(handles, *n_handles, limits[0], limits[1], converged, out, out + 1);
}
void potus_R_backtest_scores(double *state_out, int *dims /*[3]: T, S, day*/, double *ev, int *won, double *out, int *status) {
  *status = potus_backtest_scores(state_out, dims[0], dims[1], dims[2], ev, won, out);
}
void potus_R_saved_count(int *handle, int *n_saved, int *status) { *status = potus_get_draws(*handle, nullptr, n_saved); }
void potus_R_last_error(char **buf, int *len) { potus_last_error(buf[0], *len); }
void potus_R_destroy(int *handle, int *status) { *status = potus_destroy(*handle); }

} // extern "C"
