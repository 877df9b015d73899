// potus_nuts_twin.hpp -- two workgroups per chain, one per end of the NUTS trajectory, for the one-workgroup-per-chain sampler.
//
// With 65-128 chains on a device of 256 compute units the one-workgroup sampler (potus_nuts.hpp) leaves up to half of the
// chip idle and clusters of 4 no longer fit.  The two ends of a trajectory do not depend on each other (potus_cluster.hpp,
// "twin" mode, has the argument and the protocol): side 1 integrates the forward doublings, side 0 the backward ones, each
// in its own state block (block chain + side * chains of RunParams::state / scal); the bookkeeping of transition() -- accept
// step of the new subtree, rho over the whole trajectory, the U-turn checks across it -- is a "combine" taken strictly in
// doubling order by the side that built the subtree, with the trajectory-level state travelling as TT_N tagged words through
// the chain's mailbox (RunParams::twbuf).  This file is that protocol on top of potus_nuts.hpp's tree loop; the mailbox
// helpers (tw_*), the word layout (TT_*, TWB_*) and the watchdog are potus_cluster.hpp's.
// Same arithmetic in the same order as the one-workgroup sampler (which sums the metropolis terms per doubling for that
// reason): the draws of the two are the same bytes.
// Reference: the algorithm is Stan 2.24's base_nuts.hpp::transition, as restated in potus_nuts.hpp.
#pragma once
#include "potus_nuts.hpp"
#include "potus_cluster.hpp"

// the watchdog word of a block (8 words of 16 bytes per block in RunParams::xbuf; no exchange slots: K = 1, XW = 0)
__device__ __forceinline__ Xch tw1_watch(CRp R, int blk, unsigned launch) {
  Xch x;
  x.xb = make_rsrc(uni_ptr(R->xbuf + (size_t)blk * 16), 128u);
  x.epoch = 0; x.launch = uni32(launch); x.x1e = 0; x.K = 1; x.m = 0; x.XW = 0; x.local = 0;
  return x;
}

// copy of a whole vector of the state block; SD / SS: the destination is read by the other side (write-through) / the source
// was stored write-through
template <bool SD, bool SS>
__device__ __forceinline__ void vop_copy_s(const Chain &c, unsigned s_dst, unsigned s_src) {
  const int tid0 = fresh_tid(c);
  for (int base = tid0; base < c.D; base += PT_UNR * PT_THREADS) {
    double v[PT_UNR];
#pragma unroll
    for (int k = 0; k < PT_UNR; k++) {
      const int i = base + k * PT_THREADS;
      v[k] = SS ? bld_s(c.st, i < c.D ? 8u * i : PT_OOB, s_src) : bld(c.st, i < c.D ? 8u * i : PT_OOB, s_src);
    }
#pragma unroll
    for (int k = 0; k < PT_UNR; k++) {
      const int i = base + k * PT_THREADS;
      if (SD) bst_s(c.st, i < c.D ? 8u * i : PT_OOB, s_dst, v[k]);
      else bst(c.st, i < c.D ? 8u * i : PT_OOB, s_dst, v[k]);
    }
  }
  __syncthreads();
}

struct Tw1Args { const DevModel *Mg; const RunParams *Rg; int chain, side; unsigned launch; };

__device__ __noinline__ void cold_twin1_combine(const DevModel *Mg, const RunParams *Rg, int chain_, int side_, unsigned launch_, uint32_t iter_,
                                                int depth_, int valid_);

// The doublings of one transition that go this side's way, each followed by its combine.
__device__ __forceinline__ void transition_tree_twin(const Chain &c, const Xch &x, const Tw1Args &ta, uint32_t iter) {
  ltp ts = c.ts;
  const int tid = c.tid, side = ta.side;
  const double eps = ts->eps;
  int depth = 0;
  if (tid < 64) {
    // the directions of every doubling of this transition, and the trajectory-level state before the first one
    const unsigned long long fw = __ballot(tid < c.max_depth && rng_uniform(c.key, iter, RNG_DIRECTION, 0, (uint32_t)(tid < c.max_depth ? tid : 0)) > 0.5);
    if (tid == 0) {
      const int id = ts->sample_qid;
      ts->tw_dirs = (int)(unsigned)fw; ts->tw_seq = 0; ts->tw_over = 0; ts->tw_keep = id;
      ts->tt[TT_STOP] = 0.0; ts->tt[TT_DEPTH] = 0.0; ts->tt[TT_LSW] = 0.0; ts->tt[TT_SSIDE] = -1.0; ts->tt[TT_SSLOT] = (double)id;
      ts->tt[TT_SLP] = ts->q_lp[id]; ts->tt[TT_SH] = ts->q_h[id]; ts->tt[TT_METRO] = 0.0; ts->tt[TT_NLEAP] = 0.0; ts->tt[TT_DIV] = 0.0;
      ts->tt[TT_RHOSIDE] = -1.0; ts->tt[TT_N - 1] = 0.0;
    }
  }
  while (true) {
    __syncthreads();
    if (uni_i(ts->tw_over) || uni_i(cl_dead)) break;
    const int dirs = uni_i(ts->tw_dirs);
    while (depth < c.max_depth && ((dirs >> depth) & 1) != side) depth++;   // the doublings of the other end are not this side's business
    if (depth >= c.max_depth) break;
    // bounded speculation (potus_cluster.hpp): a doubling none of the last four transitions needed waits for its turn
    const unsigned sl = (unsigned)uni_i(c.sc->spec_limit);
    const int spec_limit = (int)max(max(sl & 0xffu, (sl >> 8) & 0xffu), max((sl >> 16) & 0xffu, sl >> 24));
    if (depth >= spec_limit && uni_i(ts->tw_seq) < depth) {
      if (tid < 64) {
        CRp R = (CRp)uni_ptr(ta.Rg);
        const Twin t = make_twin(R, ta.chain, side, x.launch, iter);
        tw_catch_up(x, t, ts, depth, true);
      }
      __syncthreads();
      if (uni_i((int)ts->tt[TT_STOP])) {
        if (tid == 0) ts->tw_over = 1;
        continue;                                   // the loop head leaves
      }
    }
    if (tid == 0) {
      ts->dir = side;
      ts->pmask = 0;
      ts->qmask = 1u << ts->tw_keep;
      ts->sum_metro = 0.0; ts->n_leap = 0; ts->divergent = 0;     // of this subtree; added to the trajectory's at the combine
    }
    __syncthreads();
    const int dir = side;
    vop_copy_s<false, true>(c, c.soff(V_PNEAR), c.soff(V_PF0 + dir));   // (the end momentum was stored write-through by the last combine)
    bool valid = true;
    const int nleaf = 1 << depth;
    for (int n = 0; n < nleaf; n++) {
      if (tid == 0) { unsigned pm = ts->pmask; ts->leaf_id = pool_alloc(pm, PT_NPP); ts->pmask = pm; }
      __syncthreads();
      if (tid >= PT_THREADS - 64) {
        // has the trajectory ended at the other end?  One look per leaf, by this wave while the others start the pass
        CRp R = (CRp)uni_ptr(ta.Rg);
        const Twin t = make_twin(R, ta.chain, side, x.launch, iter);
        double sv;
        const bool over = tw_try(t, TWB_STOP, 1, t.ittag, sv);
        if (tid == PT_THREADS - 64) ts->tw_ext = over ? 1 : 0;
      }
      const double e = dir ? eps : -eps;
      const int sel = ts->qsel[dir];              // buffer holding this leaf's position
      const unsigned s_leaf = c.soff(V_POOLP + ts->leaf_id);
      const int pair = n & 1, m_leaf = __builtin_ctz(~(unsigned)n);
      LeapPolicy lp{c.st, c.soff((sel ? V_QB0 : V_QA0) + dir), c.soff((sel ? V_QA0 : V_QB0) + dir), c.soff(V_PH0 + dir), c.soff(V_MINV),
                    s_leaf, 0.5 * e, e, c.soff(V_POOLP + (pair ? ts->pend_beg[0] : ts->leaf_id)), m_leaf == 1 ? c.soff(V_RHOLEV + 1) : c.soff(V_SCR1),
                    pair, {0.0, 0.0, 0.0}};
      const double lpv = model_pass(c.M, c.lds, c.pst, lp);
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
        ts->abort = div | ts->tw_ext;             // (a trajectory that is over: this subtree is dropped at its combine)
        ts->m = __builtin_ctz(~(unsigned)n);
        ts->qsel[dir] = sel ^ 1;                  // the next leaf of this end reads the buffer just written
      }
      __syncthreads();
      if (ts->abort) { valid = false; break; }
      const int m = ts->m;
      for (int j = 1; j <= m; j++) {
        const int ib = ts->pend_beg[j - 1], ie = ts->pend_end[j - 1], cb = ts->cur_beg, ce = ts->cur_end;
        const unsigned a_rho = j == 1 ? c.soff(V_POOLP + ib) : c.soff(V_RHOLEV + j - 1);
        const unsigned b_rho = j == 1 ? c.soff(V_POOLP + cb) : c.soff(V_SCR0 + ((j - 1) & 1));
        const unsigned out = j == m ? c.soff(V_RHOLEV + j) : c.soff(V_SCR0 + (j & 1));
        const bool persist = j == 1 ? (lp.extra[1] > 0 && lp.extra[2] > 0)      // (level 1 came out of the leaf's epilogue)
                                    : vop_merge(c, c.soff(V_POOLP + ib), c.soff(V_POOLP + ie), a_rho, c.soff(V_POOLP + cb), c.soff(V_POOLP + ce), b_rho, out);
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
    }
    cold_twin1_combine(ta.Mg, ta.Rg, ta.chain, side, ta.launch, iter, depth, valid ? 1 : 0);
    depth++;
  }
  __syncthreads();
}
