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
// An exchange word travels as 16 bytes {value, tag = (launch id, exchange number)} written with one 16-byte store -- write-through
// (sc1) in general, plain when the launch has found every member of the cluster on one XCD (cl_find_local: a plain store reaches
// that XCD's L2, where the sc1 loads of its compute units see it, in 0.23 us instead of 0.55; it is invisible to another XCD) --;
// a reader re-loads (sc1) until the tag is the one it expects -- no counter, no flag, no barrier on the consumer side (struct Xch
// below; scripts/micro/pingpong.hip).  All members sum the
// partials in the same fixed order, so they hold bit-identical scalars and run the NUTS control
// flow redundantly without ever diverging; results are reproducible run to run for a given K.
//
// Launch: grid = chains * K, block b -> chain b % chains, member b / chains, so that with 8 chains
// the members of a chain land on one XCD (blocks are dealt round-robin to the 8 XCDs) and share
// its L2.  All blocks must be co-resident (grid <= number of CUs; checked on the host).
#pragma once
#include "potus_nuts.hpp"
#include "potus_dpp.hpp"

// Days per wave (DW) is a compile-time parameter of the pass: 4 (members of at most 32 days, e.g. 2016 with
// K = 16) keeps the per-day register arrays small enough not to spill; 8 covers up to 64 days per member.
#define CL_MAXDAYS 64                    // LDS sizing: days per member at DW = 8
#define CL_MAXK 32
#define CL_DW4_MAXAVG 26                 // days per member on average up to which the 4-days-per-wave build of the pass is used
#define CL_AUX_SC1 16                    // cache-policy bit of the buffer intrinsics: sc1 (agent scope)
#define CL_NCHUNK PT_NW                  // chunks the member's polls are cut into for the adjoint gather (host: build_cluster)
#define CL_SEG_SHIFT 64                  // the level-2 segment sums of phase E start at this thread: wave 0 turns the chunk totals into prefixes there
#define CL_SPIN_LIMIT 8000000u
#ifndef CL_SPIN_SLEEP
#define CL_SPIN_SLEEP 1                  // s_sleep argument between two looks at an exchange word that has not arrived
#endif

// fields of one member's part descriptor (ints)
enum { CP_D0 = 0, CP_ND, CP_P0, CP_NP, CP_E0, CP_NE, CP_R0, CP_NR, CP_NSUB, CP_NSEG, CP_WB,
       CP_O_WD, CP_O_MASK, CP_O_SUB, CP_O_SEGPTR, CP_O_SEGKIND, CP_O_SEGIDX, CP_O_WT, CP_E_SH, CP_NCELL, CP_O_CELL, CP_N = 24 };
#define CL_CELLS_PER_THREAD 2            // (state, day) cells of a member's polls per thread in the adjoint scatter (<= 1024 cells)
#define CL_G_PAD 16                      // spare doubles behind G: the dump slot of idle scatter threads
// payload layout of exchange X2 (doubles); X1 and the scalar all-reduces use the first words
enum { XP_PRE = 0, XP_AR = 64, XP_S = 66, XP_P = 72 };
// the exchange sent ahead from the epilogue: words [0, S) suffix totals of the next position (X1), [XQ0, XQ0 + NREP) the
// next position of the small vectors themselves
enum { XQ0 = 64 };

struct ClModel {
  int K, XW, NR, NREP, NDP, npmax, nsubmax, Dint;   // Dint: internal length of a vector (parameters + line padding)
  const int *part;          // [K][CP_N]
  const int *sched;         // per member: wd_a|wd_b [PT_THREADS], wd0|wnd [PT_NW], sub16, seg_ptr, seg_kind, seg_index
  const double *wt;         // weights of the weighted level-1 tasks
  const int *rep_pos;       // [NREP] internal index of the small parameters every member reads: zT | zb | c,m,pop,mue,rho,ze
  const int *rep_owner;     // [NREP] member that owns each of them
  const double *rep_scale;  // [NR] scale of owned slot r (sigma_c ... ; 1 for zT, zb)
  const int *perm;          // [D] internal index -> Stan index
  int GS, GROWS;            // adjoint on the matrix cores (cl_adjoint_mfma): G[pseudo-state][local day], row stride, rows (l_G = 0: gather walk)
  int l_C, l_G, l_Lw, l_prior, l_pm, l_py, l_pun, l_sub, l_tab, l_ru, l_wide, l_wout, l_X, l_Y, l_r, l_rep, l_bT, l_e, l_c1, l_c2, l_c3, l_ge, l_P, l_scal, l_red, l_st, l_prof;
  int lds_doubles;
};
typedef const ClModel AS_C *CCp;

#define CL_WIDE 96                       // widest all-reduce of a leaf (cl_wide_publish): 2 + 6 * (levels + 1) + 1 words
// A member's LDS layout (offsets in doubles) as ONE function of the capacities, used by the host for the dynamic builds
// (build_cluster: the posterior's own sizes) and at compile time for the fixed build below -- the two cannot drift apart.
struct ClLay {
  int l_C, l_G, l_Lw, l_prior, l_pm, l_py, l_pun, l_sub, l_tab, l_ru, l_wide, l_wout, l_X, l_Y, l_r, l_rep, l_bT, l_e, l_c1, l_c2, l_c3, l_ge, l_P, l_scal, l_red, l_st, l_prof;
  int l_rm, l_rpf, l_rrs;   // the member's share of three vectors, resident across the leaves of a doubling (fixed builds; see ClLeapPolicyRes)
  int total;
};
constexpr int cl_ev(int n) { return (n + 1) & ~1; }   // LDS blocks start on 16-byte boundaries
constexpr ClLay cl_layout(int S, int SE, int SP, int NDP, int npcap, int nsubcap, int nrepcap, int nrcap, int tcap, int g_doubles, int necap = 0) {
  ClLay L{};
  int o = 0;
  L.l_C = o; o += cl_ev(S * NDP);                         // C[state][local day]: suffix sums, then the adjoint's running sums
  L.l_G = g_doubles ? o : 0; o += cl_ev(g_doubles);       // G[pseudo-state][local day] (matrix-core build only)
  L.l_Lw = o; o += cl_ev((SE + 1) * SP);                  // the walk's factor, the row v = L_W' w and a zero row
  L.l_prior = o; o += cl_ev(SE);
  L.l_pm = o; o += cl_ev(npcap + 8);                      // per-poll constants: {state, day, pollster, mode, population}
  L.l_py = o; o += cl_ev(npcap + 8);                      // {y, N} as two int32
  L.l_pun = o; o += cl_ev(npcap + 8);
  L.l_sub = o; o += cl_ev(nsubcap * 4 + 4);               // level-1 task lists, 16-bit entries
  L.l_tab = o; o += cl_ev((npcap + 64) / 2 + 2);          // gather program words
  L.l_ru = o; o += cl_ev(npcap + 2);
  L.l_wide = o; o += cl_ev(CL_WIDE * PT_NW);
  L.l_wout = o; o += cl_ev(CL_WIDE);
  L.l_X = o; o += cl_ev((PT_NW + 1) * SE);
  L.l_Y = o; o += cl_ev((PT_NW + 1) * SE > nsubcap ? (PT_NW + 1) * SE : nsubcap);
  L.l_r = o; o += cl_ev(npcap + 2);
  L.l_rep = o; o += cl_ev(nrepcap + 2);
  L.l_bT = o; o += cl_ev(SE);
  L.l_e = o; o += cl_ev(tcap);
  L.l_c1 = o; o += cl_ev(CL_MAXDAYS);                     // tangent recurrences of the AR(1) bias: the member's own days only
  L.l_c2 = o; o += cl_ev(CL_MAXDAYS);
  L.l_c3 = o; o += cl_ev(CL_MAXDAYS);
  L.l_ge = o; o += cl_ev(CL_MAXDAYS);
  L.l_P = o; o += cl_ev(nrcap + 8);
  L.l_scal = o; o += cl_ev(SC_N);
  L.l_red = o; o += cl_ev((PT_NW + 1) * PT_NRED);
  L.l_st = o; o += cl_ev((npcap + 8 + 7) / 8);
  L.l_prof = o; o += cl_ev(PT_NPROF);
  L.l_rm = necap ? o : 0; o += necap ? cl_ev(necap + 8) : 0;       // [necap] + a dump slot (index necap) for masked lanes
  L.l_rpf = necap ? o : 0; o += necap ? cl_ev(necap + 8) : 0;
  L.l_rrs = necap ? o : 0; o += necap ? cl_ev(necap + 8) : 0;
  L.total = o;
  return L;
}

// The fourth build of the pass (tag 16): what the reference's posteriors look like on clusters of 16 -- 51 states, at most 32 days
// and 256 polls per member -- with the LDS LAYOUT FIXED AT COMPILE TIME.  Every LDS address of the pass is then an immediate
// (DS instructions carry a 16-bit offset) instead of a base kept in a scalar register, and the state count is a constant: the
// kernel keeps ~150 wave-uniform values alive for 102 scalar registers, and one instruction in ten of the dynamic build
// restores one of them from a vector-register lane.  The host lays the member's LDS out with the same function and these
// capacities (build_cluster) when the model fits them; anything else takes the dynamic builds.
struct ClFixed {
  static constexpr int S = 51, SE = 52, SP = 51, NDP = 33;
  static constexpr int NPCAP = 256, NSUBCAP = 384, NREPCAP = 768, NRCAP = 512, TCAP = 320;   // polls, level-1 tasks per member; small parameters; slots; days
  static constexpr int KMAX = 16;                    // members per cluster: every loop over the members is a single batch of sixteen tagged words
  static constexpr int NECAP = 1920;                 // elements of a vector per member (CP_NE) the resident share holds: 32 days x 51 states + 256 polls + slots (2016 on 16 members: 1792)
  static constexpr int XW = 832;                     // exchange words per member: max(XP_P + NRCAP, XQ0 + NREPCAP) = max(72 + 512, 64 + 768)
  // (the model variant is the tag's: 16 = poll_model_2020.stan with mode / population effects and the AR(1) bias, 17 = the no_mode_adjustment variant)
  static constexpr int GS = 48, GROWS = 52;
  static constexpr ClLay L = cl_layout(S, SE, SP, NDP, NPCAP, NSUBCAP, NREPCAP, NRCAP, TCAP, 0, NECAP);
#define CLF(f) static constexpr int f = L.f
  CLF(l_C); CLF(l_G); CLF(l_Lw); CLF(l_prior); CLF(l_pm); CLF(l_py); CLF(l_pun); CLF(l_sub); CLF(l_tab); CLF(l_ru); CLF(l_wide); CLF(l_wout); CLF(l_X); CLF(l_Y);
  CLF(l_r); CLF(l_rep); CLF(l_bT); CLF(l_e); CLF(l_c1); CLF(l_c2); CLF(l_c3); CLF(l_ge); CLF(l_P); CLF(l_scal); CLF(l_red); CLF(l_st); CLF(l_prof); CLF(l_rm); CLF(l_rpf); CLF(l_rrs);
#undef CLF
  static constexpr int lds_doubles = L.total;
};
template <int TAG> struct ClTag {
  static constexpr int DW = (TAG == 12 || TAG == 16 || TAG == 17) ? 4 : TAG;   // days per wave
  static constexpr bool MF = TAG == 12;                               // adjoint product on the matrix cores
  static constexpr bool FX = TAG == 16 || TAG == 17;                  // LDS layout and state count fixed at compile time ...
  static constexpr int FULL = TAG == 16 ? 1 : 0;                      // ... and the model variant: 16 = poll_model_2020.stan, 17 = poll_model_2020_no_mode_adjustment.stan
};
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
// Every payload word travels as 16 bytes {value, tag}: tag = (launch id, exchange number).  A reader
// simply re-loads until the tag is the one it expects, so publishing is a fire-and-forget store and
// there is no counter, no separate flag and no barrier on the consumer side.  Payload buffers rotate
// over four slots (exchange number mod 4): the slot of exchange e is rewritten by e+4.  With pipelined leaves a pass
// owns three exchanges -- X2(n) = e, the X1 / small parameters sent ahead for pass n+1 = e+1, X3(n) = e+2 -- and the
// readers are: X2(n) in pass n phase F; e+1 in pass n+1 phases A and B; X3(n) in pass n+1 phase C.  The writer of e+4
// (the ahead-exchange of pass n+1, sent from its epilogue) has by then taken the verdicts of leaf n, for which it needed
// X3(n) of EVERY member; a member sends its X3(n) after its own pass n, i.e. after it has read X2(n).  Likewise e+5 =
// X3(n+1) follows the writer's own reads of e+1, and the other members read e+1 before they can send the X3(n+1)
// ... that the writer of e+5's successors wait for.  Spelled out per word range: X1 words of member m are read by
// the members before m, which send their X2 (needed by m's prefix) afterwards; the small-parameter words are read by
// all members in phase A, before their X3 of that pass; X3 words are read by all in phase C, before their X2-consuming
// phase F of the same pass completes and hence before anybody's next X3.  Exchanges outside the tree (all-reduces of
// the begin / end code) are all-to-all and consumed at once.  A missed assumption would not corrupt data silently:
// a word overwritten too early carries a newer tag and the reader spins until the watchdog ends the launch.
// Watchdog.  The members of a cluster wait for each other, so they must all be resident; the host checks that for what
// it launches itself (potus_create, potus_run_many), but it cannot see other processes on the GPU.  A wave that has
// waited CL_SPIN_LIMIT rounds for a word gives up: it raises the chain's watchdog word (the 16 bytes behind the
// exchange slots, polled by every other spinning wave of the cluster), sets the workgroup's cl_dead flag and ends.
// Ended waves no longer count at barriers; every loop of the sampler is bounded or tests cl_dead, so the launch
// drains in about a second, the chain scalars get status POTUS_ERR_WATCHDOG and potus_run returns that error --
// instead of a trap, which would leave the whole process with a sticky HIP error.
__shared__ int cl_dead;
__shared__ int cl_local;   // the members of this cluster sit on one XCD (set by k_cl_run once per launch, Xch::local)
// A buffer store of more than 64 bits reads its data registers AFTER it has issued, and hipcc 7.2 pads the pair "store, VALU write
// of a data register" only when the store has no register soffset (LLVM's createsVALUHazard) -- every store of an exchange word
// has one.  Found in round 4: once the 51 x 51 mat-vecs had left the pass, the word of a slot partial was followed directly by
// `v_cndmask v2, ...` (the next LDS address, in the register that held the value's low dword) and one word in ~10^5 arrived with a
// foreign low dword: a gradient wrong in its ninth digit, runs no longer reproducible.  The wait states are written by hand, in a
// statement that names the data registers so that nothing can be scheduled onto them before it;
// scripts/check_store_hazard.py scans every build for the shape (tests/test_code_object.py).
#define STORE128_PAD(w) asm volatile("s_nop 1" :: "v"(w) : "memory")
struct Xch {
  rsrc_t xb;               // the chain's exchange buffer [4][K][XW] words of 16 bytes (+ the watchdog word)
  unsigned epoch;          // exchanges published so far in this launch (identical in every member)
  unsigned launch;         // launch id (host counter): stale words of earlier launches never match
  unsigned x1e;            // number of an X1 published ahead for the next pass (0 = none), see cl_pass_partial
  int K, m, XW;
  int local;               // 1: every member of the cluster runs on the same XCD -- exchange words are published with plain stores (see xst)
};
// byte offset of member mm's payload: w = the exchange being assembled (number epoch+1), r = the one
// just published (call after epoch++)
__device__ __forceinline__ unsigned xch_wslot(const Xch &x, int mm) { return ((((x.epoch + 1u) & 3u) * (unsigned)x.K + (unsigned)mm) * (unsigned)x.XW) * 16u; }
__device__ __forceinline__ unsigned xch_rslot(const Xch &x, int mm) { return (((x.epoch & 3u) * (unsigned)x.K + (unsigned)mm) * (unsigned)x.XW) * 16u; }
__device__ __forceinline__ unsigned xch_eslot(const Xch &x, unsigned e, int mm) { return (((e & 3u) * (unsigned)x.K + (unsigned)mm) * (unsigned)x.XW) * 16u; }
__device__ __forceinline__ unsigned xch_wdoff(const Xch &x) { return 4u * (unsigned)x.K * (unsigned)x.XW * 16u; }
__device__ __forceinline__ bool xch_watchdog_raised(const Xch &x) {
  return __builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(x.xb, 0u, __builtin_amdgcn_readfirstlane(xch_wdoff(x)), CL_AUX_SC1)) != 0u;
}
__device__ __forceinline__ void xch_give_up(const Xch &x) {
  __builtin_amdgcn_raw_buffer_store_b32(1u, x.xb, 0u, __builtin_amdgcn_readfirstlane(xch_wdoff(x)), CL_AUX_SC1);
  cl_dead = 1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_endpgm();
}
// publish one word of the exchange being assembled (voff = 16 * word, or PT_OOB for idle lanes)
__device__ __forceinline__ void xst(const Xch &x, unsigned voff, double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const u32x4 w = {(unsigned)u, (unsigned)(u >> 32), x.epoch + 1u, x.launch};
  // (soffset through readfirstlane: the exchange counter ends up in a vector register wherever it was updated under a
  // branch whose condition came out of LDS, and a "divergent" scalar offset costs a waterfall loop around every access)
  // A write-through (sc1) store is acknowledged by the memory side; a plain one is visible to the sc1 loads of the same XCD as soon as it is in that XCD's L2
  // (scripts/micro/pingpong.hip: 552 against 1 304 cycles one way on an idle GPU) and to nobody else.  When the launch has found all members of the cluster on
  // one XCD (Xch::local, cl_find_local) the word goes out plain; both stores are always issued, one of them out of range -- no branch around a memory operation.
  const unsigned so = __builtin_amdgcn_readfirstlane(xch_wslot(x, x.m));
  __builtin_amdgcn_raw_buffer_store_b128(w, x.xb, x.local ? voff : PT_OOB, so, 0);
  __builtin_amdgcn_raw_buffer_store_b128(w, x.xb, x.local ? PT_OOB : voff, so, CL_AUX_SC1);
  STORE128_PAD(w);
}
// fetch NB words of the exchange just published (per-lane byte offsets vo, PT_OOB = idle lane -> 0;
// uniform slot offsets so); spins until every tag of the wave matches
template <int NB>
__device__ __forceinline__ bool xld(const Xch &x_in, const unsigned (&vo)[NB], const unsigned (&so)[NB], double (&out)[NB], unsigned tag = 0, ldp xprof = nullptr) {
  Xch x = x_in;
  if (tag) x.epoch = tag;                          // an exchange other than the latest one
  x.epoch = __builtin_amdgcn_readfirstlane(x.epoch); x.launch = __builtin_amdgcn_readfirstlane(x.launch);
#ifdef POTUS_PROF_FETCH
  const long long xt0_ = clock64();
#else
  (void)xprof;
#endif
  u32x4 w[NB];
  bool done[NB];                                   // wave-uniform
#pragma unroll
  for (int u = 0; u < NB; u++) w[u] = __builtin_amdgcn_raw_buffer_load_b128(x.xb, vo[u], __builtin_amdgcn_readfirstlane(so[u]), CL_AUX_SC1);
  bool all = true;
#pragma unroll
  for (int u = 0; u < NB; u++) {
    done[u] = __all(vo[u] == PT_OOB || (w[u][2] == x.epoch && w[u][3] == x.launch));
    all = all && done[u];
  }
#ifdef POTUS_PROF_FETCH
  if (xprof && (threadIdx.x & 63) == 0) { xprof[56] += (double)(clock64() - xt0_); xprof[57] += all ? 0.0 : 1.0; }
#endif
  // Words that were not there yet are fetched again (words already in hand are not re-read, so a spinning wave
  // does not flood the memory pipeline).  The re-fetch must stay an agent-scope (sc1) load: the `volatile` flavour of
  // the builtin becomes a system-scope load that costs ~2 us per round here.  What keeps the compiler from hoisting
  // it out of the loop is the laundered offset and the memory clobber.
  for (unsigned spins = 0; !all; spins++) {
    // a member is missing (or another wave of the cluster has already given up): leave instead of hanging the GPU
    if (spins > CL_SPIN_LIMIT || ((spins & 1023u) == 1023u && xch_watchdog_raised(x))) xch_give_up(x);
#ifdef POTUS_PROF_FETCH
    if (xprof && (threadIdx.x & 63) == 0) xprof[58] += 1.0;
#endif
    __builtin_amdgcn_s_sleep(CL_SPIN_SLEEP);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < NB; u++)
      if (!done[u]) {
        unsigned v = vo[u];
        asm volatile("" : "+v"(v));
        w[u] = __builtin_amdgcn_raw_buffer_load_b128(x.xb, v, __builtin_amdgcn_readfirstlane(so[u]), CL_AUX_SC1);
      }
    all = true;
#pragma unroll
    for (int u = 0; u < NB; u++) {
      if (!done[u]) done[u] = __all(vo[u] == PT_OOB || (w[u][2] == x.epoch && w[u][3] == x.launch));
      all = all && done[u];
    }
  }
#pragma unroll
  for (int u = 0; u < NB; u++) out[u] = __hiloint2double((int)w[u][1], (int)w[u][0]);   // idle lanes loaded zeros
  return true;
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// NOTE on position vectors (QC and the proposal-pool slots): their small-vector part is written with sc1
// (write-through) stores, which do not refresh a line this compute unit's own L1 may still hold; every read of a
// position outside the pass therefore uses sc1 loads as well (bld_s), never plain ones.
// Sum N <= 8 values over every thread of every member; all threads of all members return the same bits.
// Also a cluster-wide barrier: plain or write-through stores issued by any wave of any member before
// the call are visible to sc1 loads issued after it.
template <int N>
__device__ __forceinline__ void cl_allreduce(double (&v)[N], ldp red, Xch &x, int tid, ldp prof = nullptr) {
  (void)prof;
  static_assert(N <= 8, "payload words");
  const int lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const double t = dpp_scan_sum(v[k]);
    if (lane == 63) red[w * N + k] = t;
  }
  PROF_MARK(20);
  drain_vmem();   // this wave's stores are complete before the workgroup publishes
  PROF_MARK(21);
  __syncthreads();
  PROF_MARK(22);
  {
    // (global memory operations are kept out of branches: the compiler drains the whole memory queue, i.e.
    // waits for the write-through store to complete, wherever control flow joins after one)
    double s = 0.0;
    const int k = lane < N ? lane : N - 1;
#pragma unroll
    for (int i = 0; i < PT_NW; i++) s += red[i * N + k];
    xst(x, (w == 0 && lane < N) ? 16u * (unsigned)lane : PT_OOB, s);
  }
  x.epoch++;
  PROF_MARK(23);
  // wave 0: lane l fetches word (l & 7) of members (l >> 3) + 8u; fixed-shape tree over the members
  if (w == 0) {
    double s[CL_MAXK / 8];
    unsigned vo[CL_MAXK / 8], so[CL_MAXK / 8];
#pragma unroll
    for (int u = 0; u < CL_MAXK / 8; u++) {
      const int mm = (lane >> 3) + 8 * u;
      vo[u] = (mm < x.K && (lane & 7) < N) ? (unsigned)mm * (unsigned)x.XW * 16u + 16u * (unsigned)(lane & 7) : PT_OOB;
      so[u] = xch_rslot(x, 0);
    }
    xld(x, vo, so, s);
    PROF_MARK(24);
    double t = ((s[0] + s[1]) + s[2]) + s[3];
    t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    if (lane < N) red[PT_NW * N + lane] = t;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; k++) v[k] = red[PT_NW * N + k];
  PROF_MARK(26);
}
#define WP(v, w) ((w) * CL_WIDE + (v))   // per-wave partial sums of value v: a row per wave, so that lane v reads its eight partials from eight conflict-free rows
// All-reduce of nv <= CL_WIDE values whose per-wave partial sums sit in LDS (part[WP(v, wave)]), in two halves
// so that the wait can be taken later: cl_wide_publish sends this member's sums and returns the exchange number;
// cl_wide_consume (wave 0 only, no barrier) collects the totals into out[0 .. nv) (LDS), bit-identical everywhere.
__device__ __forceinline__ unsigned cl_wide_publish(ldp part, int nv, Xch &x, ldp prof = nullptr) {
  (void)prof;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  PROF_MARK(20);
  drain_vmem();                                    // every store of the leaf is complete before the member reports it
  PROF_MARK(21);
  __syncthreads();
  PROF_MARK(22);
  {
    const bool ok = w == 0 && lane < nv;
    double p8[PT_NW];
#pragma unroll
    for (int i = 0; i < PT_NW; i++) p8[i] = part[WP(ok ? lane : 0, i)];
    ISSUE_FENCE();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < PT_NW; i++) s += p8[i];
    xst(x, ok ? 16u * (unsigned)lane : PT_OOB, s);
  }
  if (nv > 64) {                                   // trees deeper than 10 doublings only
    const int l = 64 + lane;
    const bool ok = w == 0 && l < nv;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < PT_NW; i++) s += part[WP(ok ? l : 0, i)];
    xst(x, ok ? 16u * (unsigned)l : PT_OOB, s);
  }
  x.epoch++;
  PROF_MARK(23);
  return x.epoch;
}
__device__ __forceinline__ void cl_wide_consume(const Xch &x, unsigned tag, int nv, ldp out, ldp xprof = nullptr) {   // one wave
  const int lane = threadIdx.x & 63;
  for (int l0 = 0; l0 < nv; l0 += 64) {
    const int l = l0 + lane;
    double tot = 0.0;
    for (int mm0 = 0; mm0 < x.K; mm0 += 16) {
      double t16[16];
      unsigned vo[16], so[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { const int mm = mm0 + u; vo[u] = (mm < x.K && l < nv) ? 16u * (unsigned)l : PT_OOB; so[u] = xch_eslot(x, tag, mm < x.K ? mm : 0); }
      xld(x, vo, so, t16, tag, xprof);
#pragma unroll
      for (int u = 0; u < 16; u++) tot += t16[u];
    }
    if (l < nv) out[l] = tot;
  }
}

// What thread 0 needs to take the verdicts of a leaf once its totals are in (base_nuts::build_tree's bookkeeping):
// leaf number n of the doubling, m levels of subtrees it closes, whether it is the last leaf (top), its momentum
// slot, the slots of its position and of the position it produced.  n < 0: nothing pending.
struct LeafCtx { int n, m, top, depth, dir, leaf, inq, outq, nv; unsigned tag; int ext; };   // ext: word of the totals that says "the other
                                                                                              // cluster ended the trajectory" (twin mode), or -1

// totals: wout[0] = lp, [1] = kinetic, [2 + 6 (j-1) ..] the six dot products of level j, then the top check.
// Run by every lane of wave 0 with identical values (uniform control flow, lane 0 stores): what the verdicts read from
// LDS is fetched in one round -- the scalars by all lanes, the data of subtree level j by lane j-1 -- and then handed
// around with v_readlane, instead of some thirty dependent LDS round trips of a single thread.
__device__ __forceinline__ void cl_leaf_logic(ltp ts, const LeafCtx &L, ldp wout) {   // wave 0
  const int lane = threadIdx.x & 63;
  const int m = L.m, leaf = L.leaf, inq = L.inq;
  const int jl = lane < m ? lane : 0;              // this lane holds level jl + 1
  const double H0 = ts->H0, lpv = wout[0], kin = wout[1], sum_metro0 = ts->sum_metro, lsw_top = ts->lsw, u_top = ts->u_top;
  const int div0 = ts->divergent, n_leap0 = ts->n_leap, sample0 = ts->sample_qid;
  unsigned qm = ts->qmask, pm = ts->pmask;
  const int l_pb = ts->pend_beg[jl], l_pe = ts->pend_end[jl], l_pp = ts->pend_prop[jl];
  const double l_pl = ts->pend_lsw[jl], l_us = ts->u_sub[L.n & 1][jl + 1];
  int l_persist, t_persist;
  {
    ldp d = wout + 2 + 6 * jl;
    const double d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5];
    ldp e = wout + 2 + 6 * m;                        // the doubling's own check (meaningful for the last leaf)
    const double e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3], e4 = e[4], e5 = e[5];
    l_persist = d0 > 0 && d1 > 0 && d2 > 0 && d3 > 0 && d4 > 0 && d5 > 0;
    t_persist = e0 > 0 && e1 > 0 && e2 > 0 && e3 > 0 && e4 > 0 && e5 > 0;
  }
  double h = 0.5 * kin - lpv;
  if (isnan(h)) h = INFINITY;
  const int div = (h - H0 > 1000.0) ? 1 : div0;
  const double wgt = H0 - h;
  const double sum_metro = sum_metro0 + (wgt > 0 ? 1.0 : exp(wgt));
  int cur_beg = leaf, cur_prop = inq, abort = div;   // the leaf's own position is its subtree's first proposal
  if (L.ext >= 0 && wout[L.ext] > 0.0) abort = 1;    // twin mode: the trajectory ended at the other end, this subtree is dropped
  const int cur_end = leaf;
  double cur_lsw = wgt;
  for (int j = 1; j <= m && !abort; j++) {           // wave-uniform
    const int ib = __builtin_amdgcn_readlane(l_pb, j - 1), ie = __builtin_amdgcn_readlane(l_pe, j - 1), pprop = __builtin_amdgcn_readlane(l_pp, j - 1);
    const int persist = __builtin_amdgcn_readlane(l_persist, j - 1), cb = cur_beg;
    const double lsw_sub = d_lse(dpp_readlane_d(l_pl, j - 1), cur_lsw);
    bool take_final;
    if (cur_lsw > lsw_sub) take_final = true;
    else take_final = dpp_readlane_d(l_us, j - 1) < exp(cur_lsw - lsw_sub);
    if (take_final) pool_free(qm, pprop);
    else { pool_free(qm, cur_prop); cur_prop = pprop; }
    if (ie != ib) pool_free(pm, ie);
    if (cb != cur_end) pool_free(pm, cb);
    cur_beg = ib;
    cur_lsw = lsw_sub;
    abort = !persist;
  }
  int depth_new = -1, stop_new = 0, sample_new = sample0;
  double lsw_new = lsw_top;
  if (!abort && L.top) {
    depth_new = L.depth + 1;
    const double lsw_sub = cur_lsw;
    bool accept;
    if (lsw_sub > lsw_top) accept = true;
    else accept = u_top < exp(lsw_sub - lsw_top);
    if (accept) { pool_free(qm, sample0); sample_new = cur_prop; }
    else pool_free(qm, cur_prop);
    lsw_new = d_lse(lsw_top, lsw_sub);
    stop_new = !t_persist;
  }
  if (lane == 0) {
    ts->divergent = div; ts->sum_metro = sum_metro; ts->n_leap = n_leap0 + 1;
    ts->nextq[L.dir] = L.outq;                      // the next leaf of this end evaluates the position just written
    ts->q_lp[inq] = lpv; ts->q_h[inq] = h;
    if (!abort) {
      ts->pend_beg[m] = cur_beg; ts->pend_end[m] = cur_end; ts->pend_lsw[m] = cur_lsw; ts->pend_prop[m] = cur_prop;
      if (L.top) {
        ts->depth = depth_new; ts->sample_qid = sample_new; ts->lsw = lsw_new;
        if (stop_new) ts->stop = 1;
      }
    }
    ts->qmask = qm; ts->pmask = pm;
    ts->abort = abort;
  }
}
__device__ __forceinline__ void cl_sync(Xch &x, ldp red) {
  double v[1] = {0.0};
  cl_allreduce(v, red, x, (int)threadIdx.x);
}
// Once per launch: do all members of the cluster run on the same XCD?  (Blocks are dealt to the XCDs round-robin, so with a multiple of eight chains they do --
// but nothing promises it, and a plain store is invisible to another XCD.)  One all-reduce of the XCC ids (write-through, as everything is until the answer is in).
__device__ __forceinline__ void cl_find_local(Xch &x, ldp red) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  const double xi = threadIdx.x == 0 ? (double)(id & 15u) : 0.0;
  double v[2] = {xi, xi * xi};
  x.local = 0;
  cl_allreduce(v, red, x, (int)threadIdx.x);
  const int local = (double)x.K * v[1] == v[0] * v[0] ? 1 : 0;      // n sum x^2 == (sum x)^2  <=>  all ids equal (small integers: exact)
  if (threadIdx.x == 0) cl_local = local;
  __syncthreads();
  x.local = local;
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
  __device__ __forceinline__ double g_fin(unsigned vo, double v, double q, const GT &) { bst(rg, vo, sg, v); return q; }
  __device__ __forceinline__ double gs_fin(unsigned vo, double v, double q, const GT &) { bst(rg, vo, sg, v); return q; }
};
// Pre-kicked leapfrog (LeapPolicy of potus_nuts.hpp); positions that other members read next pass
// (the small vectors, raw_e_bias) are stored write-through.
struct ClLeapPolicy {
  rsrc_t r;
  unsigned sQc, sQn, sPH, sM, sL;
  double he, e;
  // Level-1 U-turn check fused into the epilogue: when this leaf closes a two-leaf subtree (odd leaf number)
  // the previous leaf's momentum is its whole left half (begin = end = rho), so the six dot products of
  // build_tree collapse to two, p_prev . (p_prev + p) and p . (p_prev + p), and cost one extra load; the
  // sum p_prev + p (rho of the pair) is stored for the checks one level up.
  unsigned sPrev, sOut1;
  bool fuse1;
  static constexpr int NEXTRA = 3;
  double extra[3];
  struct QT { double q; };
  struct GT { double p, m, pp; };
  __device__ __forceinline__ void q_load(unsigned vo, QT &t) { t.q = bld(r, vo, sQc); }
  __device__ __forceinline__ void qs_load(unsigned vo, QT &t) { t.q = bld_s(r, vo, sQc); }
  __device__ __forceinline__ double q_fin(QT &t) { return t.q; }
  __device__ __forceinline__ void g_load(unsigned vo, GT &t) {
    t.p = bld(r, vo, sPH); t.m = bld(r, vo, sM);
    t.pp = bld(r, fuse1 ? vo : PT_OOB, sPrev);
  }
  template <bool SHARED>
  __device__ __forceinline__ double fin(unsigned vo, double v, double q, const GT &t) {   // returns the next position
    const double pf = t.p + he * v;
    bst(r, vo, sL, pf);
    const double ph = pf + he * v;
    bst(r, vo, sPH, ph);
    const double qn = q + e * t.m * ph;
    if (SHARED) bst_s(r, vo, sQn, qn);
    else bst(r, vo, sQn, qn);
    const double rs = t.pp + pf;
    bst(r, fuse1 ? vo : PT_OOB, sOut1, rs);
    extra[0] += t.m * pf * pf;                    // masked-off elements loaded m = 0
    extra[1] += t.m * t.pp * rs;
    extra[2] += t.m * pf * rs;
    return qn;
  }
  __device__ __forceinline__ double g_fin(unsigned vo, double v, double q, const GT &t) { return fin<false>(vo, v, q, t); }
  __device__ __forceinline__ double gs_fin(unsigned vo, double v, double q, const GT &t) { return fin<true>(vo, v, q, t); }
};

// The same leapfrog with the member's share of three vectors RESIDENT IN LDS across the leaves of a doubling (fixed-layout builds; round 5).
// Every element of the member's share is finished by the same thread in every pass, and what the epilogue of leaf n stores is what leaf
// n + 1 and the U-turn sweeps of leaf n load back: through L2, as store -> load round trips on the same address.  Resident, indexed by the
// element's position in the member's share (vo / 8 - e0; masked lanes use the dump slot NECAP):
//   RM   the inverse metric (constant during a transition; filled when a doubling starts, cl_fill_resident)
//   RPF  the momentum of the latest leaf: written here, read back as the previous leaf's momentum of the pair check (pp) by the next leaf and
//        as p_end of every subtree that closes at this leaf by the sweeps (cl_vop_merge_chain) -- with another thread map, behind an LDS barrier
//   RRS  rho of the pair that closes at this leaf: the right-hand rho of the level-2 merge
// Global stores stay where somebody reads them LATER (leaf slot, next position, pre-kicked momentum, the pair's rho when it becomes a pending
// subtree's: m == 1) and become write-only: nothing on the critical path reads them back.  Same arithmetic in the same order: same bytes.
struct ClLeapPolicyRes {
  rsrc_t r;
  unsigned sQc, sQn, sPH, sL;
  double he, e;
  unsigned sOut1;
  bool fuse1, out1_global;      // out1_global: the pair's rho is a pending subtree's (this leaf closes level 1 only): it goes to memory as well
  unsigned e0b;                 // byte offset of the member's first element
  static constexpr int NEXTRA = 3;
  double extra[3];
  struct QT { double q; };
  struct GT { double p, m, pp; };
  __device__ __forceinline__ int slot(unsigned vo) const { return vo == PT_OOB ? (int)ClFixed::NECAP : (int)((vo - e0b) >> 3); }   // position in the member's share; masked lanes: the dump slot
  static __device__ __forceinline__ ldp RM() { return (ldp)lds_dyn + ClFixed::l_rm; }
  static __device__ __forceinline__ ldp RPF() { return (ldp)lds_dyn + ClFixed::l_rpf; }
  static __device__ __forceinline__ ldp RRS() { return (ldp)lds_dyn + ClFixed::l_rrs; }
  __device__ __forceinline__ void q_load(unsigned vo, QT &t) { t.q = bld(r, vo, sQc); }
  __device__ __forceinline__ void qs_load(unsigned vo, QT &t) { t.q = bld_s(r, vo, sQc); }
  __device__ __forceinline__ double q_fin(QT &t) { return t.q; }
  __device__ __forceinline__ void g_load(unsigned vo, GT &t) {
    t.p = bld(r, vo, sPH);
    const int idx = slot(vo);
    t.m = RM()[idx];                                 // the dump slot holds 0: masked-off elements add nothing
    const double ppv = RPF()[idx];
    t.pp = fuse1 ? ppv : 0.0;
  }
  template <bool SHARED>
  __device__ __forceinline__ double fin(unsigned vo, double v, double q, const GT &t) {   // returns the next position
    const int idx = slot(vo);
    const bool ok = vo != PT_OOB;
    const double pf = t.p + he * v;
    bst(r, vo, sL, pf);
    RPF()[idx] = ok ? pf : 0.0;
    const double ph = pf + he * v;
    bst(r, vo, sPH, ph);
    const double qn = q + e * t.m * ph;
    if (SHARED) bst_s(r, vo, sQn, qn);
    else bst(r, vo, sQn, qn);
    const double rs = t.pp + pf;
    RRS()[idx] = ok ? rs : 0.0;
    bst(r, (fuse1 && out1_global) ? vo : PT_OOB, sOut1, rs);
    extra[0] += t.m * pf * pf;
    extra[1] += t.m * t.pp * rs;
    extra[2] += t.m * pf * rs;
    return qn;
  }
  __device__ __forceinline__ double g_fin(unsigned vo, double v, double q, const GT &t) { return fin<false>(vo, v, q, t); }
  __device__ __forceinline__ double gs_fin(unsigned vo, double v, double q, const GT &t) { return fin<true>(vo, v, q, t); }
};

struct ClStatic {           // per-thread registers that never change during a kernel
  int s2info;               // lane j < CL_DW of wave w, for the wave's j-th day: (last polled local day <= it) + 1 (0 = none)
                            //   | chunk holding that day's last poll << 8
  int wd0, wnd;             // the wave's days: local days [wd0, wd0 + wnd), wnd <= CL_DW
  int ca, cb;               // the wave's chunk of the member's polls (day order) in the adjoint gather
  unsigned rep_vo[2];       // byte offsets of the small parameters this thread fetches for the workgroup
  unsigned rep_xo[2];       // the same parameters as exchange words of their owners (member * XW + XQ0 + index) * 16
  int sg_a, sg_b, sg_kind, sg_index;   // level-2 segment summed by this thread: task range, what the sum feeds
  double scale_r;           // scale of the small-vector slot this thread owns (threads 128 ..)
  int cellw[CL_CELLS_PER_THREAD];   // adjoint scatter: first poll (10 bits) | polls (6 bits) | offset of the cell in G (16 bits); idle: the dump slot
};

__device__ __forceinline__ ClStatic cl_load_static(CCp CL, cip part) {
  ClStatic c;
  gcip sc = as_g(CL->sched) + part[CP_O_WD];
  const int tid = threadIdx.x;
  c.s2info = sc[tid];
  c.wd0 = sc[PT_THREADS + (tid >> 6)]; c.wnd = sc[PT_THREADS + PT_NW + (tid >> 6)];
  c.ca = sc[PT_THREADS + 2 * PT_NW + (tid >> 6)]; c.cb = sc[PT_THREADS + 3 * PT_NW + (tid >> 6)];
  gcip rp = as_g(CL->rep_pos);
  const int NREP = CL->NREP;
  gcip ro = as_g(CL->rep_owner);
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int j = tid + u * PT_THREADS;
    c.rep_vo[u] = j < NREP ? 8u * (unsigned)rp[j] : PT_OOB;
    c.rep_xo[u] = j < NREP ? ((unsigned)ro[j] * (unsigned)CL->XW + (unsigned)(XQ0 + j)) * 16u : PT_OOB;
  }
  {
    gcip sch = as_g(CL->sched);
    const int nseg = part[CP_NSEG];
    // with the adjoint on the matrix cores the level-2 sums share a barrier interval with it: the segments go to the upper
    // waves first (waves 0-3 run the MFMA chains)
    // (walk build: wave 0 turns the chunk totals into prefixes in that interval, so it gets segments last)
    const int sg = CL->l_G ? ((tid + PT_THREADS / 2) & (PT_THREADS - 1)) : ((tid + PT_THREADS - CL_SEG_SHIFT) & (PT_THREADS - 1));
    const bool ok = sg < nseg;
    c.sg_a = ok ? sch[part[CP_O_SEGPTR] + sg] : 0;
    c.sg_b = ok ? sch[part[CP_O_SEGPTR] + sg + 1] : 0;
    c.sg_kind = ok ? sch[part[CP_O_SEGKIND] + sg] : 3;
    c.sg_index = ok ? sch[part[CP_O_SEGIDX] + sg] : 0;
#pragma unroll
    for (int h = 0; h < CL_CELLS_PER_THREAD; h++) c.cellw[h] = CL->l_G ? sch[part[CP_O_CELL] + h * PT_THREADS + tid] : 0;
    const int jr = tid - 128;
    c.scale_r = (jr >= 0 && jr < part[CP_NR]) ? as_g(CL->rep_scale)[part[CP_R0] + jr] : 0.0;
  }
  return c;
}
// Stage the walk factor and the (pseudo-)states of the member's polls in LDS, once per kernel.
__device__ __forceinline__ ClStatic cl_setup_lds(CMp M, CCp CL, cip part, ldp lds) {
  if (threadIdx.x == 0) { cl_dead = 0; cl_local = 0; }
#ifdef CL_POISON_LDS
  // development: whatever a kernel reads from LDS before writing it shows up as NaN instead of depending on the previous kernel
  for (int i = threadIdx.x; i < CL->lds_doubles; i += PT_THREADS) lds[i] = __builtin_nan("");
  __syncthreads();
#endif
  ldp Lw = lds + CL->l_Lw;
  gcdp src = as_g(M->mat);
  for (int i = threadIdx.x; i < (M->SE + 1) * M->SP; i += PT_THREADS) Lw[i] = i < M->SE * M->SP ? src[i] : 0.0;   // + a zero row
  // mu_b_prior and, in the pseudo-state's entry, its national average (stan:85,87).  The factors of mu_b_T and of the polling
  // bias (stan:77,85) are not staged: they are aT and aB times the walk's factor (DevModel::aT), see phase B.
  for (int i = threadIdx.x; i < M->SE; i += PT_THREADS) (lds + CL->l_prior)[i] = src[M->m_priorx + i];
  unsigned char AS_L *st = (unsigned char AS_L *)(lds + CL->l_st);
  gcip ps = as_g(M->pi) + part[CP_P0];
  const int np = part[CP_NP];
  for (int i = threadIdx.x; i < np + 8; i += PT_THREADS) st[i] = i < np ? (unsigned char)ps[i] : (unsigned char)0;
  {
    // per-poll constants: {state, local day, pollster, mode, population} packed in 64 bits; the two counts {y, N}
    // (int32 in the Stan data block) in another 64; the unadjusted flag
    unsigned long long AS_L *pm = (unsigned long long AS_L *)(lds + CL->l_pm);
    unsigned long long AS_L *pyn = (unsigned long long AS_L *)(lds + CL->l_py);
    ldp pun = lds + CL->l_pun;
    const int Npad = M->Npad, p0 = part[CP_P0], d0 = part[CP_D0], full = M->full;
    gcip pi = as_g(M->pi);
    gcdp pd = as_g(M->pd);
    for (int i = threadIdx.x; i < np + 8; i += PT_THREADS) {
      unsigned long long v = 0;
      double y = 0.0, N = 0.0, un = 0.0;
      if (i < np) {
        const int g = p0 + i;
        v = (unsigned long long)(unsigned)pi[g] | ((unsigned long long)(unsigned)(pi[Npad + g] - d0) << 8) |
            ((unsigned long long)(unsigned)pi[2 * Npad + g] << 16);
        if (full) v |= ((unsigned long long)(unsigned)pi[3 * Npad + g] << 32) | ((unsigned long long)(unsigned)pi[4 * Npad + g] << 40);
        {   // bits 48..: the wave that owns the poll's day (its suffix carry is row `wave` of the carry table, phase B)
          gcip scw = as_g(CL->sched) + part[CP_O_WD];
          const int tl_ = pi[Npad + g] - d0;
          int wt = 0;
          for (int w_ = 0; w_ < PT_NW; w_++) { const int a_ = scw[PT_THREADS + w_], n_ = scw[PT_THREADS + PT_NW + w_]; if (tl_ >= a_ && tl_ < a_ + n_) wt = w_; }
          v |= (unsigned long long)(unsigned)wt << 48;
        }
        y = pd[g]; N = pd[Npad + g]; un = full ? pd[2 * Npad + g] : 0.0;
      }
      pm[i] = v; pyn[i] = (unsigned long long)(unsigned)(int)y | ((unsigned long long)(unsigned)(int)N << 32); pun[i] = un;
    }
    // gather program: per poll (day order) state | local day << 8 | flags << 16
    unsigned AS_L *tab = (unsigned AS_L *)(lds + CL->l_tab);
    gcip src_tab = as_g(CL->sched) + part[CP_O_MASK];
    for (int i = threadIdx.x; i < np + 64; i += PT_THREADS) tab[i] = i < np ? (unsigned)src_tab[i] : 0u;
    // level-1 task lists (16 local poll indices each) as 16-bit entries
    unsigned short AS_L *sb = (unsigned short AS_L *)(lds + CL->l_sub);
    gcip src_sub = as_g(CL->sched) + part[CP_O_SUB];
    const int n16 = part[CP_NSUB] * PT_SUBLEN;
    for (int i = threadIdx.x; i < n16; i += PT_THREADS) sb[i] = (unsigned short)src_sub[i];
  }
  // G: zero once; the scatter of every pass rewrites the same cells (the poll structure is static), everything else stays zero
  if (CL->l_G) for (int i = threadIdx.x; i < CL->GROWS * CL->GS + CL_G_PAD; i += PT_THREADS) (lds + CL->l_G)[i] = 0.0;
  const ClStatic c = cl_load_static(CL, part);
#ifdef POTUS_PROF
  for (int i = threadIdx.x; i < PT_NPROF; i += PT_THREADS) (lds + CL->l_prof)[i] = 0.0;
#endif
  __syncthreads();
  return c;
}

// A scalar read from LDS (or updated under a branch on one) is the same in every lane, but the compiler must assume it
// is not: branches on it become exec-mask code and every value assigned under them moves to vector registers.  This
// tells it.
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---------------------------------------------------------------- one pass of the member's share
// Returns the chain's lp in every thread of every member; pol.extra[] are summed alongside.
// cl_pass_partial returns THIS THREAD's share of lp (and leaves the thread's share of pol.extra[] in
// pol_io); the caller reduces them over the cluster (cl_pass below, or together with the U-turn dot
// products of the leaf in cl_transition_tree).
// pubnext: the position this pass writes is the one the next pass evaluates (consecutive leaves of a subtree),
// so the suffix totals of that position (exchange X1 of the NEXT pass) are published here, as soon as the
// epilogue has produced it; the next pass then finds x.x1e set and does not wait for its X1 at all.
// pend: a leaf whose totals (exchange pend.tag) have been sent but not looked at: an otherwise idle wave collects them
// in phase C and takes the verdicts (cl_leaf_logic); if they end the trajectory the pass stops at the phase-C barrier
// (aborted = true).  This takes the wait for
// the leaf's all-reduce and the serial bookkeeping off the critical path of consecutive leaves.
template <int CL_TAG, class Pol>
__device__ __forceinline__ double cl_pass_partial(CMp M_in, CCp CL_in, cip part_in, ldp lds, const ClStatic &cst, Xch &x, Pol &pol_io,
                                                  bool pubnext, const LeafCtx &pend, ltp ts, ldp wout, bool &aborted) {
  // the builds of the pass (every template above and in potus_hmc.hip just hands the tag down): 4 / 8 = days per wave with the
  // adjoint as a walk over the polls, 12 = four days per wave with the adjoint product on the matrix cores (MF)
  constexpr int CL_DW = ClTag<CL_TAG>::DW;
  constexpr bool MF = ClTag<CL_TAG>::MF, FX = ClTag<CL_TAG>::FX;
#define LAY(f) (FX ? (int)ClFixed::f : CL->f)
  Pol pol = pol_io;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  CMp M = launder_s(M_in);
  CCp CL = launder_s(CL_in);
  cip part = launder_s(part_in);
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (FX) lds = (ldp)lds_dyn;   // the LDS base is a link-time constant
  const int S = FX ? (int)ClFixed::S : M->S, T = M->T, SE = FX ? (int)ClFixed::SE : M->SE, SP = FX ? (int)ClFixed::SP : M->SP, full = FX ? ClTag<CL_TAG>::FULL : M->full, o_c = M->o_c;
  const int NDP = FX ? (int)ClFixed::NDP : CL->NDP, NR = CL->NR, NREP = CL->NREP;
  const int d0 = part[CP_D0], nd = part[CP_ND], np = part[CP_NP], e0 = part[CP_E0], r0 = part[CP_R0], nr = part[CP_NR];
  const int K = x.K, m = x.m;
  const int wd0 = __builtin_amdgcn_readfirstlane(cst.wd0), wnd = __builtin_amdgcn_readfirstlane(cst.wnd);
  ldp C = lds + LAY(l_C), Lw = lds + LAY(l_Lw), X = lds + LAY(l_X), Y = lds + LAY(l_Y), r_lds = lds + LAY(l_r), ru_lds = lds + LAY(l_ru);
  ldp s_rep = lds + LAY(l_rep);
  ldp s_zT = s_rep, s_zb = s_rep + S, s_mid = s_rep + 2 * S;
  ldp s_bT = lds + LAY(l_bT), s_e = lds + LAY(l_e), s_c1 = lds + LAY(l_c1), s_c2 = lds + LAY(l_c2), s_c3 = lds + LAY(l_c3);
  ldp s_ge = lds + LAY(l_ge), s_P = lds + LAY(l_P), s_scal = lds + LAY(l_scal), red = lds + LAY(l_red);
  const int e_noise = e0 + S * nd, e_ze = part[CP_E_SH], e_rep = e_ze + (full ? nd : 0);   // the shared block starts on its own line
  double lp = 0.0;
#ifdef POTUS_PROF
  ldp prof = lds + LAY(l_prof);
#endif
  PROF_START();
  TSTAMP(0);

  // ---------------- phase A: the small vectors (all members read all of them), own days of the S x T block
  double cs[CL_DW], zq[CL_DW];
  {
    // the small vectors: from the exchange words their owners sent ahead with the previous leaf (x.x1e set: this
    // pass evaluates the position that leaf produced), else from the state vector (visible: older than one leaf)
    typename Pol::QT qr[2], qt[CL_DW];
    const bool ahead = x.x1e != 0;
    pol.qs_load(ahead ? PT_OOB : cst.rep_vo[0], qr[0]);
    pol.qs_load(ahead ? PT_OOB : cst.rep_vo[1], qr[1]);
    double qx[2];
    {
      const unsigned vo[2] = {ahead ? cst.rep_xo[0] : PT_OOB, ahead ? cst.rep_xo[1] : PT_OOB};
      const unsigned so[2] = {xch_eslot(x, x.x1e, 0), xch_eslot(x, x.x1e, 0)};
      xld(x, vo, so, qx, ahead ? x.x1e : 1u);
    }
#pragma unroll
    for (int j = 0; j < CL_DW; j++) {
      const int tl = wd0 + j;
      pol.q_load((lane < S && j < wnd) ? 8u * (unsigned)(e0 + lane + S * tl) : PT_OOB, qt[j]);
    }
    for (int i = tid; i < NR + 8; i += PT_THREADS) s_P[i] = 0.0;   // accumulators that this member's polls may not cover
    if (tid >= 64 && tid < 64 + CL_MAXDAYS) s_ge[tid - 64] = 0.0;
#pragma unroll
    for (int u = 0; u < 2; u++) { const int j = tid + u * PT_THREADS; s_rep[j < NREP ? j : NREP] = ahead ? qx[u] : pol.q_fin(qr[u]); }
    double run = 0.0;
#pragma unroll
    for (int j = CL_DW - 1; j >= 0; j--) {
      const int t = d0 + wd0 + j;
      const double z = pol.q_fin(qt[j]);          // 0 for masked-off elements
      zq[j] = z;
      lp -= 0.5 * z * z;                          // stan:119
      run += (t < T - 1) ? z : 0.0;               // column T is not part of the walk (stan:86)
      cs[j] = run;
    }
    if (lane < S) {
      Y[w * SE + lane] = run;
      // C[k][t] for now holds the suffix WITHIN the wave; what the later waves, the later members and u add is the same for every day of a wave: the poll phase adds
      // that one number per state from the carry table (round 6: the pass over the day block that did it, and its barrier, are gone -- same sums, same bytes)
#pragma unroll
      for (int j = 0; j < CL_DW; j++) { if (j < wnd) C[lane * NDP + wd0 + j] = cs[j]; }
    }
  }
  __syncthreads();
  PROF_MARK(0);
  TSTAMP(1);
  WPROF_T0();

  // ---------------- phase B: X1 (suffix totals); meanwhile mu_b_T / polling-bias mat-vecs and the AR(1) bias
  {
    double tot = 0.0;
#pragma unroll
    for (int w2 = 0; w2 < PT_NW; w2++) tot += Y[w2 * SE + (lane < S ? lane : S)];
    xst(x, (w == 0 && lane < S && x.x1e == 0) ? 16u * (unsigned)lane : PT_OOB, tot);
  }
  x.epoch += x.x1e == 0 ? 1u : 0u;                // X1 is on its way (or was sent by the previous pass)
  const unsigned x1tag = x.x1e ? x.x1e : x.epoch;
  x.x1e = 0;
  if (w == 1) {
    if (full) {
      // e_bias (stan:91-93): d[t] = e[t]-mu_e = rho d[t-1] + sigma_rho z[t].  Each lane owns four consecutive
      // days of a round of 256; affine scan across lanes on the DPP path.  The three tangent recurrences that turn the
      // adjoint sums of mu_e_bias / rho_e_bias into per-day dot products,
      //   c1[t] = rho c1[t-1] + 1, c2[t] = rho c2[t-1] + d[t-1], c3[t] = rho c3[t-1] + z[t]   (c.[0] = 0),
      // are only needed in phase E2 and run in phase C on an idle wave.
      const double sigma_e = M->sigma_e;
      ldp ze = s_mid + (M->o_ze - o_c);
      const double xm = s_mid[M->o_mue - o_c], xr = s_mid[M->o_rho - o_c];
      const double mue = 0.02 * xm, rho = d_inv_logit(xr);
      const double srho = sqrt(1.0 - rho * rho) * sigma_e;
      constexpr int PER = 4;
      double d_in = 0.0;                                 // d[base - 1]
      for (int base = 0; base < T; base += 64 * PER) {   // one round unless T > 256
        const int ta = base + lane * PER;
        double z[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) z[u] = ze[min(ta + u, T - 1)];
        ISSUE_FENCE();
        double A = 1.0, Bd[1] = {0.0};
#pragma unroll
        for (int u = 0; u < PER; u++) {
          const int t = ta + u;
          const bool in = t < T, first = t == 0;
          const double An = first ? 0.0 : rho * A, Bn = first ? z[u] * sigma_e - mue : rho * Bd[0] + srho * z[u];
          A = in ? An : A; Bd[0] = in ? Bn : Bd[0];
        }
        dpp_scan_affine(A, Bd);
        double d = dpp_prev_lane(A, 1.0) * d_in + dpp_prev_lane(Bd[0], 0.0);   // composite of the lanes before this one
#pragma unroll
        for (int u = 0; u < PER; u++) {
          const int t = ta + u;
          const double dn = t == 0 ? z[u] * sigma_e - mue : rho * d + srho * z[u];
          d = t < T ? dn : d;
          if (t < T) s_e[t] = d + mue;
        }
        d_in = dpp_readlane_d(d, 63);
      }
      if (lane == 0) { s_scal[SC_MUE] = mue; s_scal[SC_RHO] = rho; s_scal[SC_SRHO] = srho; s_scal[SC_XMUE] = xm; s_scal[SC_XRHO] = xr; }
    }
  }
  // The previous leaf's totals (exchange pend.tag, sent before this pass began) are fetched here by a wave that has nothing else
  // to do in this phase: an L2 round trip under load costs about 2 k cycles, which the verdict wave of phase C used to pay on
  // the critical path of that phase.  (Round 3 tried this while waves 2-7 still carried the 51 x 51 mat-vecs of mu_b_T and
  // the polling bias here and lost; those products are gone, see the carry table below.)
  // (Round 5 built the obvious merger -- the totals in the lanes the X1 fetch leaves idle, one round of sixteen loads for both, three leaves in four -- and
  //  measured it 9 % slower, 14.55 against 13.29 us per leapfrog: the totals arrive late, and wave 0 then holds the suffix carry back with them; docs/HISTORY.md.)
  if (w == 2 && pend.n >= 0) {
    cl_wide_consume(x, pend.tag, pend.nv, wout);
  }
  // mu_b[:, t] = prior + L_T z_T + L_W C[:, t] and polling_bias = L_B z_b (stan:77,85-86) enter a poll's predictor only as
  // prior[s] + L_W[s, :] . (aT z_T + aB z_b + C[:, t]): the three factors are one matrix times three scalars (stan:42-55).  So
  // u = aT z_T + aB z_b is folded into the suffix sums and no 51 x 51 product is left in the forward pass (rounds 1-3 computed
  // L_T z_T and L_B z_b here on six waves out of packed triangles in LDS, bound by LDS bandwidth).
  const double aT = M->aT, aB = M->aB;
  // suffix totals of the members that own later days: fetched once per workgroup (wave 0, which has
  // nothing else to do here) and handed to the other waves through LDS
  if (w == 0) {
    WPROF_PT(27);
    double carry_m = 0.0;
    for (int mm0 = m + 1; mm0 < K; mm0 += 16) {
      double t16[16];
      unsigned vo[16], so[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int mm = mm0 + u;
        vo[u] = (mm < K && lane < S) ? 16u * (unsigned)lane : PT_OOB;
        so[u] = xch_eslot(x, x1tag, mm < K ? mm : 0);
      }
#ifdef POTUS_PROF_FETCH
      xld(x, vo, so, t16, x1tag, prof);   // slots 56-58
#else
      xld(x, vo, so, t16, x1tag, nullptr);
#endif
#pragma unroll
      for (int u = 0; u < 16; u++) carry_m += t16[u];
    }
    if (lane < S) {
      // carry table X[w][k] (rows 0 .. 7 of X are free until phase D): what wave w's days add to their local suffix sums -- u = aT z_T + aB z_b, the later
      // members' totals and the later waves' -- summed exactly as the day-block pass used to sum it per wave
      double cy[PT_NW];
#pragma unroll
      for (int w2 = 0; w2 < PT_NW; w2++) cy[w2] = Y[w2 * SE + lane];
      const double zt = s_zT[lane], zb = s_zb[lane];
      ISSUE_FENCE();
#pragma unroll
      for (int w1 = 0; w1 < PT_NW; w1++) {
        double carry = carry_m + (aT * zt + aB * zb);
#pragma unroll
        for (int w2 = 0; w2 < PT_NW; w2++) carry += w2 > w1 ? cy[w2] : 0.0;
        X[w1 * SE + lane] = carry;
      }
    }
    WPROF_PT(28);
  }
  if (tid >= PT_THREADS - SE) s_bT[tid - (PT_THREADS - SE)] = (lds + LAY(l_prior))[tid - (PT_THREADS - SE)];
  if (tid == PT_THREADS - 64) { r_lds[np] = 0.0; ru_lds[np] = 0.0; }
  if (POTUS_PROF_WAVES == 1) WPROF_ACC(0);
  __syncthreads();
  PROF_MARK(1);
  TSTAMP(2);
  PROF_MARK(2);
  TSTAMP(3);

  // ---------------- phase C: the member's polls, one thread per poll (stan:95-112, 130-131).  The exp / log1p /
  // division of the binomial term are ~250 double-precision instructions per wave: polls are dealt to the
  // waves in blocks of 64 so that a member with n polls pays for ceil(n/64) wave passes, spread over the SIMDs
  {
    WPROF_CT0();
    const int om = M->o_m - o_c, opop = M->o_pop - o_c;
    const double sigma_c = M->sigma_c, sigma_m = M->sigma_m, sigma_pop = M->sigma_pop, sigma_ns = M->sigma_ns, sigma_nn = M->sigma_nn;
    const unsigned long long AS_L *pm = (const unsigned long long AS_L *)(lds + LAY(l_pm));
    const unsigned long long AS_L *pyn = (const unsigned long long AS_L *)(lds + LAY(l_py));
    ldp pun = lds + LAY(l_pun);
    // The scalars of rho's prior have a wave of their own in the fixed build only (profiles/r04_cl_inkernel_cycles.txt: one wave doing both set the
    // length of the phase): in the dynamic builds, which serve the no-mode posteriors among others, the mere presence of that branch cost 4 %
    // (bench.py --config 3: 698 k leapfrogs/s without, 667 k with)
    constexpr bool SPLIT_SCALARS = FX && ClTag<CL_TAG>::FULL;
    if (full && w == PT_NW - 1) {
      // the three tangent recurrences of the AR(1) bias (needed in phase E2 only) run here, on the wave that
      // has no polls unless the member has more than 448 of them, instead of lengthening phase B
      const double rho = s_scal[SC_RHO], mue = s_scal[SC_MUE], sigma_e = M->sigma_e;
      ldp ze = s_mid + (M->o_ze - o_c);
      constexpr int PER = 4;
      double c1_in = 0.0, c2_in = 0.0, c3_in = 0.0;
      for (int base = 0; base < T; base += 64 * PER) {   // one round unless T > 256
        const int ta = base + lane * PER;
        double z[PER], dp[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) { const int t = ta + u; z[u] = ze[min(t, T - 1)]; dp[u] = s_e[min(max(t - 1, 0), T - 1)] - mue; }
        ISSUE_FENCE();
        double A2 = 1.0, Bc[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < PER; u++) {
          const int t = ta + u;
          const bool in = t < T, first = t == 0;
          const double An = first ? 0.0 : rho * A2;
          const double B1 = first ? 0.0 : rho * Bc[0] + 1.0, B2 = first ? 0.0 : rho * Bc[1] + dp[u], B3 = first ? 0.0 : rho * Bc[2] + z[u];
          A2 = in ? An : A2; Bc[0] = in ? B1 : Bc[0]; Bc[1] = in ? B2 : Bc[1]; Bc[2] = in ? B3 : Bc[2];
        }
        dpp_scan_affine(A2, Bc);
        const double Ap = dpp_prev_lane(A2, 1.0);
        double c1 = Ap * c1_in + dpp_prev_lane(Bc[0], 0.0), c2 = Ap * c2_in + dpp_prev_lane(Bc[1], 0.0), c3 = Ap * c3_in + dpp_prev_lane(Bc[2], 0.0);
#pragma unroll
        for (int u = 0; u < PER; u++) {
          const int t = ta + u;
          const bool first = t == 0, in = t < T;
          const double n1 = first ? 0.0 : rho * c1 + 1.0, n2 = first ? 0.0 : rho * c2 + dp[u], n3 = first ? 0.0 : rho * c3 + z[u];
          c1 = in ? n1 : c1; c2 = in ? n2 : c2; c3 = in ? n3 : c3;
          if (in && t >= d0 && t < d0 + nd) { s_c1[t - d0] = c1; s_c2[t - d0] = c2; s_c3[t - d0] = c3; }   // only the member's own days are read (phase E2)
        }
        c1_in = dpp_readlane_d(c1, 63); c2_in = dpp_readlane_d(c2, 63); c3_in = dpp_readlane_d(c3, 63);
      }
      if (!SPLIT_SCALARS && lane == 0) {
        // what the owner of rho_e_bias needs in phase F: Jacobian + prior of rho (stan:63,124), d sigma_rho / d rho
        s_scal[SC_LPRHO] = log(rho) + log1p(-rho) - 0.5 * ((rho - 0.7) / 0.1) * ((rho - 0.7) / 0.1);
        s_scal[SC_DSRHO] = sigma_e * (-rho / sqrt(1.0 - rho * rho));
      }
    }
    if (SPLIT_SCALARS && full && w == PT_NW - 3 && lane == 0) {
      // ... on a wave of their own (idle unless the member has more than 320 polls): two logarithms, a square root and a division on
      // one lane are 3.3 k cycles, as long as the three recurrences together (profiles/r04_cl_inkernel_cycles.txt)
      const double rho = s_scal[SC_RHO];
      s_scal[SC_LPRHO] = log(rho) + log1p(-rho) - 0.5 * ((rho - 0.7) / 0.1) * ((rho - 0.7) / 0.1);
      s_scal[SC_DSRHO] = M->sigma_e * (-rho / sqrt(1.0 - rho * rho));
    }
    if (w == PT_NW - 2 && pend.n >= 0) {
      // The previous leaf's verdicts, on a wave that has no polls unless the member has more than 384 of them (its totals
      // were collected in phase B): the U-turn / accept logic runs beside the poll arithmetic of the other waves.
      WPROF_T0B();
      WPROF_PTB(29);
      cl_leaf_logic(ts, pend, wout);
      WPROF_PTB(30);
    }
    const bool no_polls_here = w == PT_NW - 2 && pend.n >= 0 && np <= 64 * w;   // the verdict wave, when it has no polls: skip the (idle) trip
    for (int i0 = 0; i0 < (no_polls_here ? 0 : np); i0 += PT_THREADS) {        // one trip unless a member has more than 512 polls
      const int il = i0 + tid;
      const bool ok = il < np, lead = ok;
      const int ic = ok ? il : np;                       // slot np holds zeros: N = y = 0
      const unsigned vq = lead ? 8u * (unsigned)(e_noise + il) : PT_OOB;
      typename Pol::QT qt;
      typename Pol::GT gt;
      pol.q_load(vq, qt);
      pol.g_load(vq, gt);
      double gval = 0.0, zn = 0.0;
      if (i0 + 64 * w < np) {                           // waves without polls skip the arithmetic (wave-uniform); no global
        const unsigned long long meta = pm[ic];          // memory operation inside the branch
        const int s = (int)(meta & 0xffu), tl = (int)((meta >> 8) & 0xffu), ip = (int)((meta >> 16) & 0xffffu);
        const int im = (int)((meta >> 32) & 0xffu), ipop = (int)((meta >> 40) & 0xffu);
        const unsigned long long yn = pyn[ic];
        const double y = (double)(int)(unsigned)yn, N = (double)(int)(unsigned)(yn >> 32), un = pun[ic];
        const int t = d0 + tl;
        ldp L0 = Lw + s * SP, C0 = C + tl, CW = X + (int)((meta >> 48) & 0xffu) * SE;   // CW: the carry of the wave that owns day tl
        double a0 = 0.0, a1 = 0.0;
          int k0 = 0;
          // 51-term dot in batches of TEN terms, the three operands of a batch -- factor row, suffix sum, the wave's carry -- in flight together: one LDS round per batch.
          // (Sixteen terms with the carry in a round of its own: +1.1 % per leapfrog; twelve or sixteen with all three in flight spill in this loop: +13 ... 20 %;
          //  the single term that 51 = 5 x 10 + 1 leaves over, requested with the first batch: 23 spills, +8 % -- profiles/r06_cl_carry_table.txt.)
          constexpr int DW_ = 10;
          for (; k0 + DW_ <= S; k0 += DW_) {
            double l[DW_], c[DW_], cw[DW_];
#pragma unroll
            for (int j = 0; j < DW_; j++) { l[j] = L0[k0 + j]; c[j] = C0[(k0 + j) * NDP]; cw[j] = CW[k0 + j]; }
            ISSUE_FENCE();
#pragma unroll
            for (int j = 0; j < DW_; j++) c[j] = c[j] + cw[j];
#pragma unroll
            for (int j = 0; j < DW_; j += 2) { a0 += l[j] * c[j]; a1 += l[j + 1] * c[j + 1]; }
          }
          for (; k0 < S; k0 += 4) {                      // remainder in fours: clamped reads, masked products
            double l[4], c[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { const int kk = min(k0 + j, S - 1); l[j] = L0[kk]; c[j] = C0[kk * NDP] + CW[kk]; }
            ISSUE_FENCE();
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
              a0 += (k0 + j < S ? l[j] : 0.0) * c[j];
              a1 += (k0 + j + 1 < S ? l[j + 1] : 0.0) * c[j + 1];
            }
          }
        const double dot = a0 + a1;
        const double sg = s == S ? sigma_nn : sigma_ns;
        WPROF_CSTAMP(0);
        zn = pol.q_fin(qt);
#if (POTUS_PROF_CMASK & 2)
        __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the element and its momentum have arrived
#endif
        WPROF_CSTAMP(1);
        double eta = s_bT[s] + sg * zn + sigma_c * s_mid[ip] + dot;   // s_bT: mu_b_prior (national: its weighted average)
        if (full) eta += sigma_m * s_mid[om + im] + sigma_pop * s_mid[opop + ipop] + un * s_e[t];
        // binomial_logit with one exp, one log1p, one division:  e = exp(-|eta|), l = log1p(e)
        const double ex = exp(-fabs(eta)), l1 = log1p(ex), pr = (eta >= 0.0 ? 1.0 : ex) / (1.0 + ex);
        const double r = y - N * pr;
        WPROF_CSTAMP(2);
        const double term = y * (fmin(eta, 0.0) - l1) + (N - y) * (fmin(-eta, 0.0) - l1) - 0.5 * zn * zn;   // stan:126-127,130-131 (zn = 0 on idle lanes)
        lp += term;
        r_lds[lead ? il : np + 1] = r;                   // slot np stays 0 (padding of the task lists), np+1 is a dump
        ru_lds[lead ? il : np + 1] = r * un;             // feeds the day sums of the AR(1) adjoint
        gval = sg * r - zn;
      }
      pol.g_fin(vq, gval, zn, gt);
      WPROF_CSTAMP(3);
    }
    WPROF_CFLUSH();
  }
  WAVE_ARRIVE(3);
  __syncthreads();
  PROF_MARK(3);
  TSTAMP(4);
  // The verdicts ended the trajectory: every member leaves here together.  What this pass has stored so far (the
  // epilogue of the poll-noise elements) went to slots nobody reads once the trajectory is over.
  if (pend.n >= 0 && uni_i(ts->abort)) { aborted = true; return 0.0; }

  // ---------------- phase D: adjoint of the walk, gC[:,t] = sum_i r_i Lw_ext[s_i,:] summed over days <= t.
  // The member's polls (day order) are cut into PT_NW equal chunks, one per wave, whatever the days: a wave
  // fetches (program word, residual, unadjusted flag) of 64 polls with one LDS round, then walks them through
  // readlane; lanes < S keep the running sum over the chunk, lane 63 the day's sum of unadjusted * residual
  // (the adjoint input of e_bias[t]).  At the last poll of a day the running values go to LDS; the owners
  // of the days pick them up in phase E.  Level-1 segment sums follow.
  if constexpr (!MF) {
    const unsigned AS_L *tab = (const unsigned AS_L *)(lds + LAY(l_tab));
    const int lk = lane < S ? lane : 0;
    const int ca = __builtin_amdgcn_readfirstlane(cst.ca), cb = __builtin_amdgcn_readfirstlane(cst.cb);
    double acc = 0.0;
    for (int base = ca; base < cb; base += 64) {
      const int mine = base + lane < cb ? base + lane : np;   // slot np: program word 0, residual 0
      const unsigned ev = tab[mine];
      const double rv = r_lds[mine];
      const int cnt = min(64, cb - base);
      for (int p0 = 0; p0 < cnt; p0 += 8) {
        unsigned e8[8];
        double r8[8], l8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int pu = p0 + u;                    // < 64; lanes beyond the chunk hold the zero slot
          e8[u] = (unsigned)__builtin_amdgcn_readlane((int)ev, pu);
          r8[u] = readlane_d(rv, pu);
          l8[u] = Lw[(int)(e8[u] & 0xffu) * SP + lk];   // (the polls of day T too: their running sum feeds mu_b_T and the polling bias only, phase E2)
        }
        ISSUE_FENCE();
#pragma unroll
        for (int u = 0; u < 8; u++) {
          acc += r8[u] * l8[u];
          if (e8[u] & 0x10000u) {                   // last poll of its day (wave-uniform): running sum to LDS
            if (lane < S) C[lane * NDP + (int)((e8[u] >> 8) & 0xffu)] = acc;
          }
        }
      }
    }
    if (lane < S) X[w * SE + lane] = acc;          // chunk total
  }
  if constexpr (MF) {
    // Adjoint on the matrix cores, step 1: G[pseudo-state][local day] = sum of the residuals of the cell's polls (the polls of a
    // member are sorted by day, then state: a cell is a run).  One thread per cell (the cells of day T included: the prefix of that
    // day feeds mu_b_T and the polling bias only, phase E2).  Idle threads write the dump slot behind G.
    ldp G = lds + LAY(l_G);
#pragma unroll
    for (int h = 0; h < CL_CELLS_PER_THREAD; h++) {
      if (h == 0 || part[CP_NCELL] > h * PT_THREADS) {     // wave-uniform
        const int cw = cst.cellw[h], a = cw & 1023, cnt = (cw >> 10) & 63, off = (int)((unsigned)cw >> 16);
        double sum = r_lds[a];                           // a cell has at least one poll; idle threads read the zero slot np
        for (int i = 1; i < cnt; i++) sum += r_lds[a + i];
        G[off] = sum;
      }
    }
  }
  {
    const int nsub = part[CP_NSUB], wb = part[CP_WB];   // tasks from wb on sum unadjusted * residual (day sums)
    const u32x4 AS_L *sb = (const u32x4 AS_L *)(lds + LAY(l_sub));
    for (int sub0 = 0; sub0 < nsub; sub0 += PT_THREADS) {
      const int sub = sub0 + tid;
      const bool ok = sub < nsub;
      const u32x4 ia = sb[ok ? 2 * sub : 0], ib = sb[ok ? 2 * sub + 1 : 0];
      double rr[PT_SUBLEN];
      ldp rsrc_ = sub >= wb ? ru_lds : r_lds;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        rr[2 * j] = rsrc_[ok ? (int)(ia[j] & 0xffffu) : np]; rr[2 * j + 1] = rsrc_[ok ? (int)(ia[j] >> 16) : np];
        rr[8 + 2 * j] = rsrc_[ok ? (int)(ib[j] & 0xffffu) : np]; rr[9 + 2 * j] = rsrc_[ok ? (int)(ib[j] >> 16) : np];
      }
      ISSUE_FENCE();
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < PT_SUBLEN; j++) sum += rr[j];
      if (ok) Y[sub] = sum;
    }
  }
  WAVE_ARRIVE(4);
  __syncthreads();
  PROF_MARK(4);
  TSTAMP(5);

  // ---------------- phase E: the owners of the days pick up the running sums; level-2 segment sums
  // (the chunk totals X[chunk][state] are turned into exclusive prefixes in place by wave 0, which keeps their sum for
  // its X2 word; after the barrier every wave looks its days' chunks up instead of carrying the eight prefixes around)
  int tlast[CL_DW], ch[CL_DW];
  double cv[CL_DW];
  double chunk_total = 0.0;
  if constexpr (!MF) {
#pragma unroll
    for (int j = 0; j < CL_DW; j++) {
      const int info = __builtin_amdgcn_readlane(cst.s2info, j);
      tlast[j] = (info & 0xff) - 1; ch[j] = info >> 8;
      cv[j] = C[(lane < S ? lane : 0) * NDP + max(tlast[j], 0)];
    }
    if (w == 0) {
      double ct[PT_NW];
      const int lx = lane < S ? lane : S;
#pragma unroll
      for (int c = 0; c < PT_NW; c++) ct[c] = X[c * SE + lx];
      ISSUE_FENCE();
#pragma unroll
      for (int c = 0; c < PT_NW; c++) ct[c] = lane < S ? ct[c] : 0.0;
#pragma unroll
      for (int c = 0; c < PT_NW; c++) { X[c * SE + lx] = chunk_total; chunk_total += ct[c]; }
    }
  } else {
    (void)tlast; (void)ch; (void)cv;
    // Adjoint on the matrix cores, step 2: gC[k][t] = sum_s Lw_ext[s][k] G[s][t] as 16 x 16 x 4 fp64 MFMA tiles
    // (v_mfma_f64_16x16x4_f64: lane l feeds A[l & 15][l >> 4] = Lw_ext[4 st + (l >> 4)][16 kt + (l & 15)] and
    // B[l >> 4][l & 15] = G[4 st + (l >> 4)][16 tt + (l & 15)] and holds D[(l >> 4) + 4 v][l & 15]), wave kt < 4 the 16 columns
    // k of its tile, the tiles of 16 days one after the other.  What the owners of the days need is the PREFIX over days
    // (dZ[:, t] = sum_{u <= t} gC[:, u] - Z[:, t]): the days of a tile sit in the 16 lanes of a DPP row, so the prefix is four
    // row_shr steps on the accumulators, plus the running total of the earlier tiles.  Its cost does not depend on the number
    // of polls: the member that owns the poll-dense last days no longer sets the pace of the cluster.
    if (w < 4) {
      ldp G = lds + LAY(l_G);
      const int GS = CL->GS, nst = CL->GROWS >> 2, g = lane >> 4, n = lane & 15;
      const int nt = (nd + 15) >> 4;
      double carry[4] = {0.0, 0.0, 0.0, 0.0};
      constexpr int NST = 16;                       // up to 64 (pseudo-)states
      for (int tt = 0; tt < nt; tt++) {             // wave-uniform
        typedef double d4_t __attribute__((ext_vector_type(4)));
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int h0 = 0; h0 < NST; h0 += 8) {       // eight steps' operands in flight at a time (registers)
          double av[8], bv[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int st = h0 + u;
            av[u] = Lw[min(4 * st + g, SE) * SP + 16 * w + n];            // rows beyond the pseudo-state: the zero row of Lw_ext
            bv[u] = G[min(4 * st + g, 4 * nst - 1) * GS + 16 * tt + n];
          }
          ISSUE_FENCE();
#pragma unroll
          for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);   // (steps beyond nst multiply zero rows)
          ISSUE_FENCE();
        }
#pragma unroll
        for (int v = 0; v < 4; v++) {
          double x = acc[v];
          x += dpp_fetch<DPP_ROW_SHR(1), 0xf>(0.0, x);
          x += dpp_fetch<DPP_ROW_SHR(2), 0xf>(0.0, x);
          x += dpp_fetch<DPP_ROW_SHR(4), 0xf>(0.0, x);
          x += dpp_fetch<DPP_ROW_SHR(8), 0xf>(0.0, x);
          x += carry[v];
          const int k = 16 * w + g + 4 * v, t = 16 * tt + n;
          if (k < S && t < nd) C[k * NDP + t] = x;
          // the tile's last day (lane 15 of the row) to every lane of the row: ds_swizzle, lane <- (lane & 0x10) | 0xf
          const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
          const unsigned lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)u, 0x1f0), hi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(u >> 32), 0x1f0);
          carry[v] = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        }
      }
    }
  }
  {
    double sum = 0.0;
    const int sg_a = cst.sg_a, sg_b = cst.sg_b, sg_kind = cst.sg_kind, sg_index = cst.sg_index;
    for (int j0 = sg_a; j0 < sg_b; j0 += 8) {
      double yy[8];
#pragma unroll
      for (int u = 0; u < 8; u++) yy[u] = Y[min(j0 + u, sg_b - 1)];
      ISSUE_FENCE();
#pragma unroll
      for (int u = 0; u < 8; u++) sum += j0 + u < sg_b ? yy[u] : 0.0;
    }
    if (sg_kind == 0) s_P[sg_index] = sum;          // pollster / mode / population partial (slot index)
    else if (sg_kind == 2) s_ge[sg_index] = sum;    // sum of unadjusted * residual over a local day
  }
  WAVE_ARRIVE(5);
  __syncthreads();
  PROF_MARK(5);
  TSTAMP(6);

  // ---------------- phase E2: payload of X2
  double arA = 1.0, arB = 0.0;                      // wave 1 keeps its per-day adjoint composites for phase F
  double pay = 0.0;                                 // the word this lane publishes (wave 0: prefix total, wave 1: AR words)
  double pre[CL_DW];
  if constexpr (!MF) {
#pragma unroll
    for (int j = 0; j < CL_DW; j++) {
      const double cc = X[ch[j] * SE + (lane < S ? lane : S)];
      pre[j] = (tlast[j] >= 0 && lane < S) ? cv[j] + cc : 0.0;   // (lanes beyond the states: column S of X is never written; their value
                                                               //  would meet a zero metric element in the epilogue, and NaN * 0 is NaN)
    }
  } else {
#pragma unroll
    for (int j = 0; j < CL_DW; j++) pre[j] = C[(lane < S ? lane : 0) * NDP + min(wd0 + j, max(nd - 1, 0))];   // (days beyond the wave's: unused)
    if (w == 0) chunk_total = nd > 0 ? C[(lane < S ? lane : 0) * NDP + nd - 1] : 0.0;
  }
  if (w == 0) {
    pay = chunk_total;
  } else if (w == 1) {
    if (full) {
      // adjoint of the AR(1) recursion over the member's days (one lane per day): a[t] = ge[t] + rho a[t+1]
      // (one lane per day, lane l = local day nd-1-l, so that the recursion runs in lane order)
      const double rho = s_scal[SC_RHO];
      const bool in = lane < nd;
      const int tl = in ? nd - 1 - lane : 0;
      const double ge = in ? s_ge[tl] : 0.0;
      double Bv[1] = {ge};
      arA = in ? rho : 1.0;
      dpp_scan_affine(arA, Bv);
      arB = Bv[0];                                  // a[t] = arB + arA * a[first day of the next member]
      // (a member without days never writes its tangents: no product with whatever LDS holds there)
      const double t1 = in ? s_c1[tl] : 0.0, t2 = in ? s_c2[tl] : 0.0, t3 = in ? s_c3[tl] : 0.0;
      const double S1 = dpp_scan_sum(ge * t1), S2 = dpp_scan_sum(ge * t2), S3 = dpp_scan_sum(ge * t3);
      // lane 63 holds the member's composite and the three sums
      const double pv0 = dpp_readlane_d(arA, 63), pv1 = dpp_readlane_d(arB, 63);
      const double pv2 = dpp_readlane_d(S1, 63), pv3 = dpp_readlane_d(S2, 63), pv4 = dpp_readlane_d(S3, 63);
      pay = lane == 0 ? pv0 : lane == 1 ? pv1 : lane == 2 ? pv2 : lane == 3 ? pv3 : pv4;   // XP_S = XP_AR + 2
    }
  }
  // The gradients of raw_mu_b_T and raw_polling_bias need L_T' g and L_B' g with g[s] = the residuals of state s + w_s * the
  // national residuals (stan:77,85).  Both factors are multiples of the walk's, and L_W_ext' g = the adjoint of the walk summed
  // over ALL the polls = the sum over the members of the prefix totals published here (the polls of the last day included: they
  // take their real row of the factor in the gather, and the prefix of that day is not used, stan:86).  So the owners of those
  // slots add up the words XP_PRE + k instead of partial transposed mat-vecs, which rounds 1-3 computed here on six waves.
  xst(x, (w == 0 && lane < S) ? 16u * (unsigned)(XP_PRE + lane) : (w == 1 && full && lane < 5) ? 16u * (unsigned)(XP_AR + lane) : PT_OOB, pay);
  PROF_MARK(6);
  TSTAMP(7);
  {
    // slot partials of the pollster / mode / population effects: every thread publishes one slot (two with more than 512
    // slots); no branch around the stores
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int rr = tid + h * PT_THREADS;
      const bool ok = rr >= 2 * S && rr < NR;
      const double vp = s_P[ok ? rr : 2 * S];
      xst(x, ok ? 16u * (unsigned)(XP_P + rr) : PT_OOB, vp);
    }
  }
  // loads that do not depend on the exchange are issued while wave 0 waits
  const bool arl = full && w == 1 && lane < nd;                 // owner of raw_e_bias[d0 + lane]
  // small-vector slots: threads 128.. take the ordinary slots of the member; mu_e_bias and rho_e_bias (the last
  // two slots, whose gradients come from the three tangent sums) go to the last lanes of the last wave of their
  // owner so that they run beside, not after, the ordinary ones (thread 511 fetches the third sum for 510)
  const int NRo = full ? NR - 2 : NR;                           // ordinary slots
  const int nro = max(0, min(r0 + nr, NRo) - r0);               // ordinary slots of this member (<= 381, checked on the host)
  const int jr = tid - 128;
  const bool own_mue = full && r0 <= NR - 2 && NR - 2 < r0 + nr, own_rho = full && r0 <= NR - 1 && NR - 1 < r0 + nr;
  const bool is_mue = own_mue && tid == PT_THREADS - 3, is_rho = own_rho && tid == PT_THREADS - 2, is_s3 = own_rho && tid == PT_THREADS - 1;
  const bool repl = (jr >= 0 && jr < nro) || is_mue || is_rho;
  const int rslot = is_mue ? NR - 2 : is_rho ? NR - 1 : (repl ? r0 + jr : 0);
  const unsigned vo_x = arl ? 8u * (unsigned)(e_ze + nd - 1 - lane) : repl ? 8u * (unsigned)(e_rep + rslot - r0) : PT_OOB;
  typename Pol::GT gx{};
  pol.g_load(vo_x, gx);
  const double scale_r = cst.scale_r;
  x.epoch++;

  // ---------------- phase F: finish the gradients of everything this member owns
  // wave 0 fetches the prefix totals of the members that own earlier days (for the whole workgroup),
  // wave 1 finishes raw_e_bias, the owners of small-vector slots finish theirs; then the S x T block
  if (w == 0) {
    double carry_m = 0.0;
    for (int mm0 = 0; mm0 < m; mm0 += 16) {
      double t16[16];
      unsigned vo[16], so[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int mm = mm0 + u;
        vo[u] = (mm < m && lane < S) ? 16u * (unsigned)(XP_PRE + lane) : PT_OOB;
        so[u] = xch_rslot(x, mm < m ? mm : 0);
      }
#ifdef POTUS_PROF_FETCH
      xld(x, vo, so, t16, 0u, prof + 3);   // slots 59-61: cycles of the first fetch, leaves that had to wait, re-fetch rounds
#else
      xld(x, vo, so, t16, 0u, nullptr);
#endif
#pragma unroll
      for (int u = 0; u < 16; u++) carry_m += t16[u];
    }
    if (lane < S) X[PT_NW * SE + lane] = carry_m;
    PROF_MARK(7);
    TSTAMP(8);
  }
  double own_g = 0.0, own_q = 0.0;                 // gradient / position of the element this thread owns besides the S x T block
  if (w == 1) {
    if (full) {
      // carry of the adjoint from the members that own later days, then raw_e_bias of the member's days
      const unsigned vA = lane < K ? (unsigned)lane * (unsigned)x.XW * 16u + 16u * (unsigned)XP_AR : PT_OOB;
      double mAB[2];
      {
        const unsigned vo[2] = {vA, lane < K ? vA + 16u : PT_OOB}, so[2] = {xch_rslot(x, 0), xch_rslot(x, 0)};
        xld(x, vo, so, mAB);
      }
      const double mA = mAB[0], mB = mAB[1];
      double a_in = 0.0;
      for (int mm = K - 1; mm > m; mm--) a_in = readlane_d(mB, mm) + readlane_d(mA, mm) * a_in;
      const double a = arB + arA * a_in;
      const int t = d0 + (arl ? nd - 1 - lane : 0);
      const double z = s_mid[M->o_ze - o_c + (arl ? t : 0)];
      const double gv = a * (t >= 1 ? s_scal[SC_SRHO] : M->sigma_e) - z;
      lp -= arl ? 0.5 * z * z : 0.0;               // stan:125
      own_g = gv; own_q = z;
    }
  } else if (w >= 2 && __any(repl || is_s3)) {
    // owned slots of the small vectors: sum the K partials in member order
    double sum = 0.0;
    const unsigned v1 = is_mue ? 16u * (unsigned)XP_S : is_rho ? 16u * (unsigned)(XP_S + 1) : is_s3 ? 16u * (unsigned)(XP_S + 2)
                        : !repl ? PT_OOB : rslot < 2 * S ? 16u * (unsigned)(XP_PRE + (rslot < S ? rslot : rslot - S))   // L_W_ext' g, see phase E2
                        : 16u * (unsigned)(XP_P + rslot);
    for (int mm0 = 0; mm0 < K; mm0 += 16) {
      double t16[16];
      unsigned vo[16], so[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { const int mm = mm0 + u; vo[u] = mm < K ? v1 : PT_OOB; so[u] = xch_rslot(x, mm < K ? mm : 0); }
      xld(x, vo, so, t16);
#pragma unroll
      for (int u = 0; u < 16; u++) sum += t16[u];
    }
    const double s3 = readlane_d(sum, 63);           // third tangent sum (meaningful in the wave that owns rho_e_bias)
    const double qv = s_rep[rslot];
    double gv = scale_r * sum - qv;
    double dl = -0.5 * qv * qv;                    // stan:117,120-122,128
    if (is_mue || is_rho) {
      const double rho = s_scal[SC_RHO];
      const double g_mue = 0.02 * (1.0 - rho) * sum - qv;                       // sum = S1
      const double g_rho = ((sum + s3 * s_scal[SC_DSRHO]) - (rho - 0.7) / 0.01) * rho * (1.0 - rho) + (1.0 - 2.0 * rho);   // S2, S3
      gv = is_mue ? g_mue : g_rho;
      dl = is_mue ? -3.912023005428146 - 0.5 * qv * qv : s_scal[SC_LPRHO];     // log(0.02) + prior (stan:62,123) : stan:63,124
    }
    lp += repl ? dl : 0.0;
    own_g = gv; own_q = qv;
  }
  {
    const double qn_own = pol.gs_fin(vo_x, own_g, own_q, gx);   // outside the branches (vo_x is out of range for non-owners)
    // ... and sent ahead to the members that evaluate the next position (word XQ0 + index of the small parameter)
    const int jrep = arl ? NR + d0 + (nd - 1 - lane) : rslot;
    xst(x, (pubnext && (arl || repl)) ? 16u * (unsigned)(XQ0 + jrep) : PT_OOB, qn_own);
  }
  PROF_SUB(52);
  typename Pol::GT gz[CL_DW];
  unsigned voz[CL_DW];
#pragma unroll
  for (int j = 0; j < CL_DW; j++) {
    voz[j] = (lane < S && j < wnd) ? 8u * (unsigned)(e0 + lane + S * (wd0 + j)) : PT_OOB;
    pol.g_load(voz[j], gz[j]);
  }
  WAVE_ARRIVE(7);
  __syncthreads();
  PROF_SUB(53);
  {
    const double carry = lane < S ? X[PT_NW * SE + lane] : 0.0;   // pre[] is already the prefix within the member
    double nrun = 0.0;                              // suffix total of the next position over the wave's days
#pragma unroll
    for (int j = 0; j < CL_DW; j++) {
      const int t = d0 + wd0 + j;
      const double qn = pol.g_fin(voz[j], (t < T - 1 ? pre[j] + carry : 0.0) - zq[j], zq[j], gz[j]);
      nrun += (j < wnd && t < T - 1) ? qn : 0.0;
    }
    PROF_SUB(54);
    if (pubnext) {                                  // wave-uniform; LDS and a barrier only, the store stays outside
      if (lane < S) Y[w * SE + lane] = nrun;
      __syncthreads();
    }
    PROF_SUB(55);
  }
  {
    double tot = 0.0;
#pragma unroll
    for (int w2 = 0; w2 < PT_NW; w2++) tot += Y[w2 * SE + (lane < S ? lane : S)];
    xst(x, (pubnext && w == 0 && lane < S) ? 16u * (unsigned)lane : PT_OOB, tot);
    x.epoch += pubnext ? 1u : 0u;
    x.x1e = pubnext ? x.epoch : 0u;
  }
  PROF_MARK(18);
  TSTAMP(9);
#pragma unroll
  for (int k = 0; k < Pol::NEXTRA; k++) pol_io.extra[k] = pol.extra[k];
  (void)red;
  return lp;
}
#undef LAY
template <int CL_DW, class Pol>
__device__ __forceinline__ double cl_pass(CMp M, CCp CL, cip part, ldp lds, const ClStatic &cst, Xch &x, Pol &pol_io) {
  double v[1 + Pol::NEXTRA];
  const LeafCtx none{-1, 0, 0, 0, 0, 0, 0, 0, 0, 0u, -1};
  bool aborted = false;
  v[0] = cl_pass_partial<CL_DW>(M, CL, part, lds, cst, x, pol_io, false, none, (ltp)nullptr, (ldp)nullptr, aborted);
#pragma unroll
  for (int k = 0; k < Pol::NEXTRA; k++) v[1 + k] = pol_io.extra[k];
  cl_allreduce(v, lds + CL->l_red, x, (int)threadIdx.x);
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
  // (slot numbers read from LDS are wave-uniform, but only readfirstlane tells the compiler: see xst)
  __device__ __forceinline__ unsigned soff(int slot) const { return __builtin_amdgcn_readfirstlane((unsigned)slot * (unsigned)Dpad * 8u); }
  __device__ __forceinline__ ldp red() const { return lds + CL->l_red; }
};

#define CL_UNR 4
__device__ __forceinline__ int cl_first(const ClChain &c) { int t = c.e0 + c.tid; asm volatile("" : "+v"(t)); return t; }

template <bool SHARED_DST, bool SHARED_SRC = false>
__device__ __forceinline__ void cl_vop_copy(const ClChain &c, unsigned s_dst, unsigned s_src) {
  for (int base = cl_first(c); base < c.e1; base += CL_UNR * PT_THREADS) {
    double v[CL_UNR];
#pragma unroll
    for (int k = 0; k < CL_UNR; k++) {
      const int i = base + k * PT_THREADS;
      v[k] = SHARED_SRC ? bld_s(c.st, i < c.e1 ? 8u * i : PT_OOB, s_src) : bld(c.st, i < c.e1 ? 8u * i : PT_OOB, s_src);
    }
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
    if (si < 0) continue;                            // padding element: momentum stays 0
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
// One U-turn check (base_nuts::build_tree / transition): the six dot products stay per-wave partial sums in LDS (part[WP(v0 + k, wave)]);
// the leaf's single all-reduce adds them over the cluster.
__device__ __forceinline__ void cl_vop_merge_partial(ClChain &c, unsigned a_beg, unsigned a_end, unsigned a_rho, unsigned b_beg, unsigned b_end,
                                                     unsigned b_rho, unsigned out, ldp part, int v0) {
  const unsigned sM = c.soff(V_MINV);
  double v[6] = {0, 0, 0, 0, 0, 0};
  for (int base = cl_first(c); base < c.e1; base += CL_UNR * PT_THREADS) {
    double mi[CL_UNR], ab[CL_UNR], ae[CL_UNR], ar[CL_UNR], bb[CL_UNR], be[CL_UNR], br[CL_UNR];
#pragma unroll
    for (int k = 0; k < CL_UNR; k++) {
      const int i = base + k * PT_THREADS;
      const unsigned o = i < c.e1 ? 8u * i : PT_OOB;
      mi[k] = bld(c.st, o, sM); ab[k] = bld(c.st, o, a_beg); ae[k] = bld(c.st, o, a_end); ar[k] = bld(c.st, o, a_rho);
      bb[k] = bld(c.st, o, b_beg); be[k] = bld(c.st, o, b_end); br[k] = bld(c.st, o, b_rho);
    }
#pragma unroll
    for (int k = 0; k < CL_UNR; k++) {
      const int i = base + k * PT_THREADS;
      const double rs = ar[k] + br[k];
      bst(c.st, i < c.e1 ? 8u * i : PT_OOB, out, rs);
      const double sab = mi[k] * ab[k], sbe = mi[k] * be[k];
      v[0] += sab * rs;
      v[1] += sbe * rs;
      const double e1 = ar[k] + bb[k];
      v[2] += sab * e1;
      v[3] += mi[k] * bb[k] * e1;
      const double e2 = br[k] + ae[k];
      v[4] += mi[k] * ae[k] * e2;
      v[5] += sbe * e2;
    }
  }
  const int lane = c.tid & 63, w = c.tid >> 6;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double t = dpp_scan_sum(v[k]);
    if (lane == 63) part[WP(v0 + k, w)] = t;
  }
  // the sweep of the next level reads the vector `out` with a different thread-to-element map only through
  // the same element index i -> same thread: no barrier needed between consecutive sweeps
}
// The merges of levels 2 .. m that a leaf closes, in ONE sweep over the member's elements: level j joins the pending subtree of
// level j - 1 (first / last momentum and rho in the pool, ts->pend_*) with the subtree that ends at this leaf, whose rho is the
// result of level j - 1 -- kept in registers from level to level (the pair's rho of level 1 comes out of the epilogue, slot
// SCR0 + 1).  Level by level this used to be: load seven vectors, store rho, load it again for the next level -- a store
// acknowledgement plus a load round trip (~4 k cycles) per level.  Here the operands of up to CL_MG levels are requested together
// and only the last rho (the new pending subtree's, slot RHOLEV + m) is stored.  Same arithmetic per element and level; the six
// dot products of a level stay per-wave partial sums in LDS (part[WP(2 + 6 (j - 1) + k, wave)]) for the leaf's single all-reduce.
#define CL_MG 3                          // levels whose operands are in flight together
#define CL_MU 2                          // elements per thread and trip (4 x 2 levels, one trip for every member: 13.78 against 13.63 us, round 5)
// RES (fixed builds, ClLeapPolicyRes): the inverse metric, the leaf's momentum (p_end of every subtree that closes here) and the pair's rho
// (the right-hand rho of level 2) come out of the member's resident share in LDS -- the epilogue wrote them a moment ago, and loading them
// back from memory meant draining the stores and a store -> load round trip through L2 in front of every sweep.  What still comes from memory
// are the pending subtrees' vectors, written leaves ago.  (Requesting those ahead of the barrier in front of the sweep was built and measured in
// round 5: 48 more live registers, 41 vector spills, 16.0 against 13.5 us per leapfrog -- docs/HISTORY.md.)
__device__ __forceinline__ void cl_merge_slots(const ClChain &c, ltp ts, int m, int j0, unsigned (&s_ab)[CL_MG], unsigned (&s_ae)[CL_MG], unsigned (&s_ar)[CL_MG], unsigned (&s_bb)[CL_MG]) {
#pragma unroll
  for (int g = 0; g < CL_MG; g++) {
    const int j = min(j0 + g, m);
    s_ab[g] = c.soff(V_POOLP + uni_i(ts->pend_beg[j - 1])); s_ae[g] = c.soff(V_POOLP + uni_i(ts->pend_end[j - 1]));
    s_ar[g] = c.soff(V_RHOLEV + j - 1); s_bb[g] = c.soff(V_POOLP + uni_i(ts->pend_beg[j - 2]));
  }
}
template <bool RES>
__device__ __forceinline__ void cl_vop_merge_chain(ClChain &c, ltp ts, int m, int leaf, ldp part) {
  const unsigned sM = c.soff(V_MINV), s_be = c.soff(V_POOLP + leaf);
  const int lane = c.tid & 63, w = c.tid >> 6;
  unsigned s_br = c.soff(V_SCR0 + 1);
  for (int j0 = 2; j0 <= m; j0 += CL_MG) {
    const int ng = min(CL_MG, m - j0 + 1);
    unsigned s_ab[CL_MG], s_ae[CL_MG], s_ar[CL_MG], s_bb[CL_MG];
    cl_merge_slots(c, ts, m, j0, s_ab, s_ae, s_ar, s_bb);
    const int jl = j0 + ng - 1;                                            // last level of the group
    const unsigned s_out = jl == m ? c.soff(V_RHOLEV + m) : c.soff(V_SCR0 + (jl & 1));
    const bool first_group = j0 == 2;                                      // its right-hand rho is the pair's (out of the epilogue)
    double v[CL_MG][6];
#pragma unroll
    for (int g = 0; g < CL_MG; g++)
#pragma unroll
      for (int k = 0; k < 6; k++) v[g][k] = 0.0;
    for (int base = cl_first(c); base < c.e1; base += CL_MU * PT_THREADS) {
      unsigned o[CL_MU];
      double mi[CL_MU], be[CL_MU], br[CL_MU], ab[CL_MG][CL_MU], ae[CL_MG][CL_MU], ar[CL_MG][CL_MU], bb[CL_MG][CL_MU];
#pragma unroll
      for (int u = 0; u < CL_MU; u++) {
        const int i = base + u * PT_THREADS;
        o[u] = i < c.e1 ? 8u * i : PT_OOB;                                  // masked elements read zeros and add nothing
        if constexpr (RES) {
          const int ii = i < c.e1 ? i - c.e0 : (int)ClFixed::NECAP;        // (the dump slot holds zeros)
          mi[u] = ClLeapPolicyRes::RM()[ii]; be[u] = ClLeapPolicyRes::RPF()[ii];
          const double brl = ClLeapPolicyRes::RRS()[ii], brg = bld(c.st, first_group ? PT_OOB : o[u], s_br);
          br[u] = first_group ? brl : brg;
        } else {
          mi[u] = bld(c.st, o[u], sM); be[u] = bld(c.st, o[u], s_be); br[u] = bld(c.st, o[u], s_br);
        }
      }
#pragma unroll
      for (int g = 0; g < CL_MG; g++)
        if (g < ng) {                                                        // wave-uniform
#pragma unroll
          for (int u = 0; u < CL_MU; u++) { ab[g][u] = bld(c.st, o[u], s_ab[g]); ae[g][u] = bld(c.st, o[u], s_ae[g]); ar[g][u] = bld(c.st, o[u], s_ar[g]); bb[g][u] = bld(c.st, o[u], s_bb[g]); }
        }
#pragma unroll
      for (int g = 0; g < CL_MG; g++)
        if (g < ng) {
#pragma unroll
          for (int u = 0; u < CL_MU; u++) {
            const double rs = ar[g][u] + br[u];
            const double sab = mi[u] * ab[g][u], sbe = mi[u] * be[u];
            v[g][0] += sab * rs;                 // p#_beg . rho_subtree
            v[g][1] += sbe * rs;                 // p#_end . rho_subtree
            const double e1 = ar[g][u] + bb[g][u];  // rho_init + p_final_beg
            v[g][2] += sab * e1;
            v[g][3] += mi[u] * bb[g][u] * e1;
            const double e2 = br[u] + ae[g][u];  // rho_final + p_init_end
            v[g][4] += mi[u] * ae[g][u] * e2;
            v[g][5] += sbe * e2;
            br[u] = rs;                          // the merged subtree is the right-hand one of the next level
          }
        }
#pragma unroll
      for (int u = 0; u < CL_MU; u++) bst(c.st, o[u], s_out, br[u]);
    }
#pragma unroll
    for (int g = 0; g < CL_MG; g++)
      if (g < ng) {
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const double t = dpp_scan_sum(v[g][k]);
          if (lane == 63) part[WP(2 + 6 * (j0 + g - 1) + k, w)] = t;
        }
      }
    s_br = s_out;
  }
}
// The inverse metric of the member's share into its resident copy (start of a doubling: the metric changes at window ends only, between
// transitions).  The momentum and rho copies start from zeros: the epilogue never touches the padding elements of the member's share (their
// momentum is 0 in memory, and the sweeps multiply it with a metric element of 1), and LDS keeps whatever the previous kernel left there --
// found as transitions that differed from the oracle's once in some ten thousand leaves.  Ends with a barrier.
__device__ __forceinline__ void cl_fill_resident(const ClChain &c) {
  const unsigned sM = c.soff(V_MINV);
  for (int i = c.tid; i <= (int)ClFixed::NECAP; i += PT_THREADS) { ClLeapPolicyRes::RPF()[i] = 0.0; ClLeapPolicyRes::RRS()[i] = 0.0; }
  for (int base = cl_first(c); base < c.e1; base += CL_UNR * PT_THREADS) {
    double v[CL_UNR];
#pragma unroll
    for (int k = 0; k < CL_UNR; k++) { const int i = base + k * PT_THREADS; v[k] = bld(c.st, i < c.e1 ? 8u * i : PT_OOB, sM); }
#pragma unroll
    for (int k = 0; k < CL_UNR; k++) { const int i = base + k * PT_THREADS; ClLeapPolicyRes::RM()[i < c.e1 ? i - c.e0 : (int)ClFixed::NECAP] = v[k]; }   // (idle lanes: 0 into the dump slot)
  }
  if (c.tid == 0) ClLeapPolicyRes::RM()[ClFixed::NECAP] = 0.0;
  __syncthreads();
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
      q[k] = bld_s(c.st, o, sq); p[k] = bld(c.st, o, sp); g[k] = bld(c.st, o, sg); m[k] = bld(c.st, o, sM);
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
  cl_sync(c.x, c.red());
}

template <int CL_DW>
__device__ __forceinline__ void cl_transition_begin(ClChain &c, uint32_t iter) {
  ltp ts = c.ts;
  const int tid = c.tid;
  const double eps = c.sc->nom_eps; // sample_stepsize(): no jitter
  const double kin0 = cl_vop_momentum(c, c.soff(V_PC), iter, RNG_MOMENTUM, 0);
  ClPlainPolicy pp{c.st, c.st, c.soff(V_QC), c.soff(V_GC), {0}};
  const double lp0 = cl_pass<CL_DW>(c.M, c.CL, c.part, c.lds, c.cst, c.x, pp); // hamiltonian.init
  if (tid == 0) {
    ts->H0 = 0.5 * kin0 - lp0;
    ts->lsw = 0.0; ts->sum_metro = 0.0; ts->n_leap = 0; ts->depth = 0; ts->divergent = 0; ts->stop = 0; ts->eps = eps;
    // every position of the trajectory lives in a slot of the proposal pool: a leaf that becomes a proposal is kept
    // by index, nothing is copied
    unsigned qm = 0;
    const int id = pool_alloc(qm, PT_NPQ);
    ts->nextq[1] = pool_alloc(qm, PT_NPQ); ts->nextq[0] = pool_alloc(qm, PT_NPQ);
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
        q[k] = bld_s(c.st, o, c.soff(V_QC)); p[k] = bld(c.st, o, c.soff(V_PC));
      }
#pragma unroll
      for (int k = 0; k < CL_UNR; k++) {
        const int i = base + k * PT_THREADS;
        const unsigned o = i < c.e1 ? 8u * i : PT_OOB;
        bst(c.st, o, s_rt, p[k]); bst(c.st, o, s_qs, q[k]);
      }
    }
  }
  cl_vop_prekick(c, c.soff(V_QC), c.soff(V_PC), c.soff(V_GC), c.soff(V_PH1), c.soff(V_POOLQ + ts->nextq[1]), c.soff(V_PF1), 0.5 * eps, eps);
  cl_vop_prekick(c, c.soff(V_QC), c.soff(V_PC), c.soff(V_GC), c.soff(V_PH0), c.soff(V_POOLQ + ts->nextq[0]), c.soff(V_PF0), -0.5 * eps, -eps);
}

// ================================================================ two clusters per chain ("twin" mode)
// With 8 chains on clusters of 16 half of the 256 compute units idle.  A NUTS trajectory has two ends, and the leapfrog
// steps at one end do not depend on those at the other: a second cluster per chain integrates the other end at the same
// time.  Side 1 owns the forward end, side 0 the backward end; each has its own state block, exchange buffer and scalar
// replicas (block chain + side * chains) and runs the once-per-transition parts (momentum draw, first gradient,
// adaptation) redundantly -- same inputs, same bits.  The directions of all doublings are known at the start of a
// transition (counter-based RNG), so a side simply builds, one after the other, the subtrees of the doublings that go its
// way (build_tree with its internal U-turn checks and multinomial sampling is local to one end), without waiting for the
// doublings of the other side in between: that is speculation -- if the trajectory ends at an earlier doubling the work is
// dropped -- on compute units that would idle otherwise (bounded: a doubling beyond what the last four transitions needed is
// not started ahead of its turn).  What is NOT local is the bookkeeping of transition() itself: the
// accept step of the new subtree's proposal, the sum of momenta over the whole trajectory and the U-turn checks across
// it.  These "combines" are taken strictly in the order of the doublings by the side that built the subtree:
//   wait until the combine of the previous doubling is published (a side that finishes doubling d has always been busy
//   longer than the other side needed for its doublings < d: in practice nothing waits)  ->  accept step, rho_top +=
//   rho_subtree, the three checks with the OTHER end's momentum (read from the other side's state block)  ->  publish.
// Trajectory-level state travels as TT_N tagged 16-byte words in the chain's mailbox (RunParams::twbuf), two slots by
// combine parity; a combine that ends the trajectory also raises the STOP word, which the speculating side polls once
// per leaf (the flag goes through the leaf's all-reduce, so all members drop out at the same leaf).  At the end of a
// transition the side that holds the new sample publishes it (QC, write-through), the other side copies it; both publish
// DONE and wait for each other before the next transition.  Expected gain: the last doubling holds half of the leaves,
// the earlier ones fall on either side at random: max(forward, backward) = 3/4 of the leaves on average.
// Floating point: the metropolis terms are summed per doubling and then added to the trajectory's sum (the one-cluster
// sampler does the same, for that reason); everything else is the same arithmetic in the same order, so the two modes
// produce the same draws bit for bit (the U-turn dot products across the trajectory are reduced over the members in a
// different order, but only their signs are used).
enum { TT_STOP = 0, TT_DEPTH, TT_LSW, TT_SSIDE, TT_SSLOT, TT_SLP, TT_SH, TT_METRO, TT_NLEAP, TT_DIV, TT_RHOSIDE, TT_N = 12 };
enum { TWB_TOP = 0 /* [2][16] */, TWB_STOP = 32, TWB_QC = 33, TWB_DONE = 34 /* [2] */, TWB_WD = 36, TWB_WORDS = 40 };

struct Twin {
  rsrc_t tb;               // the chain's mailbox
  rsrc_t ost;              // the other side's state block
  unsigned launch, ittag;  // tags: dword 2 = (iteration + 1) << 6 | (combine number + 1, or 0), dword 3 = launch id
  int side;
  const double *tbase;     // the mailbox again, as a pointer (for the cold give-up path)
};
struct ClTwinArgs {        // what the cold functions of twin mode rebuild their context from
  const DevModel *Mg;
  const ClModel *CLg;
  const RunParams *Rg;
  int chain, m, side;
  unsigned launch;
};
__device__ __forceinline__ Twin make_twin(CRp R, int chain, int side, unsigned launch, uint32_t iter) {
  Twin t;
  // (everything made wave-uniform explicitly: a descriptor the compiler cannot prove uniform wraps every access in a waterfall loop)
  t.tbase = uni_ptr(R->twbuf + (size_t)chain * TWB_WORDS * 2);
  t.tb = make_rsrc(t.tbase, TWB_WORDS * 16u);
  t.ost = make_rsrc(uni_ptr(R->state + (size_t)(chain + (1 - side) * R->chains) * V_COUNT * R->Dpad), uni32((unsigned)V_COUNT * (unsigned)R->Dpad * 8u));
  t.launch = uni32(launch); t.ittag = uni32((iter + 1u) << 6); t.side = (int)uni32((unsigned)side);
  return t;
}
// one tagged word (lanes with voff == PT_OOB store nothing)
__device__ __forceinline__ void tw_st(const Twin &t, unsigned voff, double v, unsigned tag) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const u32x4 w = {(unsigned)u, (unsigned)(u >> 32), tag, t.launch};
  __builtin_amdgcn_raw_buffer_store_b128(w, t.tb, voff, 0u, CL_AUX_SC1);
  STORE128_PAD(w);
}
// lane l < n looks at word w0 + l: true when every one carries the tag
__device__ __forceinline__ bool tw_try(const Twin &t, int w0, int n, unsigned tag, double &val) {
  const int lane = threadIdx.x & 63;
  unsigned vo = lane < n ? 16u * (unsigned)(w0 + lane) : PT_OOB;
  asm volatile("" : "+v"(vo) :: "memory");           // a fresh load every call
  const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(t.tb, vo, 0u, CL_AUX_SC1);
  val = __hiloint2double((int)w[1], (int)w[0]);
  return __all(lane >= n || (w[2] == tag && w[3] == t.launch));
}
// the same for a word that later transitions overwrite with a later tag (DONE): any tag from `tag` on will do
__device__ __forceinline__ bool tw_try_ge(const Twin &t, int w0, unsigned tag) {
  unsigned vo = 16u * (unsigned)w0;
  asm volatile("" : "+v"(vo) :: "memory");
  const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(t.tb, vo, 0u, CL_AUX_SC1);
  return __all(w[2] >= tag && w[3] == t.launch);
}
__device__ __forceinline__ void tw_give_up(const Xch &x, const Twin &t) {
  // (the descriptor rebuilt from a pointer made uniform here: where the caller's copy has ended up in vector registers the
  // store would otherwise sit in a waterfall loop)
  const rsrc_t tb = make_rsrc(uni_ptr(t.tbase), TWB_WORDS * 16u);
  __builtin_amdgcn_raw_buffer_store_b32(1u, tb, 0u, 16u * TWB_WD, CL_AUX_SC1);
  xch_give_up(x);
}
__device__ __forceinline__ bool tw_dead(const Xch &x, const Twin &t) {
  return xch_watchdog_raised(x) || __builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(t.tb, 0u, 16u * TWB_WD, CL_AUX_SC1)) != 0u;
}
__device__ __forceinline__ void tw_wait(const Xch &x, const Twin &t, int w0, int n, unsigned tag, double &val) {   // one wave
  for (unsigned spins = 0; !tw_try(t, w0, n, tag, val); spins++) {
    if (spins > CL_SPIN_LIMIT || ((spins & 1023u) == 1023u && tw_dead(x, t))) tw_give_up(x, t);
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ void tw_wait_ge(const Xch &x, const Twin &t, int w0, unsigned tag) {   // one wave
  for (unsigned spins = 0; !tw_try_ge(t, w0, tag); spins++) {
    if (spins > CL_SPIN_LIMIT || ((spins & 1023u) == 1023u && tw_dead(x, t))) tw_give_up(x, t);
    __builtin_amdgcn_s_sleep(1);
  }
}
// Wave 0 brings the side's copy of the trajectory-level state to combine number `want` -- or to the final one if the
// trajectory has ended (want < 0: wait for the end).  No barrier inside.
__device__ __forceinline__ void tw_catch_up(const Xch &x, const Twin &t, ltp ts, int want, bool patient = false) {
  const int lane = threadIdx.x & 63;
  double v = 0.0;
  int seq = -1;
  bool have = false;
  for (unsigned spins = 0;; spins++) {
    if (tw_try(t, TWB_STOP, 1, t.ittag, v)) { seq = (int)readlane_d(v, 0); break; }   // the trajectory is over: number of its last combine
    if (want >= 0 && tw_try(t, TWB_TOP + 16 * (want & 1), TT_N, t.ittag | (unsigned)(want + 1), v)) { seq = want; have = true; break; }
    if (spins > CL_SPIN_LIMIT || ((spins & 1023u) == 1023u && tw_dead(x, t))) tw_give_up(x, t);
    if (want < 0 || patient) __builtin_amdgcn_s_sleep(32);   // waiting for the other side's doublings: a look every microsecond
    else __builtin_amdgcn_s_sleep(1);
  }
  if (!have) tw_wait(x, t, TWB_TOP + 16 * (seq & 1), TT_N, t.ittag | (unsigned)(seq + 1), v);
  if (lane < TT_N) ts->tt[lane] = v;
  if (lane == 0) ts->tw_seq = seq;
}

template <int CL_DW> __device__ __noinline__ unsigned cl_cold_twin_combine(const DevModel *Mg, const ClModel *CLg, const RunParams *Rg, int chain_, int m_, int side_,
                                                                           unsigned launch_, unsigned epoch_, uint32_t iter_, int depth_, int valid_, int leaf_);

template <int CL_DW, bool TWIN>
__device__ __forceinline__ void cl_transition_tree(ClChain &c, uint32_t iter, const ClTwinArgs &ta) {
  constexpr bool FX = ClTag<CL_DW>::FX;              // (the template argument is the build's tag)
  // The member's share resident in LDS (ClLeapPolicyRes) with two clusters per chain only: measured side by side (profiles/r05_cl_resident_share.txt)
  // 13.50 -> 13.15 / 13.22 us per leapfrog there (all 256 compute units busy: a round trip through L2 costs more), 16.53 -> 16.97 / 17.03 with one cluster.
  constexpr bool RES = FX && TWIN;
  ltp ts = FX ? (ltp)((ldp)lds_dyn + ClFixed::lds_doubles) : c.ts;
  const int tid = c.tid;
  const double eps = ts->eps;
  int depth = 0;
  if (TWIN) {
    // the directions of every doubling of this transition, and the trajectory-level state before the first one
    if (tid < 64) {
      const unsigned long long fw = __ballot(tid < c.max_depth && rng_uniform(c.key, iter, RNG_DIRECTION, 0, (uint32_t)(tid < c.max_depth ? tid : 0)) > 0.5);
      if (tid == 0) {
        const int id = ts->sample_qid;
        ts->tw_dirs = (int)(unsigned)fw; ts->tw_seq = 0; ts->tw_over = 0; ts->tw_keep = id;
        ts->tt[TT_STOP] = 0.0; ts->tt[TT_DEPTH] = 0.0; ts->tt[TT_LSW] = 0.0; ts->tt[TT_SSIDE] = -1.0; ts->tt[TT_SSLOT] = (double)id;
        ts->tt[TT_SLP] = ts->q_lp[id]; ts->tt[TT_SH] = ts->q_h[id]; ts->tt[TT_METRO] = 0.0; ts->tt[TT_NLEAP] = 0.0; ts->tt[TT_DIV] = 0.0;
        ts->tt[TT_RHOSIDE] = -1.0; ts->tt[TT_N - 1] = 0.0;
      }
    }
  }
  while (true) {
    __syncthreads();
    if (TWIN) {
      if (uni_i(ts->tw_over) || uni_i(cl_dead)) break;
      const int dirs = uni_i(ts->tw_dirs);
      while (depth < c.max_depth && ((dirs >> depth) & 1) != ta.side) depth++;   // the doublings of the other end are not this side's business
      if (depth >= c.max_depth) break;
      // Speculation is free while the compute units would idle, but it is not useful beyond the depth trajectories reach: a
      // doubling that none of the last four transitions needed is started only once the trajectory has got there (in the
      // sampling phase of the 2016 posterior every tree has 8 doublings: the idle side then waits for the end instead of
      // integrating a ninth subtree that is always dropped, and leaves L2, HBM and power to the busy side).
      const unsigned sl = (unsigned)uni_i(c.sc->spec_limit);
      const int spec_limit = (int)max(max(sl & 0xffu, (sl >> 8) & 0xffu), max((sl >> 16) & 0xffu, sl >> 24));
      if (depth >= spec_limit && uni_i(ts->tw_seq) < depth) {
        if (tid < 64) {
          CRp R = (CRp)uni_ptr(ta.Rg);
          const Twin t = make_twin(R, ta.chain, ta.side, c.x.launch, iter);
          tw_catch_up(c.x, t, ts, depth, true);
        }
        __syncthreads();
        if (uni_i((int)ts->tt[TT_STOP])) {
          if (tid == 0) ts->tw_over = 1;
          continue;                                   // the loop head leaves
        }
      }
    } else {
      if (uni_i(ts->depth) >= c.max_depth || uni_i(ts->stop) || uni_i(cl_dead)) break;
      depth = uni_i(ts->depth);
    }
    c.x.epoch = uni32(c.x.epoch); c.x.x1e = uni32(c.x.x1e);
    if (tid == 0) {
      if (TWIN) {
        ts->dir = ta.side;
        ts->pmask = 0;
        ts->qmask = (1u << ts->tw_keep) | (1u << ts->nextq[ta.side]);
        ts->sum_metro = 0.0; ts->n_leap = 0; ts->divergent = 0;     // of this subtree; added to the trajectory's at the combine
      } else {
        ts->dir = rng_uniform(c.key, iter, RNG_DIRECTION, 0, (uint32_t)depth) > 0.5 ? 1 : 0;
        ts->pmask = 0;
        ts->qmask = (1u << ts->sample_qid) | (1u << ts->nextq[0]) | (1u << ts->nextq[1]);
        // the metropolis terms of a doubling are summed on their own and then added to the trajectory's sum -- the order twin
        // mode has to use: with it the two modes produce the same bytes
        ts->metro_base = ts->sum_metro; ts->sum_metro = 0.0;
      }
    }
    __syncthreads();
    const int dir = uni_i(ts->dir);
    CPROF_START(c);
    cl_vop_copy<false, TWIN>(c, c.soff(V_PNEAR), c.soff(V_PF0 + dir));   // (twin mode: the end momentum was stored write-through)
    if constexpr (RES) cl_fill_resident(c);
    CPROF_MARK(c, PF_PNEAR);
    bool valid = true;
    const int nleaf = 1 << depth;
    ldp wpart = FX ? (ldp)lds_dyn + ClFixed::l_wide : c.lds + c.CL->l_wide, wout = FX ? (ldp)lds_dyn + ClFixed::l_wout : c.lds + c.CL->l_wout;
    // Consecutive leaves of the subtree are software-pipelined: a leaf sends its totals (log density, kinetic energy,
    // U-turn dot products) and the next leaf starts at once; the totals are collected and the verdicts taken inside
    // that next pass (cl_pass_partial, `pend`).  Only the last leaf of the doubling waits for its own totals.
    LeafCtx pend{-1, 0, 0, 0, 0, 0, 0, 0, 0, 0u, -1};
    int prev_leaf = 0, prev_outq = 0;
    for (int n = 0; n < nleaf; n++) {
      if (tid == 0) {
        unsigned pm = ts->pmask, qm = ts->qmask;
        ts->leaf_id = pool_alloc(pm, PT_NPP); ts->pmask = pm;
        ts->out_q = pool_alloc(qm, PT_NPQ); ts->qmask = qm;   // slot receiving the position of the next leaf
      }
      __syncthreads();
      const double e = dir ? eps : -eps;
      const int inq = n == 0 ? uni_i(ts->nextq[dir]) : prev_outq, outq = uni_i(ts->out_q);   // slots of this leaf's position and of the one it produces
      const int leaf = uni_i(ts->leaf_id);
      c.x.epoch = uni32(c.x.epoch); c.x.x1e = uni32(c.x.x1e);
      const unsigned s_leaf = c.soff(V_POOLP + leaf);
      const int m = __builtin_ctz(~(unsigned)n);  // levels merged at this leaf
      const bool last = n == nleaf - 1;           // then m == depth
      const bool top = !TWIN && last;             // twin mode: the checks across the whole trajectory belong to the combine
      // (for an odd leaf the level-1 partner is the previous leaf: its momentum slot is known without its verdicts)
      auto make_policy = [&]() {
        if constexpr (RES)
          return ClLeapPolicyRes{c.st, c.soff(V_POOLQ + inq), c.soff(V_POOLQ + outq), c.soff(V_PH0 + dir), s_leaf, 0.5 * e, e, c.soff(V_RHOLEV + 1),
                                 m >= 1, m == 1, uni32(8u * (unsigned)c.e0), {0.0, 0.0, 0.0}};
        else
          return ClLeapPolicy{c.st, c.soff(V_POOLQ + inq), c.soff(V_POOLQ + outq), c.soff(V_PH0 + dir), c.soff(V_MINV),
                              s_leaf, 0.5 * e, e, c.soff(V_POOLP + (m >= 1 ? prev_leaf : 0)), m == 1 ? c.soff(V_RHOLEV + 1) : c.soff(V_SCR0 + 1),
                              m >= 1, {0.0, 0.0, 0.0}};
      };
      auto lp = make_policy();
      if (tid >= PT_THREADS - 64) {
        // the uniforms of this leaf's accept steps depend on nothing computed here: the last wave draws them
        // now (one lane per level) instead of thread 0 drawing them one after the other
        const int j = tid - (PT_THREADS - 64) + 1;
        if (j <= m) ts->u_sub[n & 1][j] = rng_uniform(c.key, iter, RNG_SUB_ACCEPT, 0, ((uint32_t)depth << 24) | ((uint32_t)j << 16) | (uint32_t)(n >> j));
        if (j == 64 && top) ts->u_top = rng_uniform(c.key, iter, RNG_TOP_ACCEPT, 0, (uint32_t)depth);
        if (TWIN) {
          // has the trajectory ended at the other end?  One look per leaf, by this wave while the others start the pass;
          // the flag travels with the leaf's totals, so every member of this cluster sees the same answer at the same leaf.
          CRp R = (CRp)uni_ptr(ta.Rg);
          const Twin t = make_twin(R, ta.chain, ta.side, c.x.launch, iter);
          double sv;
          const bool over = tw_try(t, TWB_STOP, 1, t.ittag, sv);
          if (j == 1) ts->tw_ext = over ? 1 : 0;
        }
      }
      bool aborted = false;
      const double lpp = cl_pass_partial<CL_DW>(c.M, c.CL, c.part, c.lds, c.cst, c.x, lp, !last, pend, ts, wout, aborted);
      if (aborted || uni_i(cl_dead)) { valid = false; c.x.x1e = 0; break; }   // the previous leaf ended the trajectory: this one is dropped unseen
      CPROF_START(c);
      CPROF_COUNT(c, PF_LEAVES);
      // One all-reduce per leaf: log density, kinetic energy and the six dot products of every U-turn check
      // this leaf completes (the subtrees of 2, 4, ... leaves that end here, and the whole new subtree against
      // the old trajectory when it is the last leaf of the doubling).  The sweeps only need vectors that are
      // already final; their verdicts are taken after the reduction, exactly in build_tree's order.
      const int nv0 = 2 + 6 * (m + (top ? 1 : 0)), nv = nv0 + (TWIN ? 1 : 0);
      {
        const int lane = tid & 63, w = tid >> 6;
        const double t0 = dpp_scan_sum(lpp), t1 = dpp_scan_sum(lp.extra[0]);
        if (lane == 63) { wpart[WP(0, w)] = t0; wpart[WP(1, w)] = t1; }
        if (m >= 1) {                               // level 1 came out of the epilogue: v0 = v2 = v4, v1 = v3 = v5
          const double a = dpp_scan_sum(lp.extra[1]), b = dpp_scan_sum(lp.extra[2]);
          if (lane == 63) {
#pragma unroll
            for (int k = 0; k < 6; k++) wpart[WP(2 + k, w)] = (k & 1) ? b : a;
          }
        }
        if (TWIN && lane == 63) wpart[WP(nv0, w)] = (w == 0 && ts->tw_ext) ? 1.0 : 0.0;
      }
      // the leaf's momentum is read back with another thread map: out of memory behind a drain, or (RES) out of LDS behind an LDS-only barrier
      if (top || (!RES && m > 1)) { drain_vmem(); __syncthreads(); }
      else if (m > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (m >= 2) cl_vop_merge_chain<RES>(c, ts, m, leaf, wpart);
      if (top) {
        // the checks at the end of transition(): old trajectory (init side) against the new subtree
        const int nb = depth >= 1 ? uni_i(ts->pend_beg[depth - 1]) : leaf;
        const unsigned n_rho = depth == 0 ? c.soff(V_POOLP + leaf) : c.soff(V_RHOLEV + depth);
        cl_vop_merge_partial(c, c.soff(V_PF1 - dir), c.soff(V_PNEAR), c.soff(V_RHOTOP), c.soff(V_POOLP + nb), c.soff(V_POOLP + leaf), n_rho,
                             c.soff(V_RHOTOP), wpart, 2 + 6 * m);
      }
      CPROF_MARK(c, PF_MERGE);
      const unsigned tag = cl_wide_publish(wpart, nv, c.x, CPROFPTR(c));
      const LeafCtx cur{n, m, top ? 1 : 0, depth, dir, leaf, inq, outq, nv, tag, TWIN ? nv0 : -1};
      prev_leaf = leaf; prev_outq = outq;
      if (!last) { pend = cur; CPROF_MARK(c, PF_LEAF_SCALAR); continue; }
      // last leaf of the doubling: its verdicts are needed before anything else can start
      if (tid < 64) {
        cl_wide_consume(c.x, tag, nv, wout);
        cl_leaf_logic(ts, cur, wout);
      }
      __syncthreads();
      CPROF_MARK(c, PF_LEAF_SCALAR);
      if (uni_i(ts->abort)) { valid = false; c.x.x1e = 0; break; }
      if (!TWIN) cl_vop_copy<false>(c, c.soff(V_PF0 + dir), c.soff(V_POOLP + leaf));   // the last leaf is the new end point
      CPROF_MARK(c, PF_COPYQ);
    }
    if (!TWIN && tid == 0) ts->sum_metro += ts->metro_base;   // (the verdicts of the doubling's last leaf are behind a barrier)
    if (TWIN) {
      c.x.epoch = uni32(cl_cold_twin_combine<CL_DW>(ta.Mg, ta.CLg, ta.Rg, ta.chain, ta.m, ta.side, ta.launch, c.x.epoch, iter, depth, valid ? 1 : 0, prev_leaf));
      depth++;
      continue;
    }
    if (!valid) break;
  }
  __syncthreads();
}

// base_hmc::init_stepsize; the chain's point is QC with gradient GC (already evaluated).
template <int CL_DW>
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
    ClLeapPolicy lp{c.st, c.soff(V_QA1), c.soff(V_QB1), c.soff(V_PH1), c.soff(V_MINV), c.soff(V_SCR0), 0.5 * eps, eps, 0u, 0u, false, {0.0, 0.0, 0.0}};
    const double lpv = cl_pass<CL_DW>(c.M, c.CL, c.part, c.lds, c.cst, c.x, lp);
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
          if (ne > 1e7 || ne == 0) { ts->done = 1; c.sc->status = POTUS_ERR_STEPSIZE; } // upstream throws here
        }
      }
    }
    __syncthreads();
    if (uni_i(ts->done) || uni_i(cl_dead)) break;
  }
  __syncthreads();
}

// adapt_diag_e_nuts::transition; the new sample is already the chain's point QC.
template <int CL_DW>
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
  const int in_window = uni_i(ts->flag_a), end_window = uni_i(ts->flag_b);
  const unsigned sQ = c.soff(V_QC), sMean = c.soff(V_WMEAN), sM2 = c.soff(V_WM2), sMinv = c.soff(V_MINV);
  if (in_window) { // welford_var_estimator::add_sample
    const double n = sc->wf_n;
    for (int i = cl_first(c); i < c.e1; i += PT_THREADS) {
      const double q = bld_s(c.st, 8u * i, sQ), mo = bld(c.st, 8u * i, sMean), delta = q - mo, mn = mo + delta / n;
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
    const double lpq = cl_pass<CL_DW>(c.M, c.CL, c.part, c.lds, c.cst, c.x, pol);
    if (tid == 0) sc->lp_cur = lpq;
    __syncthreads();
    cl_init_stepsize<CL_DW>(c, iter);
    if (tid == 0) { sc->mu = log(10.0 * sc->nom_eps); sc->s_bar = 0; sc->x_bar = 0; sc->ad_counter = 0; }
    __syncthreads();
  }
  if (tid == 0 && (int)iter == c.num_warmup - 1) sc->nom_eps = exp(sc->x_bar); // complete_adaptation
  __syncthreads();
}
