// potus_dense.hpp -- NUTS with a DENSE metric (stan::mcmc::dense_e_metric + covar_adaptation, `metric = dense_e`,
// BASELINE configs[4]) on gfx950.
//
// With a dense metric every leapfrog needs M^-1 times a momentum: 8 D^2 bytes of matrix per chain and leapfrog (1.8 GB
// for the 2016 posterior, 13.9 GB at D = 41 610) against 0.85 MB for everything else -- the one part of this path that
// is bound by HBM bandwidth, and work for the WHOLE chip rather than for the 16 compute units of a chain's cluster.
// So the dense sampler is not a persistent kernel per chain: it is a sequence of chip-wide launches per leaf,
//
//     gradient at the leaf's position  ->  kick (pf = ph + he g, ph' = pf + he g)  ->
//     k_dn_symv<2> + k_dn_symv_finish<2>:  p# = M^-1 pf  and  q' = q + e M^-1 ph'  in ONE pass over the upper triangle  ->
//     k_dn_step: the tree logic of base_nuts (one workgroup per chain, state kept in global memory between launches)
//
// with all chains of the handle advancing leaf by leaf together; a chain whose transition has ended sits out (its
// workgroups return at once: the launches are bound by the matrices they stream, so idle chains cost nothing).  The
// host only counts: it reads one flag per chain after each round to know when the transition is over for everybody.
// The position update of the NEXT leapfrog of the same trajectory end uses the same signed step, so its half-kicked
// momentum ph' is known as soon as this leaf's gradient is (the pre-kicked leapfrog of potus_nuts.hpp): both products
// of a leaf share one read of the matrix, and the arithmetic per leaf is exactly Stan's expl_leapfrog + dtau_dp.
//
// Adaptation (covar_adaptation): Stan streams a Welford covariance update per warm-up draw (16 D^2 bytes each).  Here
// the draws of a window are kept (n x D doubles, n <= 500) and the covariance is formed once at the window's end:
// the mean by Welford's own recurrence, the sum of outer products as a tiled rank-n update written straight into the
// inverse metric -- algebraically Welford's m2, no D^2 accumulator at all.  Then an in-place blocked Cholesky
// factorisation (the momentum draw p = L^-T u needs it), and base_hmc::init_stepsize.
//
// ONE matrix per chain holds both what the leapfrog and what the momentum draw need: M^-1 is symmetric, so its strict
// upper triangle (plus a vector for its diagonal) says everything, and the Cholesky factor L of the same matrix lives in
// the lower triangle including the diagonal -- the factorisation runs in place on the lower half and never touches the
// upper one.  The product M^-1 x then reads only the upper triangle: every element a_ij (i < j) serves y_i += a_ij x_j
// and y_j += a_ij x_i, i.e. 4 D^2 bytes per leapfrog instead of 8 D^2, and 13.9 GB per chain at D = 41 610 instead of
// 27.7: sixteen chains of the configs[4] shape fit one 288 GB GPU.
//
// Layout: per chain a block of DV_COUNT vectors of LD doubles (Stan's parameter order); the matrix row-major D x LD
// with LD = D rounded up to 8 doubles (rows start on 64-byte lines).
#pragma once
#include <type_traits>
#include "potus_dpp.hpp"
#include "potus_nuts.hpp"

#define DN_THREADS 512
#define DN_NB 64                                          // block size of the factorisation / triangular solve
#define DN_RB 128                                         // rows per block of the symmetric product: the unit of its column sums
#define DN_RB_BIG 256                                     // a workgroup takes DnParams::rb = 128, 256 or 512 rows at a time (16, 32 or 64 per wave), by the size of
#define DN_RB_MAX 512                                     // the matrix (potus_hmc.hip: dense_launch_shape).  Round 5: 512 from D = 16 384 on -- the per-tile costs of a
                                                          // block (x staged, three barriers, the column sums through LDS) are paid once per 512 rows instead of 256:
                                                          // 16 x 41 610: fp64 0.72 -> 0.74-0.75 of the peak, fp32 storage 0.64 -> 0.67 (profiles/r05_dense_maskless.txt)
#define DN_CT 512                                         // columns per tile of it (8 per lane)
#define DN_RG 4                                           // rows of a wave loaded together (16 loads of 16 bytes in flight per lane)
#define DN_FIN 256                                        // threads of the finishing kernel = elements per partial sum
#define DN_SPLIT_MAX 8                                    // at most this many workgroups share a pair of row blocks
#define DN_F32_PIPE_ROWS 2                                // fp32 storage: row groups software-pipelined, two buffers of this many rows (k_dn_symv)
typedef double dn_d2 __attribute__((ext_vector_type(2)));

// vector slots of a chain's dense state block
enum {
  DV_QC = 0, DV_GC, DV_G, DV_P0, DV_PH0, DV_PH1, DV_PF0, DV_PF1, DV_PSF0, DV_PSF1, DV_PNEAR, DV_PSNEAR,
  DV_RHOTOP, DV_SCR0, DV_SCR1, DV_TMPQ, DV_TMPP, DV_MEAN,
  DV_RHOLEV,                                /* PT_MAXD + 1 */
  DV_POOLP = DV_RHOLEV + PT_MAXD + 1,      /* leaf momenta */
  DV_POOLPS = DV_POOLP + PT_NPP,           /* their M^-1 p */
  DV_POOLQ = DV_POOLPS + PT_NPP,           /* trajectory positions / proposals */
  DV_COUNT = DV_POOLQ + PT_NPQ
};

// one right-hand side of a k_dn_symv launch, per chain (slots are absolute slot numbers, -1 = not used)
struct DnJob {
  int x;            // right-hand side
  int y;            // receives M^-1 x
  int qin, qout;    // qout = qin + coef * M^-1 x
  int dot;          // partial sums of dot . (M^-1 x) go to DnParams::partial (job 0 only)
  int pad;
  double coef;
};
// what the launches of one round do for a chain; written by k_dn_step / k_dn_eps_* of the previous round
struct DnRound {
  int active;       // 0: the chain sits this round out
  int qin;          // position the gradient is evaluated at
  int gout;         // slot receiving the gradient
  int ph;           // half-kicked momentum updated by the kick
  int leaf;         // slot receiving the full-step momentum
  int aux;          // RNG aux word of the next momentum draw (init_stepsize attempt)
  double he;        // signed half step
  DnJob job[3];
};

struct DnParams {
  int chains, D, LD, npart;          // npart = row blocks of the matvec = partial sums per chain
  int sc_stride, pad0;               // the chain scalars of chain c are RunParams::scal[c * sc_stride] (cluster mode keeps K replicas)
  double *state;                     // [chains][DV_COUNT][LD]
  double *A;                         // [chains][D][LD]: strict upper triangle = M^-1, lower triangle incl. diagonal = its Cholesky factor
  double *dg;                        // [chains][LD] diagonal of M^-1
  double *tpart;                     // [chains][nblk][3][LD] column sums of the symmetric product per block of DN_RB rows
  double *srow;                      // [chains][3][ntile][LD] its row sums per column tile (so that the result does not depend on
                                     // how the tiles are dealt to workgroups)
  int nblk, split;                   // blocks of DN_RB rows (allocation); workgroups sharing a pair of blocks (each takes every split-th tile)
  int rb, ntile;                     // rows a workgroup takes at a time (DN_RB or DN_RB_MAX, fixed per handle by D); column tiles
  double *win;                       // [chains][win_cap][LD] draws of the current adaptation window
  int win_cap, identity;             // identity: the metric is still the unit matrix (before the first window ends)
  double *partial;                   // [chains][npart]
  double *lpbuf;                     // [chains]
  TS *ts;                            // [chains] transition state
  DnRound *rd;                       // [chains]
  int *active;                       // [chains] copy of rd[].active for the host
  int *fail;                         // [1] Cholesky met a non-positive pivot
  int f32, pad1;                     // potus_opts.metric_storage = f32: M^-1 is kept rounded to fp32 (see dn_f32_row)
  unsigned long long *act_passes;    // [1] (chain, matrix pass) pairs that really ran: the bytes the passes streamed, whatever the host believed
  int count_passes;                  // this launch belongs to the timed set (transitions); init_stepsize / verification passes are neither timed nor counted
  // potus_opts.pooled_metric (potus_dense_pool.hpp): ONE inverse metric for all chains of the handle -- A is then a single FULL symmetric D x LD matrix,
  // dg one vector, the Cholesky factor lives in a buffer of its own
  int pooled;
  double *Lf;                        // [D][LD] pooled: the factor (lower triangle incl. diagonal)
  double *ypool;                     // [pool_split][DNP_RMAX][LD] pooled: the product's partial sums per row split
  double *pmean;                     // [LD] pooled: the mean of the handle's window draws (window end, potus_dense_pool.hpp)
  int pool_split, pool_rows;         // row splits of the pooled product (fixed per handle: the order of summation must not depend on who is active), rows per split
  float *A32;                        // [D][LD] pooled with metric_storage = f32: the matrix the pass streams (A holds the same rounded values as doubles)
};
// where a chain's matrix, factor and diagonal live (pooled: everybody's are the handle's one)
__device__ __host__ __forceinline__ double *dn_mat(const DnParams &P, int chain) { return P.pooled ? P.A : P.A + (size_t)chain * (size_t)P.D * (size_t)P.LD; }
__device__ __host__ __forceinline__ double *dn_fac(const DnParams &P, int chain) { return P.pooled ? P.Lf : P.A + (size_t)chain * (size_t)P.D * (size_t)P.LD; }
__device__ __host__ __forceinline__ double *dn_diag(const DnParams &P, int chain) { return P.pooled ? P.dg : P.dg + (size_t)chain * (size_t)P.LD; }

// fp32 storage of M^-1 (potus_opts.metric_storage): the matrix pass streams half the bytes.  The rounded matrix IS the metric:
// the lower triangle of the chain's buffer starts from the rounded values (as doubles) and becomes their Cholesky factor in
// fp64, so the momentum draw, the kinetic energy and the leapfrog all use exactly the same M^-1 and the sampler stays exact
// (a declared deviation from Stan, which adapts and stores the covariance in fp64).  Where the floats live: row i of the
// D x LD buffer holds L's row in doubles [0, i]; its second half is free, so element (i, j), j > i, is the float at index
// LD + j of the row seen as 2 LD floats -- double index (LD + j) / 2 > i: no overlap, no extra memory.
__device__ __host__ __forceinline__ float *dn_f32_row(double *A, int LD, int i) { return reinterpret_cast<float *>(A + (size_t)i * LD) + LD; }
__device__ __host__ __forceinline__ const float *dn_f32_row(const double *A, int LD, int i) { return reinterpret_cast<const float *>(A + (size_t)i * LD) + LD; }

// The chains that take part in a launch of the symmetric product, compacted: idle chains between active ones leave holes in
// the dispatch order, and the hardware then doubles workgroups up on some compute units while others idle (0.50 instead of
// 0.73 of the HBM peak with 3 scattered chains of 16; profiles/r02_dense_active_sweep.txt).  n == 0: every chain (blockIdx.y).
#define DN_ACT_MAX 64
struct DnActive {
  int n;
  int idx[DN_ACT_MAX];
};

__device__ __forceinline__ double *dn_vec(const DnParams &P, int chain, int slot) { return P.state + ((size_t)chain * DV_COUNT + slot) * (size_t)P.LD; }

// ---------------------------------------------------------------- M^-1 x for up to three right-hand sides
// Symmetric product out of the strict upper triangle.  A workgroup takes blocks of P.rb rows (128 or 256: 16 or 32 per
// wave) -- block b and its mirror nblk-1-b, so that every workgroup streams the same number of elements -- and walks the
// column tiles (DN_CT = 512) from the block's diagonal to the right edge.  A wave reads its rows' 4 KB segments of the
// tile (16 bytes per lane, DN_RG rows x 4 loads in flight per lane); every element feeds
//     the row sums    s_i += a_ij x_j   per lane, reduced per row and tile (DPP) into an LDS accumulator of the block and
//                                       stored per (tile, row),
//     the column sums t_j += a_ij x_i   per lane (its 8 columns of the tile), summed over the 8 waves through LDS per
//                                       tile and stored per (block, column): k_dn_symv_finish adds tiles and blocks in order.
// Fixed summation order everywhere: same bytes every run and for every launch shape.  Traffic besides the triangle: the
// column and row sums, written and read once = 3-4 % of the matrix.
// (job0: the first of the round's jobs this launch serves -- the three products of a transition's first pass go as 2 + 1)
#define DN_SYMV_LDS(NRHS) ((size_t)((NRHS) * (DN_CT + 2 * DN_RB_MAX) + (DN_THREADS / 64) * (NRHS) * DN_CT) * 8)
template <int NRHS, bool F32 = false>
__global__ __launch_bounds__(DN_THREADS) void k_dn_symv(const DnParams P, const DnActive act, int job0) {
  constexpr int NU = F32 ? 2 : 4, NE = F32 ? 4 : 2, RG = F32 ? 2 * DN_RG : DN_RG;   // 16-byte loads per lane and row, elements per load, rows in flight
  constexpr bool PIPE = F32;                         // fp32 storage: row groups software-pipelined (see the tile loop)
  constexpr int GR = PIPE ? DN_F32_PIPE_ROWS : RG;   // rows per group
  extern __shared__ __attribute__((aligned(16))) double dn_lds[];
  const int chain = act.n ? act.idx[blockIdx.y] : (int)blockIdx.y;
  const DnRound &rd = P.rd[chain];
  if (!rd.active) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = P.D, LD = P.LD, RB = P.rb, RW = RB / (DN_THREADS / 64), nblk = (D + RB - 1) / RB;
  double *xs = dn_lds;                                   // [NRHS][DN_CT]  x over the tile's columns
  double *xr = xs + NRHS * DN_CT;                        // [NRHS][RB]  x over the block's rows
  double *sacc = xr + NRHS * DN_RB_MAX;                  // [NRHS][RB]  row sums of the block, accumulated over the tiles
  double *tacc = sacc + NRHS * DN_RB_MAX;                // [8 waves][NRHS][DN_CT]
  const double *A = dn_mat(P, chain);
  const double *x[NRHS];
#pragma unroll
  for (int r = 0; r < NRHS; r++) x[r] = dn_vec(P, chain, rd.job[job0 + r].x);
  double *tp = P.tpart + ((size_t)chain * P.nblk * 3 + job0) * (size_t)LD;   // [block of this launch][job][LD] (room for the DN_RB blocks)
  // With few active chains a pair of blocks per workgroup does not fill the chip: `split` workgroups share a pair, each
  // taking every split-th column tile.  Column sums go to disjoint columns, row sums are stored per tile: the finishing
  // kernel adds both in a fixed order, so a chain's numbers do not depend on split, i.e. on what the other chains do.
  const int split = P.split, part = (int)blockIdx.x % split, pair = (int)blockIdx.x / split;
  double *sr = P.srow + ((size_t)chain * 3 + job0) * (size_t)P.ntile * (size_t)LD;   // [job][tile][LD]
  const unsigned rowbytes = uni32(8u * (unsigned)LD);
  for (int side = 0; side < 2; side++) {
    const int b = side == 0 ? pair : nblk - 1 - pair;
    if (side == 1 && b <= pair) break;                    // odd number of blocks: the middle one once
    const int r0 = b * RB, wrow0 = r0 + RW * w;
    // buffer addressing: the wave's DN_RW rows are one resource (rows beyond D fall outside it and read as zeros), the row is
    // a scalar offset, the lane's columns one 32-bit vector offset per load.  (Made wave-uniform explicitly: the compiler
    // otherwise keeps the descriptor in vector registers and wraps every load in a waterfall loop.)
    const int wrows = min(RW, max(0, D - wrow0));
    // (fp32 storage: the rows' float halves, dn_f32_row; the last row's ends the resource)
    const rsrc_t rsA = F32 ? make_rsrc(uni_ptr(reinterpret_cast<const char *>(A + (size_t)wrow0 * LD) + 4 * (size_t)LD),
                                       uni32(wrows > 0 ? (unsigned)(wrows - 1) * rowbytes + 4u * (unsigned)LD : 0u))
                           : make_rsrc(uni_ptr(A + (size_t)wrow0 * LD), uni32((unsigned)wrows * rowbytes));
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NRHS; r++)
      for (int i = tid; i < RB; i += DN_THREADS) { xr[r * DN_RB_MAX + i] = r0 + i < D ? x[r][r0 + i] : 0.0; sacc[r * DN_RB_MAX + i] = 0.0; }
    for (int c0 = (r0 / DN_CT + part) * DN_CT; c0 < D; c0 += split * DN_CT) {
      __syncthreads();                                    // the previous tile's column sums have been read
#pragma unroll
      for (int r = 0; r < NRHS; r++)
        for (int j = tid; j < DN_CT; j += DN_THREADS) xs[r * DN_CT + j] = c0 + j < D ? x[r][c0 + j] : 0.0;
      __syncthreads();
      double xc[NU][NE][NRHS];
      double t_acc[NU][NE][NRHS];
      unsigned voff[NU];
#pragma unroll
      for (int u = 0; u < NU; u++) {
        voff[u] = (F32 ? 4u : 8u) * (unsigned)(c0 + NE * (lane + 64 * u));
#pragma unroll
        for (int r = 0; r < NRHS; r++)
#pragma unroll
          for (int e = 0; e < NE; e++) { xc[u][e][r] = xs[r * DN_CT + NE * (lane + 64 * u) + e]; t_acc[u][e][r] = 0.0; }
      }
      // The tile overlaps the block's own rows (only j > i counts) or the matrix's right edge (fp32 storage: beyond column D a row's
      // float half runs into the next row's doubles, whose halves may read as NaN): elements are masked one by one.  Every other tile
      // -- 79 of 81 per row block at D = 41 610 -- takes the loop without the two compares and the select per element (
      // same products in the same order: same bytes).
      const bool band = c0 < r0 + RB;
      const bool masked = band || (F32 && c0 + DN_CT > D);
      auto rows = [&](auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
        // one group of GR rows: the loads ...
        auto request = [&](auto &buf, int q) {
#pragma unroll
          for (int k = 0; k < GR; k++)
#pragma unroll
            for (int u = 0; u < NU; u++)    // (rows beyond the wave's lie outside its resource: nothing is read, zeros come back)
              buf[k][u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voff[u], (unsigned)(q + k) * rowbytes, 2 /* nt: read once */);
        };
        // ... and their products
        auto consume = [&](const auto &buf, int q) {
#pragma unroll
          for (int k = 0; k < GR; k++) {
            const int lrow = RW * w + q + k;
            const int lim = band ? r0 + lrow : -1;          // no branch: outside the band every column counts
            double xrow[NRHS], sp[NRHS];
#pragma unroll
            for (int r = 0; r < NRHS; r++) { xrow[r] = xr[r * DN_RB_MAX + lrow]; sp[r] = 0.0; }
#pragma unroll
            for (int u = 0; u < NU; u++) {
#pragma unroll
              for (int e = 0; e < NE; e++) {
                const int col = c0 + NE * (lane + 64 * u) + e;
                double av;
                // (through a scalar: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index, hipcc 7.2)
                if constexpr (F32) { const unsigned wv = buf[k][u][e]; av = (double)__uint_as_float(wv); }
                else av = __hiloint2double((int)buf[k][u][2 * e + 1], (int)buf[k][u][2 * e]);   // (constexpr: the other branch would index past the vector)
                const double ae = (!MASKED || (col > lim && (!F32 || col < D))) ? av : 0.0;
#pragma unroll
                for (int r = 0; r < NRHS; r++) { sp[r] += ae * xc[u][e][r]; t_acc[u][e][r] += ae * xrow[r]; }
              }
            }
            // the row's sum over this tile: across the lanes by DPP (lane 63 ends up with it), then into the block's row sums
#pragma unroll
            for (int r = 0; r < NRHS; r++) {
              const double tot = dpp_scan_sum(sp[r]);   // (measured: without it the fp64 pass is no faster, the fp32 pass 6 %: r03_dense_storage_study.txt)
              if (lane == 63) sacc[r * DN_RB_MAX + lrow] += tot;
            }
          }
        };
        if constexpr (PIPE) {
          // fp32 storage: twice the arithmetic per byte (SQ_ACTIVE_INST_VALU 24.5 % of the wave cycles against 13.4 %, two waves per SIMD: the
          // vector pipe is half busy) while a wave that multiplies has nothing in flight.  Two buffers of GR = DN_F32_PIPE_ROWS rows: the next group's loads
          // are requested before the current group's products (same rows in the same order: same bytes).  Measured (profiles/r05_dense_maskless.txt):
          // two rows per buffer 0.62-0.63 -> 0.64-0.65 of the peak at 16 x 41 610 with 201 VGPRs instead of 256 + 4 spills; four rows per buffer spill
          // 55 registers into the loop and fall to 0.51; three buffers of two rows measure the same as two (0.64-0.65).  (Two workgroups per compute unit instead -- 128 VGPRs, amdgpu_waves_per_eu(4) -- spill
          // 21-118 registers: 0.53-0.63 with one row per buffer, 0.15 with two.)
          u32x4 a0[GR][NU], a1[GR][NU];
          request(a0, 0);
#pragma unroll 1
          for (int q = 0; q < RW; q += 2 * GR) {
            request(a1, q + GR);
            consume(a0, q);
            if (q + 2 * GR < RW) request(a0, q + 2 * GR);   // wave-uniform; the last trip has no next group (its rows would lie past the wave's: a
                                                            // buffer's scalar offset is not part of the bounds check -- ADVICE r05)
            consume(a1, q + GR);
          }
        } else {
#pragma unroll 1
          for (int q = 0; q < RW; q += GR) {                // GR = RG rows at a time: 16 loads of 16 bytes in flight per lane
            u32x4 a[GR][NU];
            request(a, q);
            consume(a, q);
          }
        }
      };
      if (masked) rows(std::true_type{}); else rows(std::false_type{});   // wave-uniform (the whole workgroup takes the same tile)
      // column sums of this (block, tile): over the 8 waves in wave order
#pragma unroll
      for (int u = 0; u < NU; u++)
#pragma unroll
        for (int r = 0; r < NRHS; r++)
#pragma unroll
          for (int e = 0; e < NE; e++) tacc[((size_t)w * NRHS + r) * DN_CT + NE * (lane + 64 * u) + e] = t_acc[u][e][r];
      __syncthreads();
      for (int e = tid; e < NRHS * DN_CT; e += DN_THREADS) {
        const int r = e / DN_CT, c = e - r * DN_CT;
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < DN_THREADS / 64; ww++) t += tacc[((size_t)ww * NRHS + r) * DN_CT + c];
        if (c0 + c < D) tp[((size_t)b * 3 + r) * LD + c0 + c] = t;
      }
      // the block's row sums over this tile (complete: the barrier above), and the accumulators cleared for the next one
      // (the barrier at the top of the loop separates this from the next tile's additions)
#pragma unroll
      for (int r = 0; r < NRHS; r++)
        for (int i = tid; i < RB; i += DN_THREADS) {
          if (r0 + i < D) sr[((size_t)r * P.ntile + c0 / DN_CT) * LD + r0 + i] = sacc[r * DN_RB_MAX + i];
          sacc[r * DN_RB_MAX + i] = 0.0;
        }
    }
  }
}
// y_i = diag_i x_i + the row sums of i's block over its column tiles (tile order) + the column sums of the blocks at or
// above row i (block order); then what the round wants done with the products: stores, the position update, partial sums
// of dot . y.  64 elements per workgroup, four threads per element: thread (i, k) adds the terms k, k + 4, ... of both
// series (up to 326 blocks at D = 41 610: one thread alone would walk them as a chain of dependent loads), then the
// four are added in order.  The order depends on i alone: same bytes every run, whatever the launch shape.
template <int NRHS>
__global__ __launch_bounds__(DN_FIN) void k_dn_symv_finish(const DnParams P, const DnActive act, int job0) {
  __shared__ double quad[NRHS][4][DN_FIN / 4];
  __shared__ double dn_part[DN_FIN / 64];
  const int chain = act.n ? act.idx[blockIdx.y] : (int)blockIdx.y;
  const DnRound &rd = P.rd[chain];
  if (!rd.active) return;
  const int e = threadIdx.x & (DN_FIN / 4 - 1), k4 = threadIdx.x / (DN_FIN / 4);
  const int i = blockIdx.x * (DN_FIN / 4) + e, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int LD = P.LD;
  const double *tp = P.tpart + ((size_t)chain * P.nblk * 3 + job0) * (size_t)LD;
  const double *sr = P.srow + ((size_t)chain * 3 + job0) * (size_t)P.ntile * (size_t)LD;
  const double *dg = dn_diag(P, chain);
  if (i < P.D) {
    const int nb = i / P.rb + 1, t_first = ((i / P.rb) * P.rb) / DN_CT;
#pragma unroll
    for (int r = 0; r < NRHS; r++) {
      double t = 0.0;
      for (int tt = t_first + k4; tt < P.ntile; tt += 4) t += sr[((size_t)r * P.ntile + tt) * LD + i];
      for (int b = k4; b < nb; b += 4) t += tp[((size_t)b * 3 + r) * LD + i];
      quad[r][k4][e] = t;
    }
  }
  __syncthreads();
  double dsum = 0.0;
  if (k4 == 0 && i < P.D) {
#pragma unroll
    for (int r = 0; r < NRHS; r++) {
      const DnJob &jb = rd.job[job0 + r];
      const double y = dg[i] * dn_vec(P, chain, jb.x)[i] + (((quad[r][0][e] + quad[r][1][e]) + quad[r][2][e]) + quad[r][3][e]);
      if (jb.y >= 0) dn_vec(P, chain, jb.y)[i] = y;
      if (jb.qout >= 0) dn_vec(P, chain, jb.qout)[i] = dn_vec(P, chain, jb.qin)[i] + jb.coef * y;
      if (job0 + r == 0 && jb.dot >= 0) dsum = dn_vec(P, chain, jb.dot)[i] * y;
    }
  }
  const double tot = dpp_wave_sum(dsum);
  if (lane == 0) dn_part[w] = tot;
  __syncthreads();
  if (threadIdx.x == 0 && job0 == 0) P.partial[(size_t)chain * P.npart + blockIdx.x] = dn_part[0];   // (k4 == 0 is wave 0)
  if (threadIdx.x == 0 && blockIdx.x == 0 && P.act_passes && P.count_passes) atomicAdd(P.act_passes, 1ull);
}

// ---------------------------------------------------------------- elementwise pieces of a round
// pf = ph + he g -> leaf slot; ph' = pf + he g -> ph   (end_update_p of this leaf merged with begin_update_p of the next)
__global__ __launch_bounds__(256) void k_dn_kick(const DnParams P) {
  const int chain = blockIdx.y;
  const DnRound &rd = P.rd[chain];
  if (!rd.active) return;
  const double *g = dn_vec(P, chain, rd.gout);
  double *ph = dn_vec(P, chain, rd.ph), *leaf = dn_vec(P, chain, rd.leaf);
  const double he = rd.he;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P.D; i += gridDim.x * 256) {
    const double pf = ph[i] + he * g[i];
    leaf[i] = pf;
    ph[i] = pf + he * g[i];
  }
}

// standard normals for a momentum draw, element i = normal (i & 1) of Philox block i >> 1 (the contract shared with
// the oracle and the diagonal samplers); the triangular solve that follows turns them into p ~ N(0, M)
__global__ __launch_bounds__(256) void k_dn_normals(const DnParams P, const RunParams *Rg, unsigned iter, unsigned purpose) {
  const int chain = blockIdx.y;
  const DnRound &rd = P.rd[chain];
  if (!rd.active) return;
  const RngKey key{Rg->seed_lo, Rg->seed_hi, (uint32_t)(Rg->chain_id_offset + chain + 1)};
  double *p = dn_vec(P, chain, DV_P0);
  for (int j = blockIdx.x * 256 + threadIdx.x; 2 * j < P.D; j += gridDim.x * 256) {
    double a, b;
    rng_normal_pair(key, iter, purpose, (uint32_t)rd.aux, (uint32_t)j, a, b);
    p[2 * j] = a;
    if (2 * j + 1 < P.D) p[2 * j + 1] = b;
  }
}

// L^T p = u in place on DV_P0 (dense_e_metric::sample_p: p = llt.matrixU().solve(u)), blocked back substitution:
// for block b from the last to the first: k_dn_trsv_diag solves the DN_NB x DN_NB triangle, k_dn_trsv_update
// subtracts the block's contribution from every earlier unknown.
__global__ __launch_bounds__(64) void k_dn_trsv_diag(const DnParams P, int b) {
  __shared__ double Lb[DN_NB][DN_NB + 1];
  const int chain = blockIdx.x;
  if (!P.rd[chain].active) return;
  const int lane = threadIdx.x, r0 = b * DN_NB, nb = min(DN_NB, P.D - r0);
  const double *L = dn_fac(P, chain);
  // (the block's rows sixteen loads at a time: one row per trip of a loop whose every trip waited for its own load made the 64 x 64 triangle 30 of
  //  the kernel's 39 us -- 651 blocks per momentum draw at D = 41 610, 9 % of a pooled-metric run: profiles/r06_dense_pooled_rocprof_summary.txt)
#pragma unroll
  for (int i0 = 0; i0 < DN_NB; i0 += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { const int i = i0 + u; v[u] = (i < nb && lane <= i) ? L[(size_t)(r0 + i) * P.LD + r0 + lane] : 0.0; }
#pragma unroll
    for (int u = 0; u < 16; u++) Lb[i0 + u][lane] = v[u];
  }
  __syncthreads();
  double *p = dn_vec(P, chain, DV_P0);
  double r = lane < nb ? p[r0 + lane] : 0.0;
  for (int i = nb - 1; i >= 0; i--) {
    const double pi = __shfl(r, i, 64) / Lb[i][i];
    if (lane == i) r = pi;
    else if (lane < i) r -= Lb[i][lane] * pi;
  }
  if (lane < nb) p[r0 + lane] = r;
}
__global__ __launch_bounds__(256) void k_dn_trsv_update(const DnParams P, int b) {
  __shared__ double pb[DN_NB];
  const int chain = blockIdx.y;
  if (!P.rd[chain].active) return;
  const int r0 = b * DN_NB, nb = min(DN_NB, P.D - r0);
  const double *L = dn_fac(P, chain);
  double *p = dn_vec(P, chain, DV_P0);
  if (threadIdx.x < DN_NB) pb[threadIdx.x] = (int)threadIdx.x < nb ? p[r0 + threadIdx.x] : 0.0;
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= r0) return;
  double s = 0.0;
  for (int i0 = 0; i0 < nb; i0 += 16) {             // sixteen loads in flight, the products added in the same order as ever (same bytes)
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = i0 + u < nb ? L[(size_t)(r0 + i0 + u) * P.LD + c] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; u++) if (i0 + u < nb) s += v[u] * pb[i0 + u];
  }
  p[c] -= s;
}

// ---------------------------------------------------------------- the tree logic (base_nuts::transition / build_tree)
// Same iterative restatement as potus_nuts.hpp / potus_cluster.hpp: a loop over leaves with a cascade of merges, every
// trajectory position in a slot of the proposal pool, proposals kept by index.  Differences: the state survives
// between launches in global memory, and p# = M^-1 p is a stored vector per leaf instead of minv * p on the fly.
struct DnSweep { int a_beg, as_beg, a_end, as_end, a_rho, b_beg, bs_beg, b_end, bs_end, b_rho, out; };

__device__ __forceinline__ bool dn_merge(const DnParams &P, int chain, const DnSweep &s, double *red_) {
  ldp red = (ldp)red_;
  const double *ab = dn_vec(P, chain, s.a_beg), *abs_ = dn_vec(P, chain, s.as_beg), *ae = dn_vec(P, chain, s.a_end), *aes = dn_vec(P, chain, s.as_end);
  const double *ar = dn_vec(P, chain, s.a_rho), *bb = dn_vec(P, chain, s.b_beg), *bbs = dn_vec(P, chain, s.bs_beg), *be = dn_vec(P, chain, s.b_end);
  const double *bes = dn_vec(P, chain, s.bs_end), *br = dn_vec(P, chain, s.b_rho);
  double *out = dn_vec(P, chain, s.out);
  double v[6] = {0, 0, 0, 0, 0, 0};
  const int tid = threadIdx.x;
  for (int i = tid; i < P.D; i += DN_THREADS) {
    const double ra = ar[i], rb = br[i], rs = ra + rb;
    const double sab = abs_[i], sbe = bes[i];
    v[0] += sab * rs;                       // p#_beg . rho_subtree
    v[1] += sbe * rs;                       // p#_end . rho_subtree
    const double e1 = ra + bb[i];           // rho_init + p_final_beg
    v[2] += sab * e1;
    v[3] += bbs[i] * e1;
    const double e2 = rb + ae[i];           // rho_final + p_init_end
    v[4] += aes[i] * e2;
    v[5] += sbe * e2;
    out[i] = rs;
  }
  (void)ab; (void)be;
  block_sum(v, red, tid);
  return v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0 && v[4] > 0 && v[5] > 0;
}
__device__ __forceinline__ void dn_copy(const DnParams &P, int chain, int dst, int src) {
  const double *s = dn_vec(P, chain, src);
  double *d = dn_vec(P, chain, dst);
  for (int i = threadIdx.x; i < P.D; i += DN_THREADS) d[i] = s[i];
}
__device__ __forceinline__ double dn_partial_sum(const DnParams &P, int chain) {   // fixed order over the row blocks
  double t = 0.0;
  const double *p = P.partial + (size_t)chain * P.npart;
  for (int i = 0; i < P.npart; i++) t += p[i];
  return t;
}

// the same sum by a whole workgroup (k_dn_step): strided partial sums per thread, then the block tree -- fixed order too
__device__ __forceinline__ double dn_partial_sum_block(const DnParams &P, int chain, double *red_) {
  const double *p = P.partial + (size_t)chain * P.npart;
  double v[1] = {0.0};
  for (int i = threadIdx.x; i < P.npart; i += DN_THREADS) v[0] += p[i];
  block_sum(v, (ldp)red_, (int)threadIdx.x);
  return v[0];
}

enum { DN_MODE_BEGIN = 0, DN_MODE_LEAF = 1 };

// called by thread 0: the next leaf of the current doubling
__device__ __forceinline__ void dn_setup_leaf(TS &ts, DnRound &rd) {
  unsigned pm = ts.pmask, qm = ts.qmask;
  const int leaf = pool_alloc(pm, PT_NPP), outq = pool_alloc(qm, PT_NPQ);
  ts.pmask = pm; ts.qmask = qm;
  ts.leaf_id = leaf; ts.out_q = outq;
  const int dir = ts.dir, inq = ts.nextq[dir];
  const double e = dir ? ts.eps : -ts.eps;
  rd.active = 1; rd.qin = DV_POOLQ + inq; rd.gout = DV_G; rd.ph = DV_PH0 + dir; rd.leaf = DV_POOLP + leaf; rd.he = 0.5 * e;
  rd.job[0] = DnJob{DV_POOLP + leaf, DV_POOLPS + leaf, -1, -1, DV_POOLP + leaf, 0, 0.0};          // p# of the leaf, kinetic energy
  rd.job[1] = DnJob{DV_PH0 + dir, -1, DV_POOLQ + inq, DV_POOLQ + outq, -1, 0, e};                // the next position of this end
}

__global__ __launch_bounds__(DN_THREADS) void k_dn_step(const DnParams P, const RunParams *Rg, unsigned iter, int mode) {
  __shared__ TS ts;
  __shared__ DnRound rd;
  __shared__ double red[(DN_THREADS / 64) * PT_NRED];
  __shared__ int sh_flag[4];
  const int chain = blockIdx.x, tid = threadIdx.x;
  static_assert(DN_THREADS / 64 == PT_NW, "block_sum is written for PT_NW waves");
  if (!P.rd[chain].active) return;
  for (int i = tid; i < (int)(sizeof(TS) / 4); i += DN_THREADS) ((int *)&ts)[i] = ((const int *)&P.ts[chain])[i];
  for (int i = tid; i < (int)(sizeof(DnRound) / 4); i += DN_THREADS) ((int *)&rd)[i] = ((const int *)&P.rd[chain])[i];
  __syncthreads();
  const RngKey key{Rg->seed_lo, Rg->seed_hi, (uint32_t)(Rg->chain_id_offset + chain + 1)};
  const int max_depth = Rg->max_depth;
  bool new_doubling = false;
  const double kin_pass = dn_partial_sum_block(P, chain, red);   // p . M^-1 p of the pass just finished
  if (mode == DN_MODE_BEGIN) {
    if (tid == 0) {
      const double kin0 = kin_pass, lp0 = P.lpbuf[chain];
      ts.H0 = 0.5 * kin0 - lp0;
      ts.lsw = 0.0; ts.sum_metro = 0.0; ts.n_leap = 0; ts.depth = 0; ts.divergent = 0; ts.stop = 0;
      ts.sample_qid = 0; ts.nextq[1] = 1; ts.nextq[0] = 2;
      ts.q_lp[0] = lp0; ts.q_h[0] = ts.H0;
    }
    dn_copy(P, chain, DV_PSF1, DV_PSF0);      // both ends start at the initial point
    new_doubling = true;
    __syncthreads();
  } else {
    const int n = ts.m;                        // leaf number inside the doubling (kept in ts.m between launches)
    const int depth = ts.depth, dir = ts.dir, leaf = ts.leaf_id;
    if (tid == 0) {
      const double kin = kin_pass, lpv = P.lpbuf[chain], H0 = ts.H0;
      double h = 0.5 * kin - lpv;
      if (isnan(h)) h = INFINITY;
      const int div = (h - H0 > 1000.0) ? 1 : ts.divergent;
      ts.divergent = div;
      const double wgt = H0 - h;
      ts.sum_metro += wgt > 0 ? 1.0 : exp(wgt);
      ts.n_leap += 1;
      const int inq = ts.nextq[dir];
      ts.cur_beg = ts.cur_end = leaf; ts.cur_lsw = wgt; ts.cur_prop = inq;
      ts.q_lp[inq] = lpv; ts.q_h[inq] = h;
      ts.nextq[dir] = ts.out_q;               // the next leaf of this end evaluates the position just written
      ts.abort = div;
    }
    __syncthreads();
    const int m = __builtin_ctz(~(unsigned)n);
    const bool top = n == (1 << depth) - 1;
    for (int j = 1; j <= m && !ts.abort; j++) {
      const int ib = ts.pend_beg[j - 1], ie = ts.pend_end[j - 1], cb = ts.cur_beg, ce = ts.cur_end;
      const DnSweep s{DV_POOLP + ib, DV_POOLPS + ib, DV_POOLP + ie, DV_POOLPS + ie, j == 1 ? DV_POOLP + ib : DV_RHOLEV + j - 1,
                      DV_POOLP + cb, DV_POOLPS + cb, DV_POOLP + ce, DV_POOLPS + ce, j == 1 ? DV_POOLP + cb : DV_SCR0 + ((j - 1) & 1),
                      j == m ? DV_RHOLEV + j : DV_SCR0 + (j & 1)};
      const bool persist = dn_merge(P, chain, s, red);
      if (tid == 0) {
        const double cur_lsw = ts.cur_lsw;
        const double lsw_sub = d_lse(ts.pend_lsw[j - 1], cur_lsw);
        bool take_final;
        if (cur_lsw > lsw_sub) take_final = true;
        else {
          const uint32_t slot = ((uint32_t)depth << 24) | ((uint32_t)j << 16) | (uint32_t)(n >> j);
          take_final = rng_uniform(key, iter, RNG_SUB_ACCEPT, 0, slot) < exp(cur_lsw - lsw_sub);
        }
        unsigned qm = ts.qmask, pm = ts.pmask;
        if (take_final) pool_free(qm, ts.pend_prop[j - 1]);
        else { pool_free(qm, ts.cur_prop); ts.cur_prop = ts.pend_prop[j - 1]; }
        if (ie != ib) pool_free(pm, ie);
        if (cb != ce) pool_free(pm, cb);
        ts.qmask = qm; ts.pmask = pm;
        ts.cur_beg = ib;
        ts.cur_lsw = lsw_sub;
        ts.abort = !persist;
      }
      __syncthreads();
    }
    if (!ts.abort) {
      if (tid == 0) { ts.pend_beg[m] = ts.cur_beg; ts.pend_end[m] = ts.cur_end; ts.pend_lsw[m] = ts.cur_lsw; ts.pend_prop[m] = ts.cur_prop; }
      __syncthreads();
      if (top) {
        // the last leaf is the new end point; then the checks at the end of transition(): old trajectory against the
        // new subtree (a = old, b = new; the six products are symmetric in the direction, see the note at dn_merge's caller)
        dn_copy(P, chain, DV_PF0 + dir, DV_POOLP + leaf);
        dn_copy(P, chain, DV_PSF0 + dir, DV_POOLPS + leaf);
        __syncthreads();
        const int nb = ts.pend_beg[depth];
        const DnSweep s{DV_PF1 - dir, DV_PSF1 - dir, DV_PNEAR, DV_PSNEAR, DV_RHOTOP,
                        DV_POOLP + nb, DV_POOLPS + nb, DV_POOLP + leaf, DV_POOLPS + leaf, depth == 0 ? DV_POOLP + leaf : DV_RHOLEV + depth, DV_RHOTOP};
        const bool persist = dn_merge(P, chain, s, red);
        if (tid == 0) {
          ts.depth = depth + 1;
          const double lsw_sub = ts.pend_lsw[depth], lsw = ts.lsw;
          bool accept;
          if (lsw_sub > lsw) accept = true;
          else accept = rng_uniform(key, iter, RNG_TOP_ACCEPT, 0, (uint32_t)depth) < exp(lsw_sub - lsw);
          unsigned qm = ts.qmask;
          if (accept) { pool_free(qm, ts.sample_qid); ts.sample_qid = ts.pend_prop[depth]; }
          else pool_free(qm, ts.pend_prop[depth]);
          ts.qmask = qm;
          ts.lsw = d_lse(lsw, lsw_sub);
          if (!persist) ts.stop = 1;
        }
        new_doubling = true;
        __syncthreads();
      } else if (tid == 0) {
        ts.m = n + 1;
        dn_setup_leaf(ts, rd);
      }
    } else if (tid == 0) rd.active = 0;       // divergence or U-turn inside the new subtree: the transition is over
  }
  if (new_doubling) {
    if (tid == 0) {
      if (ts.depth >= max_depth || ts.stop) { rd.active = 0; sh_flag[0] = 0; }
      else {
        ts.dir = rng_uniform(key, iter, RNG_DIRECTION, 0, (uint32_t)ts.depth) > 0.5 ? 1 : 0;
        ts.pmask = 0;
        ts.qmask = (1u << ts.sample_qid) | (1u << ts.nextq[0]) | (1u << ts.nextq[1]);
        ts.m = 0;
        sh_flag[0] = 1;
      }
    }
    __syncthreads();
    if (sh_flag[0]) {
      const int dir = ts.dir;
      dn_copy(P, chain, DV_PNEAR, DV_PF0 + dir);
      dn_copy(P, chain, DV_PSNEAR, DV_PSF0 + dir);
      if (tid == 0) dn_setup_leaf(ts, rd);
    }
  }
  __syncthreads();
  for (int i = tid; i < (int)(sizeof(TS) / 4); i += DN_THREADS) ((int *)&P.ts[chain])[i] = ((const int *)&ts)[i];
  for (int i = tid; i < (int)(sizeof(DnRound) / 4); i += DN_THREADS) ((int *)&P.rd[chain])[i] = ((const int *)&rd)[i];
  if (tid == 0) P.active[chain] = rd.active;
}

// ---------------------------------------------------------------- transition begin / end
// every chain takes part in a transition: (re)arm the round descriptors
__global__ void k_dn_arm(const DnParams P, const RunParams *Rg, int total_iters) {
  const int chain = blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= P.chains) return;
  const ChainScalars &sc = Rg->scal[(size_t)chain * P.sc_stride];
  const int on = sc.status == 0 && sc.iter < total_iters;
  DnRound &rd = P.rd[chain];
  rd.active = on; rd.qin = DV_QC; rd.gout = DV_GC; rd.aux = 0;
  P.active[chain] = on;
}
// after the momentum draw and the gradient at the current point (hamiltonian.init): both trajectory ends := the
// initial point, kicked half a step towards their first leaves; the three products of the begin pass
__global__ __launch_bounds__(256) void k_dn_begin(const DnParams P, const RunParams *Rg) {
  const int chain = blockIdx.y;
  DnRound &rd = P.rd[chain];
  if (!rd.active) return;
  const double eps = Rg->scal[(size_t)chain * P.sc_stride].nom_eps;   // sample_stepsize(): no jitter
  const double *q = dn_vec(P, chain, DV_QC), *g = dn_vec(P, chain, DV_GC), *p = dn_vec(P, chain, DV_P0);
  double *ph0 = dn_vec(P, chain, DV_PH0), *ph1 = dn_vec(P, chain, DV_PH1), *pf0 = dn_vec(P, chain, DV_PF0), *pf1 = dn_vec(P, chain, DV_PF1);
  double *rt = dn_vec(P, chain, DV_RHOTOP), *qs = dn_vec(P, chain, DV_POOLQ + 0);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P.D; i += gridDim.x * 256) {
    const double pi = p[i], gi = g[i];
    ph1[i] = pi + 0.5 * eps * gi; ph0[i] = pi - 0.5 * eps * gi;
    pf0[i] = pi; pf1[i] = pi; rt[i] = pi; qs[i] = q[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    P.ts[chain].eps = eps;
    rd.job[0] = DnJob{DV_P0, DV_PSF0, -1, -1, DV_P0, 0, 0.0};                 // p# of the initial point, kinetic energy
    rd.job[1] = DnJob{DV_PH1, -1, DV_POOLQ + 0, DV_POOLQ + 1, -1, 0, eps};    // first position forwards
    rd.job[2] = DnJob{DV_PH0, -1, DV_POOLQ + 0, DV_POOLQ + 2, -1, 0, -eps};   // first position backwards
  }
}

// New sample -> chain position and draws array; step-size adaptation; the draw joins the adaptation window.
// flags (host-computed from the shared warm-up schedule): bit 0 = inside a window, bit 1 = the window ends here.
__global__ __launch_bounds__(DN_THREADS) void k_dn_end(const DnParams P, const RunParams *Rg, int it, int flags, int win_n, int total_iters) {
  const int chain = blockIdx.x, tid = threadIdx.x;
  ChainScalars &sc = Rg->scal[(size_t)chain * P.sc_stride];
  if (sc.status != 0 || sc.iter != it || it >= total_iters) return;   // (chains that sat the transition out)
  TS &ts = P.ts[chain];
  const bool warm = it < Rg->num_warmup, save = !warm || Rg->save_warmup;
  const int sample = ts.sample_qid;
  const double *qs = dn_vec(P, chain, DV_POOLQ + sample);
  double *Q0 = dn_vec(P, chain, DV_QC);
  double *row = Rg->draws + ((size_t)chain * Rg->n_save_max + sc.saved) * Rg->row;
  double *wrow = (flags & 1) ? P.win + ((size_t)chain * P.win_cap + win_n) * (size_t)P.LD : nullptr;
  for (int i = tid; i < P.D; i += DN_THREADS) {
    const double v = qs[i];
    Q0[i] = v;
    if (save) row[POTUS_N_SAMPLER_COLS + i] = v;
    if (wrow) wrow[i] = v;
  }
  __syncthreads();
  if (tid == 0) {
    const double accept_stat = ts.sum_metro / (double)ts.n_leap;
    if (save) {
      row[0] = ts.q_lp[sample]; row[1] = accept_stat; row[2] = ts.eps; row[3] = ts.depth; row[4] = ts.n_leap; row[5] = ts.divergent;
      row[6] = ts.q_h[sample];
      sc.saved += 1;
    }
    sc.total_leapfrogs += ts.n_leap; sc.n_divergent += ts.divergent; sc.lp_cur = ts.q_lp[sample];
    if (warm) {   // stepsize_adaptation::learn_stepsize
      const double cnt = sc.ad_counter + 1;
      sc.ad_counter = cnt;
      const double as = accept_stat > 1 ? 1.0 : accept_stat;
      const double eta = 1.0 / (cnt + Rg->t0);
      const double s_bar = (1.0 - eta) * sc.s_bar + eta * (Rg->delta - as);
      sc.s_bar = s_bar;
      const double x = sc.mu - s_bar * sqrt(cnt) / Rg->gamma;
      const double x_eta = pow(cnt, -Rg->kappa);
      sc.x_bar = (1.0 - x_eta) * sc.x_bar + x_eta * x;
      sc.nom_eps = exp(x);
      if (!(flags & 2) && it == Rg->num_warmup - 1) sc.nom_eps = exp(sc.x_bar);   // complete_adaptation
    }
    sc.iter = it + 1;
  }
}
// after the metric update and init_stepsize at a window's end: restart the step-size adaptation around the new step
__global__ void k_dn_window_done(const DnParams P, const RunParams *Rg, int it) {
  const int chain = blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= P.chains) return;
  ChainScalars &sc = Rg->scal[(size_t)chain * P.sc_stride];
  if (sc.status != 0) return;
  sc.mu = log(10.0 * sc.nom_eps); sc.s_bar = 0; sc.x_bar = 0; sc.ad_counter = 0;
  if (it == Rg->num_warmup - 1) sc.nom_eps = exp(sc.x_bar);
}

// ---------------------------------------------------------------- base_hmc::init_stepsize, one attempt per round
// round: normals -> solve -> k_dn_eps_prekick -> matvec<2> (H0, first position) -> gradient -> k_dn_kick ->
// matvec<1> (kinetic energy at the end) -> k_dn_eps_step (direction / done / doubling or halving)
__global__ void k_dn_eps_arm(const DnParams P, const RunParams *Rg) {
  const int chain = blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= P.chains) return;
  const ChainScalars &sc = Rg->scal[(size_t)chain * P.sc_stride];
  const double e0 = sc.nom_eps;
  const int on = sc.status == 0 && !(e0 == 0 || e0 > 1e7 || isnan(e0));
  DnRound &rd = P.rd[chain];
  rd.active = on; rd.aux = 0; rd.qin = DV_QC; rd.gout = DV_GC;
  P.ts[chain].direction = 0; P.ts[chain].done = 0;
  P.active[chain] = on;
}
__global__ __launch_bounds__(256) void k_dn_eps_prekick(const DnParams P, const RunParams *Rg) {
  const int chain = blockIdx.y;
  DnRound &rd = P.rd[chain];
  if (!rd.active) return;
  const double eps = Rg->scal[(size_t)chain * P.sc_stride].nom_eps;
  const double *g = dn_vec(P, chain, DV_GC), *p = dn_vec(P, chain, DV_P0);
  double *ph = dn_vec(P, chain, DV_PH1);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P.D; i += gridDim.x * 256) ph[i] = p[i] + 0.5 * eps * g[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    rd.job[0] = DnJob{DV_P0, -1, -1, -1, DV_P0, 0, 0.0};
    rd.job[1] = DnJob{DV_PH1, -1, DV_QC, DV_TMPQ, -1, 0, eps};
    rd.qin = DV_TMPQ; rd.gout = DV_G; rd.ph = DV_PH1; rd.leaf = DV_TMPP; rd.he = 0.5 * eps;
  }
}
// between the two matrix passes of an attempt: H0 from the first, the job of the second
__global__ void k_dn_eps_mid(const DnParams P, const RunParams *Rg) {
  const int chain = blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= P.chains) return;
  DnRound &rd = P.rd[chain];
  if (!rd.active) return;
  P.ts[chain].H0 = 0.5 * dn_partial_sum(P, chain) - Rg->scal[(size_t)chain * P.sc_stride].lp_cur;
  rd.job[0] = DnJob{DV_TMPP, -1, -1, -1, DV_TMPP, 0, 0.0};
}
__global__ void k_dn_eps_step(const DnParams P, const RunParams *Rg) {
  const int chain = blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= P.chains) return;
  DnRound &rd = P.rd[chain];
  if (!rd.active) return;
  TS &ts = P.ts[chain];
  ChainScalars &sc = Rg->scal[(size_t)chain * P.sc_stride];
  double h = 0.5 * dn_partial_sum(P, chain) - P.lpbuf[chain];
  if (isnan(h)) h = INFINITY;
  const double delta_H = ts.H0 - h, thr = log(0.8), eps = sc.nom_eps;
  int done = 0;
  if (rd.aux == 0) ts.direction = delta_H > thr ? 1 : -1;
  else {
    const int dirn = ts.direction;
    if (dirn == 1 && !(delta_H > thr)) done = 1;
    else if (dirn == -1 && !(delta_H < thr)) done = 1;
    else {
      const double ne = dirn == 1 ? 2.0 * eps : 0.5 * eps;
      sc.nom_eps = ne;
      if (ne > 1e7 || ne == 0) { done = 1; sc.status = POTUS_ERR_STEPSIZE; }   // upstream throws here
    }
  }
  rd.aux += 1;
  rd.qin = DV_QC; rd.gout = DV_GC;
  if (done) rd.active = 0;
  P.active[chain] = rd.active;
}

// ---------------------------------------------------------------- covar_adaptation at a window's end
// mean by welford_covar_estimator's recurrence, draws centred in place
__global__ __launch_bounds__(256) void k_dn_center(const DnParams P, int n) {
  const int chain = blockIdx.y;
  double *W = P.win + (size_t)chain * P.win_cap * (size_t)P.LD;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < P.D; j += gridDim.x * 256) {
    double m = 0.0;
    for (int k = 0; k < n; k++) m += (W[(size_t)k * P.LD + j] - m) / (double)(k + 1);
    for (int k = 0; k < n; k++) W[(size_t)k * P.LD + j] -= m;
  }
}
// The two GEMM-shaped pieces of the window end -- the covariance (a rank-n update) and the trailing update of the
// factorisation -- run on the fp64 matrix cores: v_mfma_f64_16x16x4_f64, D[16 x 16] += A[16 x 4] B[4 x 16], lane l feeding
// A[l & 15][l >> 4] and B[l >> 4][l & 15] and holding D[(l >> 4) + 4 v][l & 15], v = 0..3.  A workgroup of four waves owns a
// 64 x 64 tile: wave w the 16 rows 16 w .. 16 w + 15, four accumulators (the four 16-column blocks), operands out of LDS.
typedef double dn_d4 __attribute__((ext_vector_type(4)));

// M^-1 = n/(n+5) * (sum_k c_k c_k') / (n-1) + 1e-3 * 5/(n+5) * I   (covar_adaptation::learn_covariance), tiles of 64 x 64,
// lower tiles computed and mirrored so that the matrix is exactly symmetric
__global__ __launch_bounds__(256) void k_dn_cov(const DnParams P, int n) {
  __shared__ double a[DN_NB][DN_NB + 1], b[DN_NB][DN_NB + 1];
  const int I = blockIdx.x, J = blockIdx.y, chain = blockIdx.z;
  if (J > I) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 15, kk = lane >> 4;
  const double *W = P.win + (size_t)chain * P.win_cap * (size_t)P.LD;
  double *A = P.A + (size_t)chain * (size_t)P.D * (size_t)P.LD;
  double *dg = P.dg + (size_t)chain * P.LD;
  dn_d4 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; cb++) acc[cb] = dn_d4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < n; k0 += DN_NB) {
    __syncthreads();
    for (int e = tid; e < DN_NB * DN_NB; e += 256) {
      const int k = e >> 6, j = e & 63;
      const bool ok = k0 + k < n;
      a[k][j] = (ok && I * DN_NB + j < P.D) ? W[(size_t)(k0 + k) * P.LD + I * DN_NB + j] : 0.0;
      b[k][j] = (ok && J * DN_NB + j < P.D) ? W[(size_t)(k0 + k) * P.LD + J * DN_NB + j] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int t0 = 0; t0 < DN_NB; t0 += 4) {
      const double av = a[t0 + kk][16 * w + m];          // A[i][k] = c_k[i]
#pragma unroll
      for (int cb = 0; cb < 4; cb++) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[t0 + kk][16 * cb + m], acc[cb], 0, 0, 0);
    }
  }
  const double nn = (double)n, f = (nn / (nn + 5.0)) / (nn - 1.0), reg = 1e-3 * (5.0 / (nn + 5.0));
#pragma unroll
  for (int cb = 0; cb < 4; cb++)
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int i = I * DN_NB + 16 * w + kk + 4 * v, j = J * DN_NB + 16 * cb + m;
      if (i < P.D && j < P.D && (I != J || j <= i)) {
        double val = f * acc[cb][v] + (i == j ? reg : 0.0);
        if (P.f32) val = (double)(float)val;     // the rounded matrix is the metric: the factor below is ITS factor
        A[(size_t)i * P.LD + j] = val;           // lower half: what the factorisation starts from (and overwrites with L)
        if (P.f32) { if (i != j) dn_f32_row(A, P.LD, j)[i] = (float)val; }   // upper half as floats in the row's free half
        else A[(size_t)j * P.LD + i] = val;      // upper half: stays, it IS the metric
        if (i == j) dg[i] = val;
      }
    }
}

// Blocked right-looking Cholesky of M^-1 -> lower factor in Lc (row-major; only the lower triangle is meaningful).
// step kb: k_dn_potrf (diagonal block), k_dn_trsm (the rows below it), k_dn_syrk (trailing update).
__global__ __launch_bounds__(256) void k_dn_potrf(const DnParams P, int kb) {
  __shared__ double a[DN_NB][DN_NB + 1];
  const int chain = blockIdx.x, tid = threadIdx.x, r0 = kb * DN_NB, nb = min(DN_NB, P.D - r0);
  double *L = dn_fac(P, chain);
  for (int e = tid; e < DN_NB * DN_NB; e += 256) { const int i = e >> 6, j = e & 63; a[i][j] = (i < nb && j <= i) ? L[(size_t)(r0 + i) * P.LD + r0 + j] : 0.0; }
  __syncthreads();
  for (int j = 0; j < nb; j++) {
    if (tid == 0) {
      const double d = a[j][j];
      if (!(d > 0.0)) { *P.fail = 1; a[j][j] = 1.0; } else a[j][j] = sqrt(d);
    }
    __syncthreads();
    const double djj = a[j][j];
    for (int i = j + 1 + tid; i < nb; i += 256) a[i][j] /= djj;
    __syncthreads();
    for (int e = tid; e < (nb - j - 1) * (nb - j - 1); e += 256) {
      const int i = j + 1 + e / (nb - j - 1), k = j + 1 + e % (nb - j - 1);
      if (k <= i) a[i][k] -= a[i][j] * a[k][j];
    }
    __syncthreads();
  }
  for (int e = tid; e < DN_NB * DN_NB; e += 256) { const int i = e >> 6, j = e & 63; if (i < nb && j <= i) L[(size_t)(r0 + i) * P.LD + r0 + j] = a[i][j]; }   // the upper half is M^-1's
}
// rows below the diagonal block: X Lkk' = A  ->  forward substitution per row (64 rows per workgroup, staged in LDS)
__global__ __launch_bounds__(256) void k_dn_trsm(const DnParams P, int kb) {
  __shared__ double lk[DN_NB][DN_NB + 1], x[DN_NB][DN_NB + 1];
  const int chain = blockIdx.y, tid = threadIdx.x, c0 = kb * DN_NB, nb = min(DN_NB, P.D - c0);
  const int r0 = (kb + 1) * DN_NB + blockIdx.x * DN_NB;
  if (r0 >= P.D) return;
  double *L = dn_fac(P, chain);
  for (int e = tid; e < DN_NB * DN_NB; e += 256) {
    const int i = e >> 6, j = e & 63;
    lk[i][j] = (i < nb && j <= i) ? L[(size_t)(c0 + i) * P.LD + c0 + j] : 0.0;
    x[i][j] = (r0 + i < P.D && j < nb) ? L[(size_t)(r0 + i) * P.LD + c0 + j] : 0.0;
  }
  __syncthreads();
  if (tid < DN_NB) {
    const int r = tid;
    for (int j = 0; j < nb; j++) {
      double s = x[r][j];
      for (int t = 0; t < j; t++) s -= x[r][t] * lk[j][t];
      x[r][j] = s / lk[j][j];
    }
  }
  __syncthreads();
  for (int e = tid; e < DN_NB * DN_NB; e += 256) { const int i = e >> 6, j = e & 63; if (r0 + i < P.D && j < nb) L[(size_t)(r0 + i) * P.LD + c0 + j] = x[i][j]; }
}
// trailing update A[I][J] -= L[I][kb] L[J][kb]' for the lower tiles I >= J > kb (fp64 MFMA, see k_dn_cov)
// (jmax: the last block column this launch updates -- the factorisation runs in panels of DN_PANEL block columns: inside a panel
//  the update after each block column stops at the panel's edge, the rest of the matrix gets the whole panel at once, k_dn_syrk_wide)
__global__ __launch_bounds__(256) void k_dn_syrk(const DnParams P, int kb, int jmax) {
  __shared__ double a[DN_NB][DN_NB + 1], b[DN_NB][DN_NB + 1];
  const int I = kb + 1 + blockIdx.x, J = kb + 1 + blockIdx.y, chain = blockIdx.z;
  if (J > I || J > jmax) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 15, kk = lane >> 4, c0 = kb * DN_NB;
  double *L = dn_fac(P, chain);
  for (int e = tid; e < DN_NB * DN_NB; e += 256) {
    const int i = e >> 6, t = e & 63;
    a[i][t] = I * DN_NB + i < P.D ? L[(size_t)(I * DN_NB + i) * P.LD + c0 + t] : 0.0;
    b[i][t] = J * DN_NB + i < P.D ? L[(size_t)(J * DN_NB + i) * P.LD + c0 + t] : 0.0;
  }
  __syncthreads();
  dn_d4 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; cb++) acc[cb] = dn_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int t0 = 0; t0 < DN_NB; t0 += 4) {
    const double av = a[16 * w + m][t0 + kk];            // A[i][t] = L[I rows][kb cols]
#pragma unroll
    for (int cb = 0; cb < 4; cb++) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[16 * cb + m][t0 + kk], acc[cb], 0, 0, 0);   // B[t][j] = L[J rows][t]
  }
#pragma unroll
  for (int cb = 0; cb < 4; cb++)
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int i = I * DN_NB + 16 * w + kk + 4 * v, j = J * DN_NB + 16 * cb + m;
      if (i < P.D && j < P.D && j <= i) L[(size_t)i * P.LD + j] -= acc[cb][v];
    }
}

// ---------------------------------------------------------------- checks of the factor (potus_dense_check: test hooks)
// y = L' x and z = L y out of the lower triangle, written for clarity, not speed: with M^-1 x from the sampler's own matrix pass
// they tell whether L L' is the metric the leapfrog uses, at sizes no host can factor in a test's time.
__global__ __launch_bounds__(256) void k_dn_chk_ltx(const DnParams P, int chain, int xslot, int yslot) {
  const double *L = dn_fac(P, chain), *x = dn_vec(P, chain, xslot);
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= P.D) return;
  double s = 0.0;
  for (int i = j; i < P.D; i++) s += L[(size_t)i * P.LD + j] * x[i];
  dn_vec(P, chain, yslot)[j] = s;
}
__global__ __launch_bounds__(256) void k_dn_chk_lx(const DnParams P, int chain, int yslot, int zslot) {
  __shared__ double red[256];
  const double *L = dn_fac(P, chain), *y = dn_vec(P, chain, yslot);
  const int i = blockIdx.x;
  double s = 0.0;
  for (int j = threadIdx.x; j <= i; j += 256) s += L[(size_t)i * P.LD + j] * y[j];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) { if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h]; __syncthreads(); }
  if (threadIdx.x == 0) dn_vec(P, chain, zslot)[i] = red[0];
}

// Trailing update behind a PANEL of nk block columns (c0 = 64 pb .. 64 (pb + nk)): A[I][J] -= L[I][panel] L[J][panel]' for every
// 128 x 128 tile I >= J behind the panel.  With one block column at a time the update reads and rewrites the whole trailing
// matrix per 64 columns factored (4 flops per byte moved: 10-11 TFLOP/s, bound by that read-modify-write); with the panel's 256
// columns per pass and 128 x 128 tiles it is 11 flops per byte.  Four waves, one 64 x 64 quadrant each = 4 x 4 MFMA tiles
// (v_mfma_f64_16x16x4_f64, operand layout as in k_dn_cov), the panel's columns staged through LDS sixteen at a time, stored
// [k][row] with a row stride of 16 mod 32 doubles so that the four k-rows a wave reads per step fall on disjoint banks.
#define DN_PANEL 4
#define DN_WT 128
#define DN_WS 144
__global__ __launch_bounds__(256, 2) void k_dn_syrk_wide(const DnParams P, int pb, int nk) {   // (two workgroups per compute unit: one tile's read-modify-write tail under the other's products)
  __shared__ double a[16][DN_WS], b[16][DN_WS];
  const int r_first = (pb + nk) * DN_NB;                 // first row / column behind the panel
  const int I0 = r_first + (int)blockIdx.x * DN_WT, J0 = r_first + (int)blockIdx.y * DN_WT, chain = blockIdx.z;
  if (J0 > I0 || I0 >= P.D) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, n = lane & 15, wr = w >> 1, wc = w & 1;
  const int c0 = pb * DN_NB, K = min(nk * DN_NB, P.D - c0);
  double *L = dn_fac(P, chain);
  dn_d4 acc[4][4];
#pragma unroll
  for (int ti = 0; ti < 4; ti++)
#pragma unroll
    for (int tj = 0; tj < 4; tj++) acc[ti][tj] = dn_d4{0.0, 0.0, 0.0, 0.0};
  const int lrow = tid >> 1, lk = (tid & 1) * 8;         // staging: thread = (row of the tile, half of the sixteen columns)
  const bool oka = I0 + lrow < P.D, okb = J0 + lrow < P.D;
  const double *ga = L + (size_t)(oka ? I0 + lrow : 0) * P.LD + c0 + lk, *gb = L + (size_t)(okb ? J0 + lrow : 0) * P.LD + c0 + lk;
  double ra[8], rb[8];                                   // the next sixteen columns travel in registers while the matrix cores work
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const bool ok = k0 + lk + u < K;
      ra[u] = (ok && oka) ? ga[k0 + u] : 0.0;
      rb[u] = (ok && okb) ? gb[k0 + u] : 0.0;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += 16) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; u++) { a[lk + u][lrow] = ra[u]; b[lk + u][lrow] = rb[u]; }
    __syncthreads();
    if (k0 + 16 < K) fetch(k0 + 16);
#pragma unroll
    for (int st = 0; st < 4; st++) {
      double av[4], bv[4];
#pragma unroll
      for (int t = 0; t < 4; t++) { av[t] = a[4 * st + g][64 * wr + 16 * t + n]; bv[t] = b[4 * st + g][64 * wc + 16 * t + n]; }
#pragma unroll
      for (int ti = 0; ti < 4; ti++)
#pragma unroll
        for (int tj = 0; tj < 4; tj++) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti], bv[tj], acc[ti][tj], 0, 0, 0);
    }
  }
#pragma unroll
  for (int ti = 0; ti < 4; ti++)
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int i = I0 + 64 * wr + 16 * ti + g + 4 * v, j = J0 + 64 * wc + 16 * tj + n;
        if (i < P.D && j < P.D && j <= i) L[(size_t)i * P.LD + j] -= acc[ti][tj][v];
      }
}

// unit metric: M^-1 = L = I (the matrix buffer is zero apart from the diagonal)
__global__ void k_dn_identity(const DnParams P) {
  const int chain = blockIdx.y;
  double *A = dn_mat(P, chain), *dg = dn_diag(P, chain), *Lf = dn_fac(P, chain);   // (pooled: launched for one "chain")
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.D; i += gridDim.x * blockDim.x) {
    A[(size_t)i * P.LD + i] = 1.0; Lf[(size_t)i * P.LD + i] = 1.0; dg[i] = 1.0;
    if (P.pooled && P.f32) P.A32[(size_t)i * P.LD + i] = 1.0f;
  }
}
// fills the upper triangle and the diagonal vector with a symmetric positive definite test matrix on the device (rates
// at sizes that would take seconds to upload): a_ij = exp(-|i-j|/50) * (1 + 0.1 c) + (i == j ? 1 : 0)
__global__ void k_dn_fill(const DnParams P) {
  const size_t n = (size_t)P.D * P.D;
  for (int c = 0; c < P.chains; c++)
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
      const int i = (int)(e / P.D), j = (int)(e % P.D);
      const int d = i > j ? i - j : j - i;
      const double v = exp(-(double)d / 50.0) * (1.0 + 0.1 * c) + (i == j ? 1.0 : 0.0);
      if (j > i) {
        if (P.f32) dn_f32_row(P.A + (size_t)c * P.D * P.LD, P.LD, i)[j] = (float)v;   // (development: POTUS_PROBE_F32)
        else P.A[(size_t)c * P.D * P.LD + (size_t)i * P.LD + j] = v;
      }
      if (j == i) { P.A[(size_t)c * P.D * P.LD + (size_t)i * P.LD + j] = 1.0; P.dg[(size_t)c * P.LD + i] = v; }
    }
}

#include "potus_dense_pool.hpp"
