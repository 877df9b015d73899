// potus_dense_pool.hpp -- the dense metric POOLED over the chains of a handle (potus_opts.pooled_metric; round 6).
//
// stan::mcmc::adapt_dense_e_nuts gives every chain its own covariance estimate (SURVEY 8 f4 lists the pooled warm-up as an extra; CmdStan
// runs one process per chain and cannot pool).  With one D x D matrix per chain the matrix pass of a leaf round streams one triangle per
// ACTIVE chain -- sixteen private 6.9 GB triangles at the configs[4] shape, 12 ms per round -- and sixteen matrices are 222 GB.  Pooled:
// at every window end the draws of ALL chains of the handle form ONE regularised covariance (covar_adaptation::learn_covariance applied to
// the pooled sample), ONE Cholesky factor; a leaf round then is
//
//     Y[D x R] = M^-1 [D x D]  X[D x R],     R = (active chains) x (right-hand sides per chain) <= 48,
//
// i.e. the matrix streamed ONCE per round whatever the number of chains, the products on the fp64 matrix cores.  A declared deviation
// from Stan (like rhat_stop and metric_storage): the sampler is exact for the metric it uses -- one factor draws the momenta, the same
// matrix multiplies them -- but the metric is estimated from chains x n draws instead of n.
//
// Storage: M^-1 as a FULL symmetric matrix (both triangles and the diagonal; D x LD doubles), its factor in a buffer of its own (memory
// is no longer the limit: 2 x 13.85 GB at D = 41 610 instead of 16 x 13.85).  The full matrix lets the product load every operand block
// straight into the lanes the matrix instruction wants, coalesced, without a transpose:
//     v_mfma_f64_16x16x4_f64   D[m][n] += sum_k A[m][k] B[k][n];   lane l feeds A[l & 15][l >> 4] and B[l >> 4][l & 15], holds D[(l >> 4) + 4 v][l & 15]
//     here  m = a COLUMN of M^-1, k = a ROW, n = a right-hand side:  D[col][r] += sum_row M^-1[row][col] x_r[row]  = (M^-T X)[col][r] = (M^-1 X)[col][r]
// so lane l loads M^-1[row0 + (l >> 4)][col(l & 15)]: sixteen consecutive columns of four rows = four 128-byte segments per instruction.
// Bytes per round 8 D^2 (twice the triangle), flops 2 D^2 R: at R = 32 and D = 41 610 2.3 ms of HBM at 6 TB/s against 1.4 ms of the 78.6
// TFLOP/s fp64 matrix peak -- HBM-bound; the pass over the triangle alone would be bound by the matrix cores (1.15 ms of HBM).
#pragma once

#define DNP_THREADS 512
#define DNP_COLS 256                      // columns of M^-1 per workgroup: 32 per wave, as two interleaved 16-column operand tiles (even / odd columns)
#define DNP_KB 32                         // rows per batch: eight 16-byte loads per lane, TWO batches in flight (the next one is requested before this one is multiplied)
#define DNP_XS 34                         // LDS row stride of a staged right-hand side (doubles): conflict-free for the B-operand reads of a half wave
#define DNP_RMAX 48                       // right-hand sides per launch (three operand tiles of sixteen)
#define DNP_SPLIT_MAX 8
// waves per SIMD the product is compiled for (hipcc: the second launch bound is waves per execution unit): four, i.e. 128 registers per lane and two workgroups
// per compute unit, wherever the variant fits them (DNP_WAVES_PER_SIMD_OF below; fp32 storage with three operand tiles needs 165).  Squeezing a variant that does not
// fit costs more than it gives: three tiles of fp64 at 32-row batches needed 158 registers and lost a third of their rate in 128 (70 spills) -- profiles/r06_dense_pooled.txt
#define DNP_LDS(NT) ((size_t)2 * (NT) * 16 * DNP_XS * 8)
// potus_opts.metric_storage = f32 on top of pooled_metric: the ONE matrix is kept rounded to fp32 as well (DnParams::A32; the rounded matrix IS the metric, its
// factor is the factor of the rounded values -- the argument of dn_f32_row), the pass streams 4 D^2 bytes.  A 16-byte load then brings FOUR columns: a wave
// takes 64 columns as four interleaved operand tiles, a workgroup 512; the arithmetic stays fp64 (v_cvt_f64_f32 on the way into the matrix instruction).
#define DNP_CT(F32) ((F32) ? 4 : 2)                     // operand tiles (= columns per 16-byte load) per wave
#define DNP_COLS_OF(F32) (128 * DNP_CT(F32))            // columns per workgroup
// Rows per batch of a variant.  One operand tile of right-hand sides: 32 (eight 16-byte loads per lane and batch).  Two or three tiles: 16 -- half the registers of the two
// batches in flight, which lets TWO workgroups share a compute unit (fp64 storage, three tiles: 113 registers instead of 158; fp32 storage, two tiles: 127 instead of 174),
// and the matrix pipe no longer idles while the one resident workgroup sits at its barrier: 2.03 -> 1.81 ms at 32 right-hand sides with fp32 storage (61.4 TFLOP/s = 0.78
// of the matrix peak), 3.38 -> 3.18 at 48 with fp64 (profiles/r06_dense_pooled.txt, section 6).  The order of the sums does not depend on it (steps of four rows, in row order).
#define DNP_KB_OF(NT, F32) ((NT) >= 2 ? 16 : DNP_KB)
#define DNP_WAVES_PER_SIMD_OF(NT, F32) ((F32) ? ((NT) <= 2 ? 4 : 2) : 4)

// the right-hand sides of a pooled launch: column r = (chain, job)
struct DnPoolRhs {
  int n;                                  // right-hand sides (<= DNP_RMAX)
  int nrhs;                               // jobs per chain in this launch
  int job0;
  unsigned char chain[DNP_RMAX];
};

// Y_part[split][r][col] = sum over the split's rows of M^-1[row][col] x_r[row].  grid (column panels, row splits); NT = operand tiles of sixteen right-hand sides.
// Software pipeline over batches of DNP_KB rows, two register sets and two LDS buffers: while batch b is multiplied, the matrix block and the right-hand sides of
// batch b + 1 are on their way (128 registers at NT = 2: two workgroups per compute unit, i.e. four waves per SIMD to hide what is left).  First version of the
// round (one batch of 64 rows in flight, nothing requested ahead): 4.6 TB/s with 32 right-hand sides at D = 41 610, the matrix pipe 47 % busy.
template <int NT, bool F32 = false, int KB = DNP_KB_OF(NT, F32)>
__global__ __launch_bounds__(DNP_THREADS, DNP_WAVES_PER_SIMD_OF(NT, F32)) void k_dn_pool_mm(const DnParams P, const DnPoolRhs R, int rows_per_split) {
  constexpr int XS = KB + 2, NS = (16 * NT * KB + DNP_THREADS - 1) / DNP_THREADS;   // LDS row stride of a staged right-hand side; staging slots per thread
  constexpr int CT = DNP_CT(F32), EB = F32 ? 4 : 8;                  // operand tiles per wave; bytes per stored element
  extern __shared__ __attribute__((aligned(16))) double dnp_lds[];   // [2][NT * 16][XS]
  typedef double d4_t __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = P.D, LD = P.LD;
  const int c0 = (int)blockIdx.x * DNP_COLS_OF(F32) + 16 * CT * w;    // the wave's 32 (fp32 storage: 64) columns
  const int rbeg = (int)blockIdx.y * rows_per_split, rend = min(D, rbeg + rows_per_split);
  const int mcol = lane & 15, krow = lane >> 4;
  // the lane's two columns (operand tile 0: even, tile 1: odd) as one 16-byte load; columns at or beyond D read nothing
  const int colA = c0 + CT * mcol;
  const unsigned rowbytes = uni32((unsigned)EB * (unsigned)LD);
  // (the lane's row of a four-row step travels in the vector offset -- a scalar offset must be wave-uniform -- and rows beyond the split's end are
  //  masked lane by lane: the bounds check of a raw buffer does not look at the scalar offset)
  const unsigned voffA = colA + CT - 1 < LD && colA < D ? (unsigned)EB * (unsigned)colA + (unsigned)krow * rowbytes : PT_OOB;   // (LD is a multiple of 8 and the padding columns hold zeros)
  d4_t acc[CT][NT];
#pragma unroll
  for (int t = 0; t < CT; t++)
#pragma unroll
    for (int n = 0; n < NT; n++) acc[t][n] = d4_t{0.0, 0.0, 0.0, 0.0};
  // staging of the right-hand sides: thread -> (row tid & 31 of the batch, right-hand sides (tid >> 5) + 16 i); their vectors are looked up once
  const int srow = tid & (KB - 1), sr = tid / KB;
  const double *xp[NS];
#pragma unroll
  for (int i = 0; i < NS; i++) {
    const int r = sr + (DNP_THREADS / KB) * i;
    xp[i] = nullptr;
    if (r < R.n) {
      const int chain = R.chain[r], job = R.job0 + r % R.nrhs;
      const DnRound &rd = P.rd[chain];
      if (rd.active && rd.job[job].x >= 0) xp[i] = dn_vec(P, chain, rd.job[job].x);   // (the host's list of active chains may be a few rounds old)
    }
  }
  auto request = [&](u32x4 (&a)[KB / 4], int r0) {    // the batch's rows are one resource (none left: size 0, zeros come back); the step of four rows is the scalar offset
    const int nrows = max(0, min(KB, rend - r0));
    const rsrc_t rsA = F32 ? make_rsrc(uni_ptr(P.A32 + (size_t)min(r0, D - 1) * LD), uni32((unsigned)nrows * rowbytes))
                           : make_rsrc(uni_ptr(P.A + (size_t)min(r0, D - 1) * LD), uni32((unsigned)nrows * rowbytes));
#pragma unroll
    for (int k = 0; k < KB / 4; k++)
      a[k] = __builtin_amdgcn_raw_buffer_load_b128(rsA, 4 * k + krow < nrows ? voffA : PT_OOB, (unsigned)(4 * k) * rowbytes, 2 /* nt: read once per round */);
  };
  auto xload = [&](double (&xv)[NS], int r0) {
#pragma unroll
    for (int i = 0; i < NS; i++) xv[i] = (xp[i] && r0 + srow < rend) ? xp[i][r0 + srow] : 0.0;
  };
  auto xstore = [&](double *xs, const double (&xv)[NS]) {
#pragma unroll
    for (int i = 0; i < NS; i++) { const int r = sr + (DNP_THREADS / KB) * i; if (r < 16 * NT) xs[r * XS + srow] = xv[i]; }
  };
  auto multiply = [&](const u32x4 (&a)[KB / 4], const double *xs) {
#pragma unroll
    for (int k = 0; k < KB / 4; k++) {
      double av[CT];
      if constexpr (F32) {
#pragma unroll
        for (int t = 0; t < CT; t++) av[t] = (double)__uint_as_float((unsigned)a[k][t]);
      } else {
        av[0] = __hiloint2double((int)a[k][1], (int)a[k][0]); av[1] = __hiloint2double((int)a[k][3], (int)a[k][2]);
      }
      double bv[NT];
#pragma unroll
      for (int n = 0; n < NT; n++) bv[n] = xs[(16 * n + mcol) * XS + 4 * k + krow];
#pragma unroll
      for (int n = 0; n < NT; n++)
#pragma unroll
        for (int t = 0; t < CT; t++) acc[t][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[n], acc[t][n], 0, 0, 0);
    }
  };
  double *xs0 = dnp_lds, *xs1 = dnp_lds + (size_t)NT * 16 * XS;
  u32x4 a0[KB / 4], a1[KB / 4];
  double xv[NS];
  request(a0, rbeg);
  xload(xv, rbeg);
  xstore(xs0, xv);
  __syncthreads();
  for (int r0 = rbeg; r0 < rend; r0 += 2 * KB) {
    request(a1, r0 + KB);
    xload(xv, r0 + KB);
    multiply(a0, xs0);
    xstore(xs1, xv);
    __syncthreads();
    request(a0, r0 + 2 * KB);
    xload(xv, r0 + 2 * KB);
    multiply(a1, xs1);                                                  // (past the split's end: zeros times zeros)
    xstore(xs0, xv);
    __syncthreads();
  }
  // D[(l >> 4) + 4 v][l & 15]: column c0 + CT ((l >> 4) + 4 v) + tile, right-hand side 16 n + (l & 15)
  double *yp = P.ypool + (size_t)blockIdx.y * DNP_RMAX * (size_t)LD;
#pragma unroll
  for (int t = 0; t < CT; t++)
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int col = c0 + CT * (krow + 4 * v) + t, r = 16 * n + mcol;
        if (col < D && r < R.n) yp[(size_t)r * LD + col] = acc[t][n][v];
      }
}

// y = the row splits in order; then what the round wants done with the product (as k_dn_symv_finish): stores, the position update, partial sums
// of dot . y in blocks of 64 elements (the layout dn_partial_sum adds up).  grid (blocks of 64 elements, right-hand sides), 64 threads.
__global__ __launch_bounds__(64) void k_dn_pool_finish(const DnParams P, const DnPoolRhs R, int nsplit) {
  const int r = blockIdx.y, chain = R.chain[r], job = R.job0 + r % R.nrhs;
  if (threadIdx.x == 0 && blockIdx.x == 0 && r == 0 && P.act_passes && P.count_passes) atomicAdd(P.act_passes, 1ull);   // one pass over ONE matrix, whoever took part
  const DnRound &rd = P.rd[chain];
  if (!rd.active) return;
  const DnJob &jb = rd.job[job];
  const int i = blockIdx.x * 64 + threadIdx.x;
  double dsum = 0.0;
  if (i < P.D) {
    double y = 0.0;
    for (int s = 0; s < nsplit; s++) y += P.ypool[((size_t)s * DNP_RMAX + r) * (size_t)P.LD + i];
    if (jb.y >= 0) dn_vec(P, chain, jb.y)[i] = y;
    if (jb.qout >= 0) dn_vec(P, chain, jb.qout)[i] = dn_vec(P, chain, jb.qin)[i] + jb.coef * y;
    if (job == 0 && jb.dot >= 0) dsum = dn_vec(P, chain, jb.dot)[i] * y;
  }
  const double tot = dpp_wave_sum(dsum);
  if (threadIdx.x == 0 && job == 0) P.partial[(size_t)chain * P.npart + blockIdx.x] = tot;
}

// Pooled covar_adaptation at a window's end, in three steps so that a host can pool further -- over the handles of a process and, through an all-reduce, over
// the GPUs of a node -- between the second and the third (pooled_metric = 2: potus_dense_pool_window / potus_dense_pool_finish):
//   1. the mean over the draws of every chain of the handle (welford_covar_estimator's recurrence over the chains one after the other) -> P.pmean, the
//      draws centred in place;
//   2. M2 = sum over chains and draws of c c' (tiles of 64 x 64 on the matrix cores as k_dn_cov, lower tiles computed and mirrored: exactly symmetric) -> P.A,
//      unscaled;   [host: M2 += n_loc (mean_loc - mean)(mean_loc - mean)' and the sums over handles / ranks: Chan's pairwise update]
//   3. M^-1 = N/(N+5) * M2 / (N-1) + 1e-3 * 5/(N+5) * I with N the number of pooled draws (learn_covariance on the pooled sample): both triangles and the
//      diagonal to the matrix, the lower triangle also to the factor's buffer, where the factorisation starts from it.
__global__ __launch_bounds__(256) void k_dn_pool_center(const DnParams P, int n) {
  for (int j = blockIdx.x * 256 + threadIdx.x; j < P.D; j += gridDim.x * 256) {
    double m = 0.0, cnt = 0.0;
    for (int c = 0; c < P.chains; c++) {
      const double *W = P.win + (size_t)c * P.win_cap * (size_t)P.LD;
      for (int k = 0; k < n; k++) { cnt += 1.0; m += (W[(size_t)k * P.LD + j] - m) / cnt; }
    }
    for (int c = 0; c < P.chains; c++) {
      double *W = P.win + (size_t)c * P.win_cap * (size_t)P.LD;
      for (int k = 0; k < n; k++) W[(size_t)k * P.LD + j] -= m;
    }
    P.pmean[j] = m;
  }
}
__global__ __launch_bounds__(256) void k_dn_pool_cov(const DnParams P, int n) {
  __shared__ double a[DN_NB][DN_NB + 1], b[DN_NB][DN_NB + 1];
  const int I = blockIdx.x, J = blockIdx.y;
  if (J > I) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 15, kk = lane >> 4;
  dn_d4 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; cb++) acc[cb] = dn_d4{0.0, 0.0, 0.0, 0.0};
  for (int c = 0; c < P.chains; c++) {
    const double *W = P.win + (size_t)c * P.win_cap * (size_t)P.LD;
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
        const double av = a[t0 + kk][16 * w + m];
#pragma unroll
        for (int cb = 0; cb < 4; cb++) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[t0 + kk][16 * cb + m], acc[cb], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int cb = 0; cb < 4; cb++)
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int i = I * DN_NB + 16 * w + kk + 4 * v, j = J * DN_NB + 16 * cb + m;
      if (i < P.D && j < P.D && (I != J || j <= i)) { P.A[(size_t)i * P.LD + j] = acc[cb][v]; P.A[(size_t)j * P.LD + i] = acc[cb][v]; }
    }
}
__global__ __launch_bounds__(256) void k_dn_pool_scale(const DnParams P, double nn) {
  const double f = (nn / (nn + 5.0)) / (nn - 1.0), reg = 1e-3 * (5.0 / (nn + 5.0));
  const int i = blockIdx.y;
  for (int j = blockIdx.x * 256 + threadIdx.x; j <= i; j += gridDim.x * 256) {
    double val = f * P.A[(size_t)i * P.LD + j] + (i == j ? reg : 0.0);
    if (P.f32) {                                   // the rounded matrix is the metric: the factor is ITS factor
      val = (double)(float)val;
      P.A32[(size_t)i * P.LD + j] = (float)val; P.A32[(size_t)j * P.LD + i] = (float)val;
    }
    P.A[(size_t)i * P.LD + j] = val;
    P.A[(size_t)j * P.LD + i] = val;
    P.Lf[(size_t)i * P.LD + j] = val;
    if (i == j) P.dg[i] = val;
  }
}
// metric_storage = f32 for a matrix that arrived in fp64 (development: the probes' uploads): A rounded in place, the floats beside it
__global__ void k_dn_pool_round32(const DnParams P) {
  const size_t n = (size_t)P.D * P.LD;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float v = (float)P.A[e];
    P.A32[e] = v; P.A[e] = (double)v;
    if (e / P.LD == e % P.LD) P.dg[e / P.LD] = (double)v;
  }
}
// a symmetric positive definite test matrix generated on the device, as k_dn_fill's of chain 0: a_ij = exp(-|i-j|/50) + (i == j ? 1 : 0), FULL storage
__global__ void k_dn_pool_fill(const DnParams P) {
  const size_t n = (size_t)P.D * P.D;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / P.D), j = (int)(e % P.D);
    const int d = i > j ? i - j : j - i;
    const double v = exp(-(double)d / 50.0) + (i == j ? 1.0 : 0.0);
    P.A[(size_t)i * P.LD + j] = P.f32 ? (double)(float)v : v;
    if (P.f32) P.A32[(size_t)i * P.LD + j] = (float)v;
    if (i == j) P.dg[i] = P.f32 ? (double)(float)v : v;
  }
}
