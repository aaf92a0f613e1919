// Fit-side kernels (gfx950): scaling, kernel-matrix assembly, blocked Cholesky, triangular inverse,
// alpha solve and MFMA-fragment packing of W = L^-1.
//
// What they replace (SK = sklearn, reference paths relative to /root/reference):
//   prescale      X / length_scale                         SK/gaussian_process/kernels.py:1711,1556
//   kmat          Matern(nu=2.5)/RBF __call__(X) + alpha*I  kernels.py:1711-1738, :1556-1565; _gpr.py:346-347
//   gemm          the products of cholesky(K, lower=True) -> LAPACK dpotrf (_gpr.py:349; the chain itself: chol_kernels.hip)
//   trtri         W = L^-1: turns the per-candidate solve_triangular (_gpr.py:454-456) into a GEMM
//   trmv          alpha = cho_solve((L, True), y)           _gpr.py:360-364, as W^T (W y)
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "gpbo_internal.h"
#include "gemm_tile.h"
#include "fit_bodies.h"

namespace gpbo {

// ------------------------------------------------------------------------------------------------
// X / length_scale into a zero-padded [n_pad][DP] image (true division, as numpy does): prescale_elem (fit_bodies.h).
__global__ void prescale_kernel(const double* __restrict__ X, int64_t n, int d, int DP,
                                const double* __restrict__ ls, double* __restrict__ out,
                                int64_t total, int64_t lane_stride) {
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  ls += (int64_t)blockIdx.y * lane_stride;     // lane mode: X is shared, length scales and output are per lane
  out += (int64_t)blockIdx.y * lane_stride;
  prescale_elem(X, n, d, DP, ls, out, idx);
}

int launch_prescale(gpbo_ctx* ctx, const double* X, int64_t n, int d, int DP, const double* ls,
                    double* out, int64_t n_pad) {
  int64_t total = n_pad * DP;
  if (total == 0) return GPBO_OK;
  int64_t blocks = (total + 255) / 256;
  prescale_kernel<<<dim3((unsigned)blocks, (unsigned)ctx->lanes), dim3(256), 0, ctx->stream>>>(X, n, d, DP, ls, out, total,
                                                                                             ctx->lane_stride);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// ------------------------------------------------------------------------------------------------
// (kernel_value and the tile body: fit_bodies.h)
// K (lower block triangle, 64x64 tiles): one workgroup per LOWER tile (linear block id -> (bi, bj), no idle
// workgroups), 4x4 outputs per thread, the two point tiles staged k-major in LDS.  HBM-write bound: N^2/2 * 8 B.
// `out` is K, or directly the buffer the Cholesky factorises in place (no K -> L copy on the fit path).
// Not an HBM kernel despite what it writes: per element 2 DP + ~45 fp64 VALU operations (differences, sqrt, exp,
// polynomial) = 16.5 us of VALU issue at N = 4096, d = 16 (38 us measured, 1.8 TB/s) and 93 us at N = 8192, d = 32 (160
// measured) — at d = 32 the arithmetic alone caps the store rate at 2.9 TB/s = 0.46 of a copy's 6.29.  Round 3 measured a
// row-walking variant (lane = column with its point in registers, the row's coordinates as SGPR operands through
// s_load_dwordx16, one full 512-byte row segment per store, no LDS): the same bits, 44 / 158 us — no faster, 2x slower
// at N <= 1024 (each row is a fresh scalar-cache miss) — and dropped it (scripts/archive/r03_kmat_probe.py, profiles/r03_kmat_probe.json).
template <int KERNEL>
__global__ __launch_bounds__(256) void kmat_kernel(const double* __restrict__ Xs, int DP, int64_t N,
                                                   int64_t NP, double noise, double* __restrict__ K,
                                                   int64_t lane_stride) {
  // blockIdx.x = bi (bi + 1) / 2 + bj, bj <= bi
  int bi, bj;
  lower_tile_of((int)blockIdx.x, bi, bj);
  Xs += (int64_t)blockIdx.z * lane_stride;
  K += (int64_t)blockIdx.z * lane_stride;
  extern __shared__ __attribute__((aligned(16))) double kmat_smem[];
  kmat_tile_body<KERNEL>(Xs, DP, N, NP, noise, K, bi, bj, kmat_smem, (int)threadIdx.x);
}

int launch_kmat(gpbo_ctx* ctx, Model& m, double noise, double* out) {
  const int64_t nt = m.NP / 64;
  dim3 grid((unsigned)(nt * (nt + 1) / 2), 1, (unsigned)ctx->lanes);
  const size_t lds = (size_t)2 * m.DP * 64 * sizeof(double);
  if (m.kernel == GPBO_KERNEL_MATERN25)
    kmat_kernel<GPBO_KERNEL_MATERN25><<<grid, dim3(256), lds, ctx->stream>>>(m.Xs, m.DP, m.N, m.NP, noise, out, ctx->lane_stride);
  else
    kmat_kernel<GPBO_KERNEL_RBF><<<grid, dim3(256), lds, ctx->stream>>>(m.Xs, m.DP, m.N, m.NP, noise, out, ctx->lane_stride);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// ------------------------------------------------------------------------------------------------
// (The 64-column Cholesky steps of rounds 1-2 — potrf_diag_kernel / chol_step_kernel — lived here; the factorisation is
// chol_kernels.hip's 128-column schedule since round 3 and the old kernels were retired in round 4.)
// Output tile of a workgroup of the 64x64-tile kernels.  Longest k-ranges first (workgroups are handed out in block-id order, x
// fastest).  A lower triangular: row tile bm multiplies bm + 1 k-tiles -> rows from the bottom up.  B lower triangular: column
// tile bn starts at k-tile bn -> the grid is transposed (x = row tile), so all tiles of column 0 go first, then column 1, ...
__device__ __forceinline__ void gemm64_tile_of_block(const GemmArgs& g, int& bm, int& bn) {
  bm = (int)blockIdx.y; bn = (int)blockIdx.x;
  if (g.a_lower) bm = (int)(gridDim.y - 1 - blockIdx.y);
  if (g.b_lower) { bm = (int)blockIdx.x; bn = (int)blockIdx.y; }
  if (g.tri_grid) lower_tile_of((int)blockIdx.x, bm, bn);      // live tiles only, rows from the top (W^T W: longest k first)
}

template <bool BT, bool AT>
__global__ __launch_bounds__(256) void gemm_f64_kernel(GemmArgs g) {
  const int zl = blockIdx.z / g.batch, bz = blockIdx.z - zl * g.batch;   // lane mode: z = lane * batch + b
  int bm, bn;
  gemm64_tile_of_block(g, bm, bn);
  __shared__ __attribute__((aligned(16))) double gt_lds[GT_LDS_DOUBLES];
  gemm_tile_body<BT, AT>(g, bm, bn, zl, bz, gt_lds);
}

// ------------------------------------------------------------------------------------------------
// The same 64x64 tile by SIXTEEN waves (round 6): four 256-thread groups of one 1024-thread workgroup each run the tile body
// over a QUARTER of the tile's k-range, park their partial tiles in LDS (each in its own staging area, free by then) and all
// 1024 threads add them in a fixed order — ((q0 + q1) + (q2 + q3)), then alpha, beta as the one-group kernel applies them — and
// store the tile in 32-byte row pieces.  Deterministic; NOT the bits of the one-group kernel (another summation order over k),
// so launch_gemm's rule for it depends on the product's own shape only (never on lanes): a lane of gpbo_lml_batch stays bitwise
// gpbo_lml, and no product of a fit with NP <= 2048 is deep enough to take it (the three fit paths keep their bits).
// Why.  fp64 MFMAs reach their 64-cycle cadence only with four waves on a SIMD (posterior_kernel_v2.hip: 1 wave 140 cycles per
// instruction, 2 waves 102, 4 waves 63), and a 64x64 tile is ONE wave per SIMD: a tile with the full k-range of a triangular
// product at N = 4096 (k = 2048: 128 stages x 16 MFMAs) needs >= 119 us even with a CU to itself — more than the whole product
// needs at the matrix peak (109 us) — and was the launch (175 us measured; profiles/r06_before_lml_4096_timeline.txt).  The levels
// of W = L^-1 and the rank-1024 trailing updates of the Cholesky have 128 ... 1024 such tiles: too few for four one-group
// workgroups per CU to fill every SIMD, too uneven for 128x128 tiles.  With sixteen waves per tile a CU works on one tile at
// the pipe's full rate, the longest tile takes a quarter of the stages per wave, and the tiles are still dealt longest first.
constexpr size_t FAT_LDS_BYTES = (size_t)4 * GT_LDS_DOUBLES * sizeof(double);   // 139 264 B: one workgroup per CU
template <bool BT, bool AT>
__global__ __launch_bounds__(1024) void gemm_fat_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double fat_lds[];
  const int zl = blockIdx.z / g.batch, bz = blockIdx.z - zl * g.batch;
  int bm, bn;
  gemm64_tile_of_block(g, bm, bn);
  if (g.lower_only && bn > bm) return;
  if (bn < g.skip00 && bm < g.skip00) return;
  int kbeg, kend;
  gemm_tile_krange(g, bm, bn, kbeg, kend);
  const int q = max(kend - kbeg, 0) / 4;              // a multiple of 16 (launch_gemm: k and the triangular cuts are multiples of 64)
  const int grp = (int)(threadIdx.x >> 8);
  double* part = fat_lds + grp * GT_LDS_DOUBLES;       // staging buffers of the group's k-loop, then its 64x64 partial tile
  GemmArgs h = g;
  h.alpha = 1.0;
  gemm_tile_body_k<BT, AT, true>(h, bm, bn, zl, bz, part, (int)(threadIdx.x & 255), true, part, kbeg + grp * q, kbeg + (grp + 1) * q);
  __syncthreads();
  const int row = (int)(threadIdx.x >> 4), c4 = (int)(threadIdx.x & 15) * 4;
  const d4 p0 = *reinterpret_cast<const d4*>(fat_lds + row * 64 + c4);
  const d4 p1 = *reinterpret_cast<const d4*>(fat_lds + GT_LDS_DOUBLES + row * 64 + c4);
  const d4 p2 = *reinterpret_cast<const d4*>(fat_lds + 2 * GT_LDS_DOUBLES + row * 64 + c4);
  const d4 p3 = *reinterpret_cast<const d4*>(fat_lds + 3 * GT_LDS_DOUBLES + row * 64 + c4);
  const d4 sum = (p0 + p1) + (p2 + p3);
  double* C = g.C + (int64_t)zl * g.lane_stride + (int64_t)bz * g.strideC;
  d4* cp = reinterpret_cast<d4*>(C + ((int64_t)bm * 64 + row) * g.ldc + (int64_t)bn * 64 + c4);
  d4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = g.alpha * sum[e];
  if (g.beta != 0.0) {
    const d4 c0 = *cp;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += g.beta * c0[e];
  }
  *cp = v;
}

// ------------------------------------------------------------------------------------------------
// fp64 MFMA GEMM for the big fit-side products (rank-512 trailing updates, W = L^-1 levels, W^T W):
// 128x128 output tile per 512-thread workgroup — 8 waves as 4 (rows) x 2 (columns), wave tile 32 x 64 = 2 x 4
// v_mfma_f64_16x16x4_f64 tiles = 64 accumulator VGPRs, two workgroups per CU (4 waves per SIMD, the occupancy the
// matrix pipe needs: posterior_kernel_v2.hip) — BK = 16 per stage, DOUBLE-BUFFERED LDS with ONE barrier per stage:
// the global loads of stage s+1 are issued before the MFMAs of stage s and stored to the other buffer after them.
// Both operand tiles live k-major in LDS ([k][m], row stride 144 doubles = 128 + 16, so the two k-rows a 32-lane
// ds_read_b64 group touches fall 32 banks apart): a fragment read is conflict-free whichever way the operand lies
// in memory, and only the global->LDS copy differs between the layouts:
//   row-type source (k contiguous: A, or B given transposed): thread = (row t & 127, k-quad t >> 7), one 32-byte
//     load, four 8-byte LDS stores (a wave stores 64 consecutive rows of one k: conflict-free);
//   col-type source (m contiguous: A given transposed, or B): thread = (k t >> 5, column quad t & 31), one 32-byte
//     load, two 16-byte LDS stores.
// Same flags as gemm_f64_kernel.  m, n multiples of 64 (a ragged last 128-block clamps its loads and drops the
// stores), k multiple of 16.  With lower_only the diagonal blocks skip the wave tiles strictly above the diagonal.
constexpr int G2_B = 128, G2_BK = 16;
// LDS stage: [operand A | B][k-quad q (4)][16-row / 16-column group (8)][k in quad (4)][row / column in group (16)] doubles,
// i.e. blocks of 64 doubles = 512 B holding exactly one fragment read of v_mfma_f64_16x16x4_f64 (lane l -> element l).
// Every fragment read of both stage buffers is one of two loop-invariant per-lane addresses (A, B) plus an immediate
// multiple of 512 B (ds_read2st64_b64, conflict-free: a wave reads 512 contiguous bytes); the staging stores hit their
// 128-byte bank windows 4- to 8-fold (groups lie 512 B apart), i.e. take twice their minimum — 8 stores against 24
// fragment reads and 32 MFMAs per wave and stage.  The operands arrive through buffer descriptors with
// the k-walk folded into the per-stage descriptor base (SALU): the MFMA loop carries no address arithmetic on the VALU
// (round 2 first version: [k][144] rows and flat global loads, 0.6 non-MFMA VALU instructions per MFMA; the same change
// as in posterior_kernel_v2.hip, where it was worth 4 %).
constexpr int G2_OP = 4 * 8 * 64;        // doubles per operand per stage buffer (2048 = 16 KiB)
constexpr int G2_TILE = G2_OP;           // (LDS bytes = 4 * G2_TILE * 8)
constexpr int G2_BUF_FLAGS = 0x00020000; // gfx9 raw buffer descriptor word 3

typedef unsigned int g2_u4 __attribute__((ext_vector_type(4)));

template <bool BT, bool AT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm128_f64_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double g2_smem[];   // [2 buffers][A | B][G2_OP]
  int bn = blockIdx.x, bm = blockIdx.y;                             // long k-ranges first (see gemm_f64_kernel)
  if (g.a_lower) bm = (int)(gridDim.y - 1 - blockIdx.y);
  if (g.b_lower) { bm = (int)blockIdx.x; bn = (int)blockIdx.y; }
  if (g.tri_grid) lower_tile_of((int)blockIdx.x, bm, bn);          // live tiles only
  if (g.lower_only && bn > bm) return;
  const int zl = blockIdx.z / g.batch, bz = blockIdx.z - zl * g.batch;   // lane mode: z = lane * batch + b
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t lo = (int64_t)zl * g.lane_stride;
  const double* A = g.A + lo + (int64_t)bz * g.strideA;
  const double* B = g.B + lo + (int64_t)bz * g.strideB;
  double* C = g.C + lo + (int64_t)bz * g.strideC;
  int kbeg = 0, kend = g.k;
  if (g.a_lower) kend = min(kend, (bm + 1) * G2_B);
  if (g.b_lower) kbeg = bn * G2_B;
  if (g.k_from_tile) kbeg = max(bm, bn) * G2_B;   // both operands vanish above their diagonal tiles
  const int nst = (kend - kbeg) / G2_BK;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;
  const bool idle = g.lower_only && bm == bn && wn >= wm + 32;   // wave tile strictly above the diagonal

  // ---- global -> registers: wave-uniform tile base (advanced per stage by scalar arithmetic) + a constant per-lane offset
  const double* abase; const double* bbase; int64_t astep, bstep;   // doubles
  unsigned avoff, bvoff;                                            // bytes
  int a_st, b_st;                                                   // LDS store index (doubles) inside an operand tile
  auto blk = [](int q, int grp, int kk4, int x16) { return (q * 8 + grp) * 64 + kk4 * 16 + x16; };
  if (AT) {   // A given as (k, m): thread = (k row kk, 4 consecutive m)
    const int kk = tid >> 5, c4 = (tid & 31) * 4;
    int col = bm * G2_B + c4;
    if (col >= g.m) col = g.m - 4;
    abase = A + (int64_t)kbeg * g.lda + (int64_t)bm * G2_B;
    avoff = (unsigned)(((int64_t)kk * g.lda + (col - bm * G2_B)) * 8);
    astep = (int64_t)G2_BK * g.lda;
    a_st = blk(kk >> 2, c4 >> 4, kk & 3, c4 & 15);
  } else {    // A (m, k): thread = (row, 4 consecutive k = one k-quad)
    const int row = tid & 127, qs = tid >> 7;
    int r = bm * G2_B + row;
    if (r >= g.m) r = g.m - 1;
    abase = A + (int64_t)bm * G2_B * g.lda + kbeg;
    avoff = (unsigned)(((int64_t)(r - bm * G2_B) * g.lda + qs * 4) * 8);
    astep = G2_BK;
    a_st = blk(qs, row >> 4, 0, row & 15);     // k = 4 qs + j goes to a_st + 16 j
  }
  if (BT) {   // B given as (n, k)
    const int row = tid & 127, qs = tid >> 7;
    int r = bn * G2_B + row;
    if (r >= g.n) r = g.n - 1;
    bbase = B + (int64_t)bn * G2_B * g.ldb + kbeg;
    bvoff = (unsigned)(((int64_t)(r - bn * G2_B) * g.ldb + qs * 4) * 8);
    bstep = G2_BK;
    b_st = blk(qs, row >> 4, 0, row & 15);
  } else {    // B (k, n)
    const int kk = tid >> 5, c4 = (tid & 31) * 4;
    int col = bn * G2_B + c4;
    if (col >= g.n) col = g.n - 4;
    bbase = B + (int64_t)kbeg * g.ldb + (int64_t)bn * G2_B;
    bvoff = (unsigned)(((int64_t)kk * g.ldb + (col - bn * G2_B)) * 8);
    bstep = (int64_t)G2_BK * g.ldb;
    b_st = blk(kk >> 2, c4 >> 4, kk & 3, c4 & 15);
  }
  auto gload = [&](int st, d2v(&ra)[2], d2v(&rb)[2]) {
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(abase + (int64_t)st * astep), 0,
                                                                         0x7fffffff, G2_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(bbase + (int64_t)st * bstep), 0,
                                                                         0x7fffffff, G2_BUF_FLAGS);
    ra[0] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rsa, avoff, 0, 0));
    ra[1] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rsa, avoff + 16u, 0, 0));
    rb[0] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rsb, bvoff, 0, 0));
    rb[1] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rsb, bvoff + 16u, 0, 0));
  };
  auto lstore = [&](int buf, const d2v(&ra)[2], const d2v(&rb)[2]) {
    double* As = g2_smem + buf * 2 * G2_OP;
    double* Bs = As + G2_OP;
    if (AT) {
      *reinterpret_cast<d2v*>(As + a_st) = ra[0];
      *reinterpret_cast<d2v*>(As + a_st + 2) = ra[1];
    } else {
      As[a_st] = ra[0].x; As[a_st + 16] = ra[0].y; As[a_st + 32] = ra[1].x; As[a_st + 48] = ra[1].y;
    }
    if (BT) {
      Bs[b_st] = rb[0].x; Bs[b_st + 16] = rb[0].y; Bs[b_st + 32] = rb[1].x; Bs[b_st + 48] = rb[1].y;
    } else {
      *reinterpret_cast<d2v*>(Bs + b_st) = rb[0];
      *reinterpret_cast<d2v*>(Bs + b_st + 2) = rb[1];
    }
  };

  d4 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t][u] = d4{0.0, 0.0, 0.0, 0.0};

  if (nst > 0) {
    d2v ra[2], rb[2];
    gload(0, ra, rb);
    lstore(0, ra, rb);
    __syncthreads();
    const int last = nst - 1;
    // per-lane fragment addresses: the wave's first row / column group, the lane's slot in a block
    const double* fa = g2_smem + (wm >> 4) * 64 + lane;
    const double* fb = g2_smem + G2_OP + (wn >> 4) * 64 + lane;
    for (int st = 0; st < nst; ++st) {
      const int buf = st & 1;
      gload(min(st + 1, last), ra, rb);            // clamped look-ahead keeps the body branch-free
      __builtin_amdgcn_sched_barrier(0);           // keep the global loads at the top of the stage
      if (!idle) {
        const double* fas = fa + buf * 2 * G2_OP;  // the stage buffer: two address updates per stage, the rest immediates
        const double* fbs = fb + buf * 2 * G2_OP;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double a0 = fas[q * 512];
          const double a1 = fas[q * 512 + 64];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double b = fbs[q * 512 + 64 * u];
            acc[0][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0][u], 0, 0, 0);
            acc[1][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1][u], 0, 0, 0);
          }
        }
      }
      lstore(buf ^ 1, ra, rb);    // the other buffer: everyone finished reading it before the previous barrier
      __syncthreads();
    }
  }
  if (idle) return;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = (int64_t)bm * G2_B + wm + 16 * t + (lane >> 4) + 4 * r;
        const int64_t colx = (int64_t)bn * G2_B + wn + 16 * u + (lane & 15);
        if (row < g.m && colx < g.n) {
          double* cp = C + row * g.ldc + colx;
          double v = g.alpha * acc[t][u][r];
          if (g.beta != 0.0) v += g.beta * (*cp);
          *cp = v;
        }
      }
}

static bool gemm128_enabled() {
  static const bool on = !(dbg_env("GPBO_GEMM128") && dbg_env("GPBO_GEMM128")[0] == '0');
  return on;
}

// Sixteen waves per 64x64 tile (gemm_fat_kernel) where one wave per SIMD and tile is what holds a product back: products over a
// TRIANGULAR operand (every tile another k-length, the longest tile is the launch) with at most ~2 tiles per CU, and any deep
// product with less than one tile per CU.  Measured at N = 4096 (profiles/r06_gemm_fat_ab.json, r06 timelines): the 2048-level of
// W = L^-1 (2 x 256 tiles, k <= 1024) 82.8 -> 54.9 us per launch, the last trailing update (136 tiles, k = 1024) 55 -> 44; NOT the
// 4096-level (1024 tiles: 175 us either way — the tile body's own ~0.77 of the matrix pipe and the operand traffic of a 64x64
// tile, 8 flop per byte, bound it) and not the rank-1024 updates with 400-650 uniform tiles (130 -> 140: four one-group workgroups
// per CU already put four waves on every SIMD there).  The rule reads the product's OWN shape, never g.lanes: a lane of
// gpbo_lml_batch stays bitwise gpbo_lml.  Debug build: GPBO_GEMM_FAT=0 off; GPBO_GEMM_FAT_LIMIT / _SMALL / _MINK the thresholds.
bool gemm_fat_rule(const GemmArgs& g) {
  const char* fe = dbg_env("GPBO_GEMM_FAT");
  if (fe && fe[0] == '0') return false;
  const int limit = dbg_env("GPBO_GEMM_FAT_LIMIT") ? atoi(dbg_env("GPBO_GEMM_FAT_LIMIT")) : 600;
  const int small = dbg_env("GPBO_GEMM_FAT_SMALL") ? atoi(dbg_env("GPBO_GEMM_FAT_SMALL")) : 200;
  const int mink = dbg_env("GPBO_GEMM_FAT_MINK") ? atoi(dbg_env("GPBO_GEMM_FAT_MINK")) : 512;
  if (g.k % 64 || g.k < mink || g.k_from_tile || (g.ldc % 4)) return false;
  const int64_t tiles64 = (int64_t)(g.m / 64) * (g.n / 64) * g.batch / (g.lower_only ? 2 : 1);
  return ((g.a_lower || g.b_lower) && tiles64 <= limit) || tiles64 <= small;
}

int launch_gemm(gpbo_ctx* ctx, const GemmArgs& g_in) {
  if (g_in.m <= 0 || g_in.n <= 0 || g_in.batch <= 0) return GPBO_OK;
  GemmArgs g = g_in;
  g.lanes = ctx->lanes;
  g.lane_stride = ctx->lane_stride;
  if (g.m % 64 || g.n % 64 || g.k % 16) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gemm: m,n must be multiples of 64 and k of 16");
  if (g.a_trans && g.b_trans) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gemm: a_trans with b_trans is not instantiated");
  // big products -> the 128x128 double-buffered kernel; panels and small blocks -> the 64x64 kernel
  // (a triangular k-range must start/stop on a 128 boundary there, which the callers' block sizes guarantee from 128 on)
  // Measured (scripts/archive/r02_fit_probe.py, MI355X): 4096^3 60 vs 50 TFLOP/s, rank-512 SYRK at 3584 / 7680 rows 44 / 47 vs 38 / 41 —
  // but the 64x64 kernel wins where there are too few 128-blocks to fill 256 CUs (1024^3: 15 vs 28) and on rank-64 panel
  // updates (8 vs 15), so: deep k and at least ~a chip's worth of 128x128 blocks.
  const int64_t blocks128 = (int64_t)((g.m + 127) / 128) * ((g.n + 127) / 128) * g.batch * g.lanes / (g.lower_only ? 2 : 1);
  // Triangular operands give every tile another k-length (W = L^-1, W^T W): a grid of at most one 128x128 tile per
  // workgroup slot (512) ends on its longest tiles while most CUs idle, so such products take the 64x64 tiles (4x as
  // many, each a quarter of the work, handed out longest first).  Measured (r02 fit probe): W = L^-1 at N = 4096
  // 0.93 -> 0.82 ms; with deeper grids (N = 8192: 1024 tiles) the 128x128 kernel wins, 3.91 vs 4.11 ms.
  // GPBO_TRI64=0 turns the rule off (A/B runs).
  static const bool tri64 = !(dbg_env("GPBO_TRI64") && dbg_env("GPBO_TRI64")[0] == '0');
  const bool triangular = g.a_lower || g.b_lower || g.k_from_tile;
  // ... and W^T W (k_from_tile: the first tile's k-loop is the whole N) takes them while one 128x128 tile per slot would make that
  // tile the launch: N = 4096, 528 tiles: 1.04 ms, the longest tile alone 1.05 ms at a 512th of the chip's rate
  static const int tri64_limit = dbg_env("GPBO_TRI64_LIMIT") ? atoi(dbg_env("GPBO_TRI64_LIMIT")) : 600;
  // Lower-only square products (W^T W, the SYRK-shaped trailing updates): a 1-D grid of the LIVE tiles in lower-triangle order
  // instead of the square grid whose upper half exits at once.  The hardware deals workgroups to the 8 XCDs by id mod 8; in the
  // square grid that is the column tile mod 8, and column c of a lower triangle has nt - c live tiles: XCD 0 carried 38 % more
  // of W^T W's work at N = 4096 than XCD 7 (6672 against 4824 k-tiles; mean 5720) and the launch ended on it.  In lower-triangle
  // order neighbouring ids are tiles of (almost) equal k-length — rows from the top = longest first for W^T W — so every XCD gets
  // every eighth of them.  Same tiles, same arithmetic, same bits.  (GPBO_TRI_GRID=0: the square grid, debug build, A/B.)
  const bool tri_grid = g.lower_only && g.m == g.n && !g.a_lower && !g.b_lower &&
                        !(dbg_env("GPBO_TRI_GRID") && dbg_env("GPBO_TRI_GRID")[0] == '0');
  g.tri_grid = tri_grid ? 1 : 0;
  {
    if (g.fat > 0 || (g.fat == 0 && gemm_fat_rule(g))) {
      if (!(ctx->func_attrs & ATTR_GEMM_FAT)) {
        GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fat_kernel<true, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)FAT_LDS_BYTES));
        GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fat_kernel<false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)FAT_LDS_BYTES));
        GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fat_kernel<false, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)FAT_LDS_BYTES));
        ctx->func_attrs |= ATTR_GEMM_FAT;
      }
      dim3 grid((unsigned)(g.n / 64), (unsigned)(g.m / 64), (unsigned)(g.batch * g.lanes));
      if (g.b_lower) std::swap(grid.x, grid.y);
      if (tri_grid) { grid.x = grid.y * (grid.y + 1) / 2; grid.y = 1; }
      if (g.b_trans)
        gemm_fat_kernel<true, false><<<grid, dim3(1024), FAT_LDS_BYTES, ctx->stream>>>(g);
      else if (g.a_trans)
        gemm_fat_kernel<false, true><<<grid, dim3(1024), FAT_LDS_BYTES, ctx->stream>>>(g);
      else
        gemm_fat_kernel<false, false><<<grid, dim3(1024), FAT_LDS_BYTES, ctx->stream>>>(g);
      GPBO_HIP(ctx, hipGetLastError());
      return GPBO_OK;
    }
  }
  const bool prefer64 = tri64 && triangular && (blocks128 < 512 || (g.k_from_tile && blocks128 < tri64_limit));
  if (gemm128_enabled() && !prefer64 && g.m >= 128 && g.n >= 128 && g.k >= 256 && blocks128 >= 192) {
    constexpr size_t lds = (size_t)4 * G2_TILE * sizeof(double);   // 65 536 B
    if (!(ctx->func_attrs & ATTR_GEMM128)) {
      GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(gemm128_f64_kernel<true, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(gemm128_f64_kernel<false, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(gemm128_f64_kernel<false, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      ctx->func_attrs |= ATTR_GEMM128;
    }
    dim3 grid((unsigned)((g.n + 127) / 128), (unsigned)((g.m + 127) / 128), (unsigned)(g.batch * g.lanes));
    if (g.b_lower) std::swap(grid.x, grid.y);
    if (tri_grid) { grid.x = grid.y * (grid.y + 1) / 2; grid.y = 1; }
    if (g.b_trans)
      gemm128_f64_kernel<true, false><<<grid, dim3(512), lds, ctx->stream>>>(g);
    else if (g.a_trans)
      gemm128_f64_kernel<false, true><<<grid, dim3(512), lds, ctx->stream>>>(g);
    else
      gemm128_f64_kernel<false, false><<<grid, dim3(512), lds, ctx->stream>>>(g);
    GPBO_HIP(ctx, hipGetLastError());
    return GPBO_OK;
  }
  dim3 grid((unsigned)(g.n / 64), (unsigned)(g.m / 64), (unsigned)(g.batch * g.lanes));
  if (g.b_lower) std::swap(grid.x, grid.y);
  if (tri_grid) { grid.x = grid.y * (grid.y + 1) / 2; grid.y = 1; }
  if (g.b_trans)
    gemm_f64_kernel<true, false><<<grid, dim3(256), 0, ctx->stream>>>(g);
  else if (g.a_trans)
    gemm_f64_kernel<false, true><<<grid, dim3(256), 0, ctx->stream>>>(g);
  else
    gemm_f64_kernel<false, false><<<grid, dim3(256), 0, ctx->stream>>>(g);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// ------------------------------------------------------------------------------------------------
// W := blockdiag(dinv), behind a zero fill of the whole square when `zero_fill` (see trtri, gpbo_api.hip).
__global__ __launch_bounds__(256) void fill_w_diag_kernel(const double* __restrict__ dinv,
                                                          double* __restrict__ W, int64_t NP, int64_t lane_stride, int zero_right) {
  fill_w_diag_body(dinv + (int64_t)blockIdx.y * lane_stride, W + (int64_t)blockIdx.y * lane_stride, NP, (int)blockIdx.x,
                   (int)threadIdx.x, zero_right != 0);
}

int launch_fill_w_diag(gpbo_ctx* ctx, Model& m, bool zero_fill) {
  if (!zero_fill && dbg_env("GPBO_POISON_W")) {
    // debug build (tests/test_gpu_parity.py): what an LML evaluation must not read is made of NaNs instead of being left as it was
    for (int l = 0; l < ctx->lanes; ++l)
      GPBO_HIP(ctx, hipMemsetAsync(m.W + (int64_t)l * ctx->lane_stride, 0xFF, (size_t)m.NP * m.NP * sizeof(double), ctx->stream));
  }
  if (zero_fill) {
    if (ctx->lanes == 1)
      GPBO_HIP(ctx, hipMemsetAsync(m.W, 0, (size_t)m.NP * m.NP * sizeof(double), ctx->stream));
    else
      GPBO_HIP(ctx, hipMemset2DAsync(m.W, (size_t)ctx->lane_stride * sizeof(double), 0, (size_t)m.NP * m.NP * sizeof(double),
                                     (size_t)ctx->lanes, ctx->stream));
  }
  fill_w_diag_kernel<<<dim3((unsigned)(m.NP / 64), (unsigned)ctx->lanes), dim3(256), 0, ctx->stream>>>(m.dinv, m.W, m.NP,
                                                                                                      ctx->lane_stride, zero_fill ? 0 : 1);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// ------------------------------------------------------------------------------------------------
// t = W y (one wave per row, fixed shuffle tree) and alpha = W^T t (64 columns per workgroup): bodies in fit_bodies.h.
__global__ __launch_bounds__(256) void trmv_lower_kernel(const double* __restrict__ W,
                                                         const double* __restrict__ y,
                                                         double* __restrict__ t, int64_t NP, int64_t lane_stride) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= NP) return;
  W += (int64_t)blockIdx.y * lane_stride;
  y += (int64_t)blockIdx.y * lane_stride;
  t += (int64_t)blockIdx.y * lane_stride;
  trmv_lower_row(W, y, t, NP, i, lane);
}

// alpha = W^T t in two deterministic passes: (column block of 64) x (row split) partial sums, then a
// fixed-order reduction over the row splits.
__global__ __launch_bounds__(256) void trmv_lower_t_kernel(const double* __restrict__ W,
                                                           const double* __restrict__ t,
                                                           double* __restrict__ partial, int64_t NP,
                                                           int64_t lane_stride) {
  __shared__ double red[4 * 64];
  W += (int64_t)blockIdx.z * lane_stride;
  t += (int64_t)blockIdx.z * lane_stride;
  partial += (int64_t)blockIdx.z * lane_stride;
  trmv_lower_t_body(W, t, partial, NP, (int)blockIdx.x, (int)blockIdx.y, red, (int)threadIdx.x);
}

__global__ __launch_bounds__(256) void trmv_reduce_kernel(const double* __restrict__ partial,
                                                          double* __restrict__ alpha, int64_t NP,
                                                          int64_t lane_stride) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= NP) return;
  partial += (int64_t)blockIdx.y * lane_stride;
  alpha += (int64_t)blockIdx.y * lane_stride;
  trmv_reduce_elem(partial, alpha, NP, j);
}

int launch_trmv(gpbo_ctx* ctx, Model& m) {
  const unsigned lanes = (unsigned)ctx->lanes;
  const int64_t ls = ctx->lane_stride;
  trmv_lower_kernel<<<dim3((unsigned)((m.NP + 3) / 4), lanes), dim3(256), 0, ctx->stream>>>(m.W, m.yn, m.tvec, m.NP, ls);
  GPBO_HIP(ctx, hipGetLastError());
  // m.tmp (>= NP*64 doubles) is free again after trtri
  trmv_lower_t_kernel<<<dim3((unsigned)(m.NP / 64), TRMV_SPLITS, lanes), dim3(256), 0, ctx->stream>>>(m.W, m.tvec, m.tmp, m.NP, ls);
  GPBO_HIP(ctx, hipGetLastError());
  trmv_reduce_kernel<<<dim3((unsigned)((m.NP + 255) / 256), lanes), dim3(256), 0, ctx->stream>>>(m.tmp, m.alpha, m.NP, ls);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// ------------------------------------------------------------------------------------------------
// Rank-one growth of the factorisation (gpbo_fit_append): observation j joins a fitted model of j rows.
//   append_kvec   : kv[i] = k(x_i, x_j) for i < j (the arithmetic of kmat_kernel, so K's new row is the row a
//                   full fit would write), 0 for i >= j; also written to K[j][0..j-1], K[j][j] = 1 + noise
//   trmv_lower    : l = W kv            (existing kernel; rows >= j give 0 because kv is 0 there)
//   trmv_lower_t  : u = W^T l           (existing kernels)
//   append_finish : lambda^2 = 1 + noise - sum l^2 (fixed tree); L[j][:] = [l, lambda]; W[j][:] = [-u/lambda, 1/lambda]
template <int KERNEL>
__global__ __launch_bounds__(256) void append_kvec_kernel(const double* __restrict__ Xs, int DP, int64_t j, int64_t NP,
                                                          double noise, double* __restrict__ kv, double* __restrict__ K) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= NP) return;
  double v = 0.0;
  if (i < j) {
    const double* xi = Xs + i * DP;
    const double* xj = Xs + j * DP;
    double d2 = 0.0;
    for (int t = 0; t < DP; ++t) {
      const double df = xj[t] - xi[t];
      d2 = fma(df, df, d2);
    }
    v = kernel_value<KERNEL>(d2);
    K[j * NP + i] = v;
  } else if (i == j) {
    K[j * NP + j] = 1.0 + noise;
  }
  kv[i] = v;
}

__global__ __launch_bounds__(256) void append_finish_kernel(const double* __restrict__ l, const double* __restrict__ u,
                                                            int64_t j, int64_t NP, double noise, double* __restrict__ L,
                                                            double* __restrict__ W, int* __restrict__ info) {
  __shared__ double sh[4];
  __shared__ double lam_sh;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < j; i += 256) s = fma(l[i], l[i], s);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double piv = (1.0 + noise) - (((sh[0] + sh[1]) + sh[2]) + sh[3]);
    double lam = 0.0;
    if (piv > 0.0) lam = sqrt(piv);
    else if (*info == 0) *info = (int)(j + 1);     // LAPACK potrf: order of the first non-positive minor
    lam_sh = lam;
  }
  __syncthreads();
  const double lam = lam_sh;
  if (!(lam > 0.0)) return;
  const double inv = 1.0 / lam;
  for (int64_t i = threadIdx.x; i < j; i += 256) {
    L[j * NP + i] = l[i];
    W[j * NP + i] = -u[i] * inv;
  }
  if (threadIdx.x == 0) {
    L[j * NP + j] = lam;
    W[j * NP + j] = inv;
  }
}

int launch_append_row(gpbo_ctx* ctx, Model& m, int64_t j) {
  // scratch in m.tmp (>= NP * 64 doubles): [0, 16 NP) partials of W^T l, then kv, l, u
  double* partial = m.tmp;
  double* kv = m.tmp + (int64_t)TRMV_SPLITS * m.NP;
  double* lv = kv + m.NP;
  double* uv = lv + m.NP;
  const unsigned vb = (unsigned)((m.NP + 255) / 256);
  if (m.kernel == GPBO_KERNEL_MATERN25)
    append_kvec_kernel<GPBO_KERNEL_MATERN25><<<dim3(vb), dim3(256), 0, ctx->stream>>>(m.Xs, m.DP, j, m.NP, m.noise, kv, m.K);
  else
    append_kvec_kernel<GPBO_KERNEL_RBF><<<dim3(vb), dim3(256), 0, ctx->stream>>>(m.Xs, m.DP, j, m.NP, m.noise, kv, m.K);
  GPBO_HIP(ctx, hipGetLastError());
  trmv_lower_kernel<<<dim3((unsigned)((m.NP + 3) / 4)), dim3(256), 0, ctx->stream>>>(m.W, kv, lv, m.NP, 0);
  GPBO_HIP(ctx, hipGetLastError());
  trmv_lower_t_kernel<<<dim3((unsigned)(m.NP / 64), TRMV_SPLITS), dim3(256), 0, ctx->stream>>>(m.W, lv, partial, m.NP, 0);
  GPBO_HIP(ctx, hipGetLastError());
  trmv_reduce_kernel<<<dim3(vb), dim3(256), 0, ctx->stream>>>(partial, uv, m.NP, 0);
  GPBO_HIP(ctx, hipGetLastError());
  append_finish_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(lv, uv, j, m.NP, m.noise, m.L, m.W, ctx->info_dev);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// ------------------------------------------------------------------------------------------------
// Pack W into the order the posterior kernel's waves consume it (pack_w_elem, fit_bodies.h).
__global__ __launch_bounds__(256) void pack_w_kernel(const double* __restrict__ W,
                                                     double* __restrict__ Wp, int64_t N, int64_t NP) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= NP * NP) return;
  pack_w_elem(W, Wp, N, NP, idx);
}

int launch_pack_w(gpbo_ctx* ctx, Model& m) {
  int64_t total = m.NP * m.NP;
  pack_w_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream>>>(m.W, m.Wp, m.N, m.NP);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

}  // namespace gpbo
