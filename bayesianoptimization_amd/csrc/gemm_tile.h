// 64x64-tile fp64 MFMA GEMM body shared by fit_kernels.hip (gemm_f64_kernel, chol_step_kernel) and chol_kernels.hip
// (chol128_step_kernel): device code only, included by the .hip translation units that instantiate it.
#pragma once

#include "gpbo_internal.h"

namespace gpbo {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int GT_LDS_DOUBLES = 2 * 2 * 16 * 68;   // two stages x (A | B) x [16][68]

// ------------------------------------------------------------------------------------------------
// fp64 MFMA GEMM, 64x64 output tile per 256-thread workgroup (4 waves, 32x32 each as 2x2
// v_mfma_f64_16x16x4_f64 tiles), BK = 16, operands staged k-major in LDS.
//   C = alpha * A(m,k) * op(B) + beta * C ;  A row-major; B row-major (k,n), or (n,k) if b_trans.
// Fragment layout (cdna_hip_programming.md §3): A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15],
// D lane l, reg r = D[(l>>4) + 4r][l&15].
typedef double d2v __attribute__((ext_vector_type(2)));

// How many workgroup barriers gemm_tile_body executes for tile (bm, bn): 0 for a tile it skips, else one after the prologue and one
// per 16-deep stage.  (fused_small.hip runs two tiles side by side in one 512-thread workgroup and evens the counts out.)
__device__ __forceinline__ int gemm_tile_barriers(const GemmArgs& g, const int bm, const int bn) {
  if (g.lower_only && bn > bm) return 0;
  if (bn < g.skip00 && bm < g.skip00) return 0;
  int kbeg = 0, kend = g.k;
  if (g.a_lower) kend = min(kend, (bm + 1) * 64);
  if (g.b_lower) kbeg = bn * 64;
  if (g.k_from_tile) kbeg = max(bm, bn) * 64;
  const int nst = (kend - kbeg) / 16;
  return nst > 0 ? 1 + nst : 0;
}

// k-range [kbeg, kend) of output tile (bm, bn): what the operands' triangular shapes leave of [0, k)
__device__ __forceinline__ void gemm_tile_krange(const GemmArgs& g, const int bm, const int bn, int& kbeg, int& kend) {
  kbeg = 0; kend = g.k;
  if (g.a_lower) kend = min(kend, (bm + 1) * 64);
  if (g.b_lower) kbeg = bn * 64;
  if (g.k_from_tile) kbeg = max(bm, bn) * 64;   // both operands vanish above their diagonal tiles
}

template <bool BT, bool AT, bool CT = false>
// `tid` = the thread's index inside the 256 threads working on this tile (a 512-thread workgroup runs two tiles side by
// side, each with its own LDS area and the same number of barriers); `write` false = go through the motions on a valid
// tile but leave C alone (the partner half of such a workgroup when it has no tile of its own).
// CT: the tile alpha * A B goes to the 64x64 row-major image `c_tile` (LDS of the caller: lml_kernels.hip consumes K^-1 tile
// by tile without a round trip through memory) instead of C.
// gemm_tile_body_k: the same over an explicit k-range [kbeg, kend) (a multiple of 16 long) — a k-slice of the tile
// (gemm_fat_kernel: four 256-thread groups of one workgroup each take a quarter of the tile's k-range).
__device__ __forceinline__ void gemm_tile_body_k(const GemmArgs& g, int bm, int bn, int zl, int bz, double* lds, const int tid,
                                                 const bool write, double* c_tile, const int kbeg, const int kend);

template <bool BT, bool AT, bool CT = false>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs& g, int bm, int bn, int zl, int bz, double* lds, const int tid = threadIdx.x,
                                               const bool write = true, double* c_tile = nullptr) {
  if (g.lower_only && bn > bm) return;
  if (bn < g.skip00 && bm < g.skip00) return;      // the leading skip00 x skip00 tiles belong to other workgroups / launches
  int kbeg, kend;
  gemm_tile_krange(g, bm, bn, kbeg, kend);
  gemm_tile_body_k<BT, AT, CT>(g, bm, bn, zl, bz, lds, tid, write, c_tile, kbeg, kend);
}

template <bool BT, bool AT, bool CT>
__device__ __forceinline__ void gemm_tile_body_k(const GemmArgs& g, int bm, int bn, int zl, int bz, double* lds, const int tid,
                                                 const bool write, double* c_tile, const int kbeg, const int kend) {
  // two LDS stages: the global loads of stage s+1 are issued before the MFMAs of stage s and parked in the other buffer
  // afterwards — one barrier per 16-deep stage (round 1: one buffer, two barriers, loads exposed in front of every stage)
  typedef double (*stage_t)[16][68];
  stage_t As = reinterpret_cast<stage_t>(lds);                    // [2][16][68]
  stage_t Bs = reinterpret_cast<stage_t>(lds + 2 * 16 * 68);      // [2][16][68]
  const int wave = tid >> 6, lane = tid & 63;
  const int64_t lo = (int64_t)zl * g.lane_stride;
  const double* A = g.A + lo + (int64_t)bz * g.strideA + (AT ? (int64_t)bm * 64 : (int64_t)bm * 64 * g.lda);
  const double* B = g.B + lo + (int64_t)bz * g.strideB + (BT ? (int64_t)bn * 64 * g.ldb : (int64_t)bn * 64);
  double* C = g.C + lo + (int64_t)bz * g.strideC;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  d4 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[t][u] = d4{0.0, 0.0, 0.0, 0.0};
  const int arow = tid >> 2, akq = (tid & 3) * 4;
  const int brow = tid >> 4, bnq = (tid & 15) * 4;
  const double* asrc = AT ? A + (int64_t)(kbeg + brow) * g.lda + bnq : A + (int64_t)arow * g.lda + kbeg + akq;
  const double* bsrc = BT ? B + (int64_t)arow * g.ldb + kbeg + akq : B + (int64_t)(kbeg + brow) * g.ldb + bnq;
  const int64_t astep = AT ? (int64_t)16 * g.lda : 16, bstep = BT ? 16 : (int64_t)16 * g.ldb;
  auto gload = [&](int st, d2v(&ra)[2], d2v(&rb)[2]) {
    const d2v* ap = reinterpret_cast<const d2v*>(asrc + (int64_t)st * astep);
    const d2v* bp = reinterpret_cast<const d2v*>(bsrc + (int64_t)st * bstep);
    ra[0] = ap[0]; ra[1] = ap[1];
    rb[0] = bp[0]; rb[1] = bp[1];
  };
  auto lstore = [&](int buf, const d2v(&ra)[2], const d2v(&rb)[2]) {
    if (AT) {
      *reinterpret_cast<d2v*>(&As[buf][brow][bnq]) = ra[0];
      *reinterpret_cast<d2v*>(&As[buf][brow][bnq + 2]) = ra[1];
    } else {
      As[buf][akq + 0][arow] = ra[0].x; As[buf][akq + 1][arow] = ra[0].y;
      As[buf][akq + 2][arow] = ra[1].x; As[buf][akq + 3][arow] = ra[1].y;
    }
    if (BT) {
      Bs[buf][akq + 0][arow] = rb[0].x; Bs[buf][akq + 1][arow] = rb[0].y;
      Bs[buf][akq + 2][arow] = rb[1].x; Bs[buf][akq + 3][arow] = rb[1].y;
    } else {
      *reinterpret_cast<d2v*>(&Bs[buf][brow][bnq]) = rb[0];
      *reinterpret_cast<d2v*>(&Bs[buf][brow][bnq + 2]) = rb[1];
    }
  };
  const int nst = (kend - kbeg) / 16;
  if (nst > 0) {
    // Global loads run GT_PF stages ahead of the MFMAs (a ring of register sets; LDS stays double-buffered): with one
    // stage of look-ahead a tile that has a CU to itself — the rank-512 update of the next panel's columns, the levels of
    // W = L^-1, every product whose grid does not fill the chip — waited ~1.2 us per 16-deep stage for its operands
    // (39 us for any k = 512 tile, however small the product); nothing changes in the arithmetic or its order.
#ifndef GPBO_GT_PF
#define GPBO_GT_PF 3
#endif
    constexpr int GT_PF = GPBO_GT_PF;
    d2v ra[GT_PF][2], rb[GT_PF][2];
    const int last = nst - 1;
    gload(0, ra[0], rb[0]);
    lstore(0, ra[0], rb[0]);
#pragma unroll
    for (int j = 1; j <= GT_PF; ++j) gload(min(j, last), ra[j % GT_PF], rb[j % GT_PF]);     // clamped: branch-free
    __syncthreads();
    auto stage = [&](const int st, d2v(&na)[2], d2v(&nb)[2]) {      // na / nb: the registers holding stage st + 1
      const int buf = st & 1;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int kr = kk * 4 + (lane >> 4);
        const double a0 = As[buf][kr][wm + (lane & 15)];
        const double a1 = As[buf][kr][wm + 16 + (lane & 15)];
        const double b0 = Bs[buf][kr][wn + (lane & 15)];
        const double b1 = Bs[buf][kr][wn + 16 + (lane & 15)];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
      }
      lstore(buf ^ 1, na, nb);    // the other buffer: everyone finished reading it before the previous barrier
      gload(min(st + 1 + GT_PF, last), na, nb);     // the set just emptied takes the stage GT_PF + 1 ahead
      __syncthreads();
    };
    int st = 0;
    for (; st + GT_PF <= nst; st += GT_PF) {
#pragma unroll
      for (int j = 0; j < GT_PF; ++j) stage(st + j, ra[(j + 1) % GT_PF], rb[(j + 1) % GT_PF]);     // st is a multiple of GT_PF
    }
#pragma unroll
    for (int j = 0; j < GT_PF - 1; ++j)
      if (st + j < nst) stage(st + j, ra[(j + 1) % GT_PF], rb[(j + 1) % GT_PF]);
  }
  if (!write) return;
  if constexpr (CT) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) c_tile[(wm + 16 * t + (lane >> 4) + 4 * r) * 64 + wn + 16 * u + (lane & 15)] = g.alpha * acc[t][u][r];
    return;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = (int64_t)bm * 64 + wm + 16 * t + (lane >> 4) + 4 * r;
        const int64_t colx = (int64_t)bn * 64 + wn + 16 * u + (lane & 15);
        double* cp = C + row * g.ldc + colx;
        double v = g.alpha * acc[t][u][r];
        if (g.beta != 0.0) v += g.beta * (*cp);
        *cp = v;
      }
}


}  // namespace gpbo
