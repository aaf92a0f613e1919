// Device bodies of the blocked Cholesky (gfx950), shared by chol_kernels.hip (one launch per phase) and fused_small.hip (the
// whole fit of a small problem in ONE workgroup): the diagonal-block workgroup, the 16-row panel solve and the next-diagonal
// update.  Device code only; included by the .hip translation units that instantiate it.  See chol_kernels.hip for the schedule.
#pragma once

#include "gemm_tile.h"
#include "gpbo_internal.h"

namespace gpbo {

// ---- LDS layout of the diagonal-block workgroup (doubles) --------------------------------------------------------
constexpr int DS = 80;                                   // stride of the column-major images: a fragment's four k-rows fall 32 banks apart
constexpr int C128_IMG = 64 * DS + 64;                   // one image: 64 columns + room for the last column's marker overflow (factor_quarter)
constexpr int C128_LC0 = 0;                              // L00:  Lc0[j * DS + i] = L[i][j], Lc0[j * DS + 64] = 1 / L[j][j]
constexpr int C128_LCX = C128_IMG;                       // L10 during the first factorisation and the SYRK, then L11
constexpr int C128_WR = 2 * C128_IMG;                    // SYRK exchange [64][81], then the waves' 16x16 diagonal inverses [8][4][272]
constexpr int C128_WR_DOUBLES = 8 * 4 * 272;    // 8704
constexpr int C128_FLAGS = C128_WR + C128_WR_DOUBLES;    // ints: [0] broken hand-off
constexpr int C128_LDS_DOUBLES = C128_FLAGS + 8;
constexpr size_t C128_LDS_BYTES = (size_t)C128_LDS_DOUBLES * sizeof(double);
static_assert(C128_WR_DOUBLES >= 64 * 81 && C128_WR_DOUBLES >= 8 * 4 * 272, "exchange area too small");
static_assert(C128_LDS_BYTES <= 160 * 1024, "the diagonal workgroup's LDS exceeds a CU's 160 KiB");
static_assert(C128_LDS_DOUBLES >= 2 * GT_LDS_DOUBLES, "the update tiles of the step launch alias the same dynamic LDS");

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// LDS is one in-order pipeline per CU: a wave's stores are performed in issue order, so "column, then marker" needs no
// s_waitcnt between the two (an atomic release store would put one on the chain, 64 times per block) — only the
// compiler has to keep the order.
#define GPBO_LDS_ORDER() asm volatile("" ::: "memory")
#define GPBO_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifndef GPBO_CHOL_POLL_SLEEP
#define GPBO_CHOL_POLL_SLEEP 8      // x 64 cycles between two looks at a marker, cut short by the owner's s_wakeup
#endif
#ifndef GPBO_CHOL_DEFER_MATE
#define GPBO_CHOL_DEFER_MATE 1
#endif
#ifndef GPBO_CHOL_WAKE_MASK
#define GPBO_CHOL_WAKE_MASK 1       // the owner wakes the sleepers behind every column jj with (jj & mask) == mask: every second one
#endif

// a[c] -= l * m_c and (the riding row) a2[c] -= l2 * m_c for N of the wave's own columns, m_c = the value of l in lane
// C0 + c (the rows of the wave's diagonal 8x8 block sit in lanes 0..7): v_readlane_b32 into FIXED scalar registers,
// consumed by the v_fma_f64 directly.  Written out because the compiler's version of the same thing hoists every
// v_readlane of a column to the front, runs out of SGPRs (the GEMM half of the step kernel keeps ~60 live) and spills
// them with v_writelane_b32 at 28 cycles apiece (scripts/archive/r03_col_stamps.py: 2/3 of a column's time), and because a
// v_readlane whose lane number comes from an SGPR instead of an inline constant is no faster.  The readlane -> fma
// distance satisfies the 2 wait states a VALU-written SGPR needs before a VALU reads it.
#define GPBO_RL(S, C) "v_readlane_b32 s" #S ", %[lo], %[" #C "]\n\tv_readlane_b32 s" GPBO_RL_NEXT_##S ", %[hi], %[" #C "]\n\t"
#define GPBO_RL_NEXT_80 "81"
#define GPBO_RL_NEXT_82 "83"
#define GPBO_RL_NEXT_84 "85"
#define GPBO_RL_NEXT_86 "87"
__device__ __forceinline__ void split64(const double l, int& lo, int& hi) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(l);
  lo = (int)(unsigned)u;
  hi = (int)(unsigned)(u >> 32);
}
template <int C0, bool FOLLOW>
__device__ __forceinline__ void bcast_fma4(double* a, double* a2, const double l, const double l2) {
  int lo, hi;
  split64(l, lo, hi);
  if constexpr (FOLLOW) {
    asm volatile(GPBO_RL(80, c0) GPBO_RL(82, c1) GPBO_RL(84, c2) GPBO_RL(86, c3)
                 "v_fma_f64 %[a0], -%[l], s[80:81], %[a0]\n\tv_fma_f64 %[a1], -%[l], s[82:83], %[a1]\n\t"
                 "v_fma_f64 %[a2], -%[l], s[84:85], %[a2]\n\tv_fma_f64 %[a3], -%[l], s[86:87], %[a3]\n\t"
                 "v_fma_f64 %[b0], -%[l2], s[80:81], %[b0]\n\tv_fma_f64 %[b1], -%[l2], s[82:83], %[b1]\n\t"
                 "v_fma_f64 %[b2], -%[l2], s[84:85], %[b2]\n\tv_fma_f64 %[b3], -%[l2], s[86:87], %[b3]\n\t"
                 : [a0] "+v"(a[C0]), [a1] "+v"(a[C0 + 1]), [a2] "+v"(a[C0 + 2]), [a3] "+v"(a[C0 + 3]),
                   [b0] "+v"(a2[C0]), [b1] "+v"(a2[C0 + 1]), [b2] "+v"(a2[C0 + 2]), [b3] "+v"(a2[C0 + 3])
                 : [l] "v"(l), [l2] "v"(l2), [lo] "v"(lo), [hi] "v"(hi), [c0] "n"(C0), [c1] "n"(C0 + 1), [c2] "n"(C0 + 2), [c3] "n"(C0 + 3)
                 : "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87");
  } else {
    asm volatile(GPBO_RL(80, c0) GPBO_RL(82, c1) GPBO_RL(84, c2) GPBO_RL(86, c3)
                 "v_fma_f64 %[a0], -%[l], s[80:81], %[a0]\n\tv_fma_f64 %[a1], -%[l], s[82:83], %[a1]\n\t"
                 "v_fma_f64 %[a2], -%[l], s[84:85], %[a2]\n\tv_fma_f64 %[a3], -%[l], s[86:87], %[a3]\n\t"
                 : [a0] "+v"(a[C0]), [a1] "+v"(a[C0 + 1]), [a2] "+v"(a[C0 + 2]), [a3] "+v"(a[C0 + 3])
                 : [l] "v"(l), [lo] "v"(lo), [hi] "v"(hi), [c0] "n"(C0), [c1] "n"(C0 + 1), [c2] "n"(C0 + 2), [c3] "n"(C0 + 3)
                 : "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87");
  }
}
template <int C0, bool FOLLOW>
__device__ __forceinline__ void bcast_fma2(double* a, double* a2, const double l, const double l2) {
  int lo, hi;
  split64(l, lo, hi);
  if constexpr (FOLLOW) {
    asm volatile(GPBO_RL(80, c0) GPBO_RL(82, c1) "s_nop 0\n\t"
                 "v_fma_f64 %[a0], -%[l], s[80:81], %[a0]\n\tv_fma_f64 %[a1], -%[l], s[82:83], %[a1]\n\t"
                 "v_fma_f64 %[b0], -%[l2], s[80:81], %[b0]\n\tv_fma_f64 %[b1], -%[l2], s[82:83], %[b1]\n\t"
                 : [a0] "+v"(a[C0]), [a1] "+v"(a[C0 + 1]), [b0] "+v"(a2[C0]), [b1] "+v"(a2[C0 + 1])
                 : [l] "v"(l), [l2] "v"(l2), [lo] "v"(lo), [hi] "v"(hi), [c0] "n"(C0), [c1] "n"(C0 + 1)
                 : "s80", "s81", "s82", "s83");
  } else {
    asm volatile(GPBO_RL(80, c0) GPBO_RL(82, c1) "s_nop 0\n\t"
                 "v_fma_f64 %[a0], -%[l], s[80:81], %[a0]\n\tv_fma_f64 %[a1], -%[l], s[82:83], %[a1]\n\t"
                 : [a0] "+v"(a[C0]), [a1] "+v"(a[C0 + 1])
                 : [l] "v"(l), [lo] "v"(lo), [hi] "v"(hi), [c0] "n"(C0), [c1] "n"(C0 + 1)
                 : "s80", "s81", "s82", "s83");
  }
}
template <int C0, bool FOLLOW>
__device__ __forceinline__ void bcast_fma1(double* a, double* a2, const double l, const double l2) {
  int lo, hi;
  split64(l, lo, hi);
  if constexpr (FOLLOW) {
    asm volatile(GPBO_RL(80, c0) "s_nop 1\n\t"
                 "v_fma_f64 %[a0], -%[l], s[80:81], %[a0]\n\tv_fma_f64 %[b0], -%[l2], s[80:81], %[b0]\n\t"
                 : [a0] "+v"(a[C0]), [b0] "+v"(a2[C0])
                 : [l] "v"(l), [l2] "v"(l2), [lo] "v"(lo), [hi] "v"(hi), [c0] "n"(C0)
                 : "s80", "s81");
  } else {
    asm volatile(GPBO_RL(80, c0) "s_nop 1\n\t"
                 "v_fma_f64 %[a0], -%[l], s[80:81], %[a0]\n\t"
                 : [a0] "+v"(a[C0])
                 : [l] "v"(l), [lo] "v"(lo), [hi] "v"(hi), [c0] "n"(C0)
                 : "s80", "s81");
  }
}
// columns C0 .. 7 of the wave's own block, nearest first
template <int C0, bool FOLLOW>
__device__ __forceinline__ void bcast_fma_from(double* a, double* a2, const double l, const double l2) {
  if constexpr (C0 + 4 <= 8) {
    bcast_fma4<C0, FOLLOW>(a, a2, l, l2);
    bcast_fma_from<C0 + 4, FOLLOW>(a, a2, l, l2);
  } else if constexpr (C0 + 2 <= 8) {
    bcast_fma2<C0, FOLLOW>(a, a2, l, l2);
    bcast_fma_from<C0 + 2, FOLLOW>(a, a2, l, l2);
  } else if constexpr (C0 + 1 <= 8) {
    bcast_fma1<C0, FOLLOW>(a, a2, l, l2);
  }
}

// What one step costs (scripts/archive/r03_latency_probe.py, one wave on its SIMD): every plain VALU instruction — v_fma_f64
// dependent or not, v_readlane_b32, v_mov — occupies the wave for 4 cycles, v_rsq_f64 16, a v_writelane_b32 (what an
// SGPR spill turns into) 28, a ds_write2_b64 ~18; an LDS write -> read round trip is ~90 cycles and a broadcast
// ds_read_b128 holds the LDS pipeline ~8 cycles.  The chain of one column (2 v_readlane of the pivot, v_rsq_f64, two
// Newton steps, the scaling, the store, the next pivot) is ~100 cycles; everything else a wave issues stands in front
// of the next column's chain, so a column is priced in INSTRUCTIONS on the owning wave and in LDS time for everybody
// else.  Eight waves of eight columns: per column the owner spends ~100 + 3.5 x (2 readlane + 2 fma) and each wave
// to its right 4 broadcast reads + 16 fma.
//
// One block (8 columns, wave w) of the right-looking factorisation of a 64-column panel held row-per-lane: thread
// (row i, wave w) keeps a[0..7] = A[i][8w .. 8w+7], and a second row (i + 64, the block below the diagonal one) rides
// along in a2 — the same multipliers, no pivots of its own (that is the panel solve of block row 1, L10 = A10 L00^-T,
// done by substitution in the shadow of the factorisation; the caller passes zeros when there is no such row).  The
// wave first applies the columns left of its own as their owners publish them, then factors its 8 columns inside the
// wave, publishing each column the moment it is final.  Every element receives its rank-1 updates in column order
// whatever the timing, so the result is deterministic.  The wave's rows are ROTATED, row i = (lane + 8 w) mod 64, so
// that the rows of its own diagonal 8x8 block sit in lanes 0..7 and every v_readlane names its lane by a constant.
//
// Column images: Lc[j * DS + i] = L[i][j] for i < 64, Lc2[j * DS + i] = L[64 + i][j], and Lc[j * DS + 64] = 1 / L[j][j]
// — the reciprocal pivot doubles as the "column j is complete (both images)" marker (zero-initialised; written by ALL
// lanes right behind the two column stores, slots 64..127, no exec mask: the overflow lands in the next column's rows
// 0..47, which are written later and read only after that).  Elements above the diagonal are NOT zeroed on the way
// (they never reach the lower triangle); the caller zeroes the registers before the global store.  A non-positive
// pivot gives a non-finite reciprocal (v_rsq_f64 of <= 0) that spreads to everything behind it; the caller finds the
// first one afterwards (LAPACK's info) — no test on the chain.
template <bool FOLLOW>
// (No __restrict__ on the images, nor on the shared-memory base they are carved from: other waves write what this one reads, and
// the compiler barriers (GPBO_LDS_ORDER) only bind accesses the optimiser cannot prove private.  With restrict-qualified images the
// marker poll was, in one inlining context of round 4, taken for loop-invariant: the loop became bare s_sleep.  A `volatile` poll
// through the generic pointer is no alternative: it compiles to flat_load ... sc0 sc1.)
__device__ __forceinline__ void factor_block8(double (&a)[8], double (&a2)[8], double* Lc, double* Lc2,
                                              int* broken, const int i, const int w, long long* stamp = nullptr) {
  {
    // Catch-up, two columns per turn (columns are published in order: the marker of column k + 1 vouches for k and
    // k + 1).  LDS time is what the waves compete for — a waiting wave that keeps re-reading slows the chain wave's
    // stores — so a waiting wave reads ONE word, the marker, and sleeps until the owner's s_wakeup (sent behind every
    // second column) or the sleep's own end; the data is read once, after the marker.
    // Measured and dropped (scripts/archive/r03_col_stamps.py, 64 columns + riding rows = 19 000 cycles with this loop): the next
    // turn's reads issued before this turn's arithmetic (two register sets) 22 500; one column per turn 22 000; the
    // multipliers by v_readlane from the rows just read (a third of the LDS traffic, twice the VALU work) 22 800 — every
    // variant that makes the waiting waves faster makes the owner slower, through LDS time or through the SIMD the
    // owner shares with one of them.
    const double* prow0 = Lc + 8 * w;      // L[8w + cc][k] = prow0[k * DS + cc]: the same address for every lane
    for (int k = 0; k < 8 * w; k += 2) {
      int spins = 0;
      // Waves w and w - 4 share a SIMD: while w - 4 owns the chain this wave stays asleep (its fmas would take the fp64 pipe
      // from under the chain) and applies that block's eight columns in one go once the block is complete — it owns the chain
      // four blocks later, there is time.  (GPBO_CHOL_DEFER_MATE=0: experiment builds without it.)
      const int kw = (GPBO_CHOL_DEFER_MATE && w >= 4 && (k >> 3) == w - 4) ? 8 * (w - 4) + 7 : k + 1;
      while (Lc[kw * DS + 64] == 0.0) {
        if (++spins > (1 << 16)) {   // (x 512 cycles = 14 ms) cannot happen while the owner wave runs; never hang the GPU on a bug
          if (i == 0) *broken = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(GPBO_CHOL_POLL_SLEEP);
        GPBO_LDS_ORDER();
      }
      GPBO_LDS_ORDER();
      double li[2], li2[2], p[2][8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        li[u] = Lc[(k + u) * DS + i];
        li2[u] = FOLLOW ? Lc2[(k + u) * DS + i] : 0.0;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) p[u][cc] = prow0[(k + u) * DS + cc];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          a[cc] = fma(-li[u], p[u][cc], a[cc]);
          if (FOLLOW) a2[cc] = fma(-li2[u], p[u][cc], a2[cc]);
        }
    }
  }
  __builtin_amdgcn_s_setprio(3);     // the chain: where two waves share a SIMD the arbiter should pick this one
  double pivsrc = a[0];        // lane jj of this holds the pivot of the wave's next column
  double* col = Lc + 8 * w * DS + i;        // column 8w + jj of the image: col[jj * DS]
  double* col2 = Lc2 + 8 * w * DS + i;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const double piv = readlane_f64(pivsrc, jj);
    // 1/sqrt(piv): v_rsq_f64 seed (2^-23) + two Newton steps y <- y + y (1/2 - (piv/2) y^2); the column is scaled by
    // the reciprocal (as LAPACK's dpotf2 does) — the diagonal element too (piv * rs)
    const double h = 0.5 * piv;
    double rs = __builtin_amdgcn_rsq(piv);
    rs = fma(rs, fma(-h, rs * rs, 0.5), rs);
    rs = fma(rs, fma(-h, rs * rs, 0.5), rs);
    const double l = a[jj] * rs;
    // the NEXT pivot first: in lane jj + 1 the rank-1 update of element (j+1, j+1) is fma(-l, l, .) — the same bits the
    // general update produces there (all updates of earlier columns are already in a[jj + 1]) — so the chain does not
    // wait for the broadcast of l
    if (jj < 7) pivsrc = fma(-l, l, a[jj + 1]);
    a[jj] = l;
    col[jj * DS] = l;
    double l2 = 0.0;
    if (FOLLOW) {
      l2 = a2[jj] * rs;
      a2[jj] = l2;
      col2[jj * DS] = l2;
    }
    GPBO_LDS_ORDER();
    col[jj * DS + 64] = rs;
    GPBO_LDS_ORDER();
    if ((jj & GPBO_CHOL_WAKE_MASK) == GPBO_CHOL_WAKE_MASK) asm volatile("s_wakeup" ::: "memory");      // the store is in the LDS queue before any reader woken by this can queue its read
    GPBO_SCHED_FENCE();
    switch (jj) {     // jj is a compile-time constant of the unrolled loop: only its own case survives
      case 0: bcast_fma_from<1, FOLLOW>(a, a2, l, l2); break;
      case 1: bcast_fma_from<2, FOLLOW>(a, a2, l, l2); break;
      case 2: bcast_fma_from<3, FOLLOW>(a, a2, l, l2); break;
      case 3: bcast_fma_from<4, FOLLOW>(a, a2, l, l2); break;
      case 4: bcast_fma_from<5, FOLLOW>(a, a2, l, l2); break;
      case 5: bcast_fma_from<6, FOLLOW>(a, a2, l, l2); break;
      case 6: bcast_fma_from<7, FOLLOW>(a, a2, l, l2); break;
      default: break;
    }
    GPBO_SCHED_FENCE();
  }
  __builtin_amdgcn_s_setprio(0);
  if (stamp && i == 0) stamp[w] = clock64();
}

// After a factorisation: info (1-based column within the image, 0 = fine) = the first column whose reciprocal pivot is
// not a positive finite number.  One wave; lane j looks at column j.
__device__ __forceinline__ int first_bad_column(const double* __restrict__ Lc, const int lane) {
  const double r = Lc[lane * DS + 64];
  const bool bad = !(r > 0.0 && r < 1.7976931348623157e308);
  const unsigned long long mask = __ballot(bad);
  return mask ? (int)__ffsll((long long)mask) : 0;
}

// The four 16x16 diagonal blocks of a 64x64 lower factor (column-major image Lc with its reciprocal pivots) inverted by ONE
// wave: lane (b = lane >> 4, c = lane & 15) runs the forward substitution for column c of inv(L_bb) and parks it
// k-major in the wave's own tile set, Dk[b][k = c][m] with stride 17 (the layout the MFMA A-fragment reads).
// Only the blocks b >= bmin are inverted (column block C of the 64x64 inverse needs D_C .. D_3).
// Column sweep, written as a pipeline (round 4): the 16 reciprocal pivots up front, column k + 1 of the block on its way while
// column k is applied — every step is one multiply and 15 - k independent fmas.  (Left to itself the compiler turned the sweep
// into a row-by-row form whose rows are chains of up to 15 dependent fmas behind LDS waits: 3 900 cycles; this form 3 400.
// Computing the four inverses once per factor and sharing them behind a barrier was no faster — 4 000 cycles with two waves at
// work: the sweep is bound by its own 16 steps, not by the eight waves' LDS traffic.)
__device__ __forceinline__ void diag16_inverses(const double* __restrict__ Lc, double* __restrict__ Dk, const int lane, const int bmin) {
  const int b = lane >> 4, c = lane & 15;
  if (b < bmin) return;
  const double* blk = Lc + (16 * b) * DS + 16 * b;      // L_bb[r][k] = blk[k * DS + r]
  double w[16], rs[16], cur[16], nxt[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    w[r] = (r == c) ? 1.0 : 0.0;
    rs[r] = Lc[(16 * b + r) * DS + 64];
  }
#pragma unroll
  for (int r = 1; r < 16; ++r) cur[r] = blk[r];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k + 1 < 16) {
#pragma unroll
      for (int r = k + 2; r < 16; ++r) nxt[r] = blk[(k + 1) * DS + r];
    }
    __builtin_amdgcn_sched_barrier(0);
    const double wk = w[k] * rs[k];
    w[k] = wk;
#pragma unroll
    for (int r = k + 1; r < 16; ++r) w[r] = fma(-cur[r], wk, w[r]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = k + 2; r < 16; ++r) cur[r] = nxt[r];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) Dk[b * 272 + c * 17 + r] = w[r];
}

// Column block C (16 columns) of W = L^-1 for a 64x64 lower factor, by ONE wave and without a barrier: blocked forward
// substitution  X_C = D_C,  X_r = -D_r sum_{t=C..r-1} L_rt X_t  (D_r = inv(L_rr)), every product a chain of
// v_mfma_f64_16x16x4_f64 whose accumulator (rows (lane>>4) + 4 reg, column lane & 15) IS the B fragment of the next
// product — nothing moves between the steps.  Written row-major to Wout[64][64] (zeros above the diagonal block).
template <int C>
__device__ __forceinline__ void inverse_colblock(const double* __restrict__ Lc, const double* __restrict__ Dk, double* __restrict__ Wout,
                                                 const int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  d4 X[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) X[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) X[C][rr] = Dk[C * 272 + lr * 17 + lk + 4 * rr];
#pragma unroll
  for (int r = C + 1; r < 4; ++r) {
    d4 T0 = d4{0.0, 0.0, 0.0, 0.0}, T1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = C; t < r; ++t) {
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        T0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lc[(16 * t + 4 * g + lk) * DS + 16 * r + lr], X[t][g], T0, 0, 0, 0);
        T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lc[(16 * t + 4 * (g + 1) + lk) * DS + 16 * r + lr], X[t][g + 1], T1, 0, 0, 0);
      }
    }
    const d4 T = T0 + T1;
    d4 Y0 = d4{0.0, 0.0, 0.0, 0.0}, Y1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      Y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Dk[r * 272 + (4 * g + lk) * 17 + lr], T[g], Y0, 0, 0, 0);
      Y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Dk[r * 272 + (4 * (g + 1) + lk) * 17 + lr], T[g + 1], Y1, 0, 0, 0);
    }
    X[r] = -(Y0 + Y1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Wout[(16 * r + lk + 4 * rr) * 64 + 16 * C + lr] = X[r][rr];
}

__device__ __forceinline__ void inverse_colblock_of(const double* Lc, const double* Dk, double* Wout, const int q, const int lane) {
  switch (q) {
    case 0: inverse_colblock<0>(Lc, Dk, Wout, lane); break;
    case 1: inverse_colblock<1>(Lc, Dk, Wout, lane); break;
    case 2: inverse_colblock<2>(Lc, Dk, Wout, lane); break;
    default: inverse_colblock<3>(Lc, Dk, Wout, lane); break;
  }
}

// Diagonal block of `nblk` (1 or 2) 64-blocks starting at block kb, all earlier updates applied: factor in place, write
// inv(L_kk) (and inv(L_kk+1)) to dinv.  512 threads = 8 waves, no launch inside, seven workgroup barriers.  Wave w,
// thread = row i (rotated, see factor_block8):
//   columns 8w .. 8w+7 of A00 with the same columns of A10 (row 64 + i) riding along  ->  L00, L10
//   SYRK  A11 -= L10 L10^T  (MFMA out of the L10 image)
//   columns 8w .. 8w+7 of A11  ->  L11
//   waves 0-3: inv(L00), waves 4-7: inv(L11), one 16-column block each (the second in reverse order: SIMD balance).
__device__ __forceinline__ void diag128_body(double* __restrict__ L, const int64_t ld, const int kb, const int nblk,
                                             double* __restrict__ dinv, int* __restrict__ info, double* smem,
                                             long long* __restrict__ stamps, const int tid) {
  double* Lc0 = smem + C128_LC0;
  double* LcX = smem + C128_LCX;
  double* Wr = smem + C128_WR;
  int* flags = reinterpret_cast<int*>(smem + C128_FLAGS);
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = (lane + 8 * w) & 63;        // the wave's diagonal 8x8 block in lanes 0..7; rows are what every address is computed from
  const bool two = nblk == 2;
  double* A = L + (int64_t)kb * 64 * ld + (int64_t)kb * 64;
  if (stamps && tid == 0) stamps[0] = clock64();
  if (tid == 0) flags[0] = 0;
  double a[8], a2[8], b[8];     // rows of A00, A10, A11
  {
    // The block comes in through LDS (round 4).  Read row-per-lane straight from memory — what the factorisation wants — every
    // 16-byte load of a wave touches 64 different cache lines, and the CU's one address unit needs ~64 cycles for each of the 96
    // of them: 7 000 cycles before the first column.  Read ROW-WISE (a wave instruction = 1 KiB of one or two rows), parked in a
    // staging image with an odd row stride and fetched back row-per-lane (conflict-free: lane i sits 2 banks behind lane i - 1),
    // the same 96 KiB take about half of that.  The staging image lies over the column images and the exchange area, all unused
    // so far; their markers are zeroed after it has been read.
    constexpr int SS = 129;                                   // staging row stride (doubles)
    static_assert(128 * SS <= C128_FLAGS, "the staging image must end before the flag words");
    double* S = smem;
    const int half = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                             // rows 0..63, columns 0..63: two rows per wave instruction
      const int r = 8 * w + 2 * q + half;
      const double2 v = *reinterpret_cast<const double2*>(A + (int64_t)r * ld + 2 * l32);
      S[r * SS + 2 * l32] = v.x;
      S[r * SS + 2 * l32 + 1] = v.y;
    }
    if (two) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {                           // rows 64..127, columns 0..127: one row per wave instruction
        const int r = 64 + 8 * w + q;
        const double2 v = *reinterpret_cast<const double2*>(A + (int64_t)r * ld + 2 * lane);
        S[r * SS + 2 * lane] = v.x;
        S[r * SS + 2 * lane + 1] = v.y;
      }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 8; ++h) a[h] = S[i * SS + 8 * w + h];
    if (two) {
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        a2[h] = S[(64 + i) * SS + 8 * w + h];
        b[h] = S[(64 + i) * SS + 64 + 8 * w + h];
      }
    } else {
#pragma unroll
      for (int h = 0; h < 8; ++h) a2[h] = b[h] = 0.0;     // a zero row rides along
    }
    __syncthreads();
  }
  if (tid < 64) {          // column markers: nothing published yet
    Lc0[tid * DS + 64] = 0.0;
    LcX[tid * DS + 64] = 0.0;
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[1] = clock64();
  // ---- columns 0..63: L00, and L10 = A10 L00^-T in its shadow
  factor_block8<true>(a, a2, Lc0, LcX, &flags[0], i, w, stamps ? stamps + 7 : nullptr);
  {
    // the wave's 8 columns are final: rows straight from registers (64 contiguous bytes per thread, zeros above the diagonal)
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) a[cc] = (i >= 8 * w + cc) ? a[cc] : 0.0;
    double2* d0 = reinterpret_cast<double2*>(A + (int64_t)i * ld + 8 * w);
#pragma unroll
    for (int h = 0; h < 4; ++h) d0[h] = make_double2(a[2 * h], a[2 * h + 1]);
    if (two) {
      double2* d1 = reinterpret_cast<double2*>(A + (int64_t)(64 + i) * ld + 8 * w);
#pragma unroll
      for (int h = 0; h < 4; ++h) d1[h] = make_double2(a2[2 * h], a2[2 * h + 1]);
    }
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[2] = clock64();
  const int lr = lane & 15, lk = lane >> 4;
  if (two) {
    // ---- SYRK: U = L10 L10^T, the ten lower 16x16 tiles over the eight waves, operands out of the LcX image
    for (int t = w; t < 10; t += 8) {             // linear lower index: ti (ti + 1) / 2 + tj
      const int ti = (t >= 6) ? 3 : (t >= 3) ? 2 : (t >= 1) ? 1 : 0;
      const int tj = t - ti * (ti + 1) / 2;
      d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int g = 0; g < 16; g += 2) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(LcX[(4 * g + lk) * DS + 16 * ti + lr], LcX[(4 * g + lk) * DS + 16 * tj + lr], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(LcX[(4 * g + 4 + lk) * DS + 16 * ti + lr], LcX[(4 * g + 4 + lk) * DS + 16 * tj + lr], acc1, 0, 0, 0);
      }
      const d4 acc = acc0 + acc1;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Wr[(16 * ti + lk + 4 * rr) * 81 + 16 * tj + lr] = acc[rr];
    }
    __syncthreads();
    if (stamps && tid == 0) stamps[3] = clock64();
    // elements above the diagonal pick up whatever the exchange area holds: they never reach the lower triangle
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) b[cc] -= Wr[i * 81 + 8 * w + cc];
    if (tid < 64) LcX[tid * DS + 64] = 0.0;      // the L10 image makes room for L11: its markers start over
    __syncthreads();   // the exchange area is re-used for the waves' diagonal inverses
    // ---- columns 64..127: L11
    factor_block8<false>(b, a2, LcX, LcX, &flags[0], i, w);
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) b[cc] = (i >= 8 * w + cc) ? b[cc] : 0.0;
    double2* d2 = reinterpret_cast<double2*>(A + (int64_t)(64 + i) * ld + 64 + 8 * w);
#pragma unroll
    for (int h = 0; h < 4; ++h) d2[h] = make_double2(b[2 * h], b[2 * h + 1]);
    if (stamps && tid == 0) stamps[4] = clock64();
    __syncthreads();
    if (stamps && tid == 0) stamps[5] = clock64();
    // waves w and w + 4 share a SIMD and column block C costs 36 / 20 / 8 / 0 MFMAs for C = 0..3: the second inverse hands its
    // blocks out in reverse, so every SIMD carries 36 or 28 of the 128 instead of 72 / 40 / 16 / 0 (round 4: 11 000 -> 8 000 cycles)
    if (w < 4) {
      diag16_inverses(Lc0, Wr + w * 4 * 272, lane, w);
      if (stamps && tid == 0) stamps[15] = clock64();
      GPBO_LDS_ORDER();   // the wave reads back its own tiles: same-wave LDS accesses are performed in order
      inverse_colblock_of(Lc0, Wr + w * 4 * 272, dinv + (int64_t)kb * 4096, w, lane);
    } else {
      diag16_inverses(LcX, Wr + w * 4 * 272, lane, 7 - w);
      GPBO_LDS_ORDER();
      inverse_colblock_of(LcX, Wr + w * 4 * 272, dinv + (int64_t)(kb + 1) * 4096, 7 - w, lane);
    }
  } else {
    if (w < 4) {
      diag16_inverses(Lc0, Wr + w * 4 * 272, lane, w);
      GPBO_LDS_ORDER();
      inverse_colblock_of(Lc0, Wr + w * 4 * 272, dinv + (int64_t)kb * 4096, w, lane);
    }
  }
  if (w == 0) {
    // LAPACK potrf: order of the first non-positive leading minor (the images are complete: the inverses started behind
    // a barrier)
    int bad = first_bad_column(Lc0, lane);
    if (two && bad == 0) {
      const int bad2 = first_bad_column(LcX, lane);
      bad = bad2 ? 64 + bad2 : 0;
    }
    if (tid == 0) {
      if (flags[0] && *info == 0) *info = -1 - kb;                  // broken hand-off (never seen): surfaces as an error
      else if (bad && *info == 0) *info = kb * 64 + bad;
      if (stamps) stamps[6] = clock64();
    }
  }
}


// Panel solve below a 128-column diagonal block, in place: X = A inv(L_blk)^T with L_blk = [[L00, 0], [L10, L11]], i.e.
//   X0 = A0 W00^T,   X1 = (A1 - X0 L10^T) W11^T        (W00, W11 = the 64x64 inverses from the diagonal workgroup).
// One 256-thread group = 16 panel rows (row block `blk` below the diagonal block), worked on TRANSPOSED (X^T = W A^T): the
// accumulator of each product is the B fragment of the next, so the three products chain through 8 KiB of LDS exchange only; wave w
// owns column tile w (tile 3 - w in the last product, which balances the triangular k-ranges).  All operand fragments come straight
// from L2 (the 96 KiB of W00 / L10 / W11 are shared by every workgroup of the launch) and are requested up front.
// `lds`: C128_PANEL_LDS_DOUBLES doubles of the group's own; `tid` = the thread's index within the group; two barriers.
constexpr int C128_PANEL_LDS_DOUBLES = 2 * 4 * 4 * 64;
__device__ __forceinline__ void chol128_panel_body(double* L, const int64_t ld, const int kb, const double* dinv, const int blk, double* lds,
                                                   const int tid) {
  typedef double (*frag_t)[4][64];
  frag_t Xs = reinterpret_cast<frag_t>(lds);                 // [k tile][k group][lane]: B fragments of X0^T
  frag_t Ts = reinterpret_cast<frag_t>(lds + 4 * 4 * 64);    // ... of T^T
  const int lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t c0 = (int64_t)kb * 64;
  double* Arow = L + (c0 + 128 + (int64_t)blk * 16) * ld + c0;       // A[n][k] = Arow[n * ld + k]
  const double* W00 = dinv + (int64_t)kb * 4096;
  const double* W11 = dinv + (int64_t)(kb + 1) * 4096;
  const double* L10 = L + (c0 + 64) * ld + c0;
  const int tj = w, tj3 = 3 - w;
  // ---- everything this wave will multiply, requested now (addresses do not depend on results)
  double bq[4][4], w0[4][4], l1[4][4], w1[4][4], cin[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bq[kt][g] = (kt <= tj) ? Arow[(int64_t)lr * ld + 16 * kt + 4 * g + lk] : 0.0;
      w0[kt][g] = (kt <= tj) ? W00[(16 * tj + lr) * 64 + 16 * kt + 4 * g + lk] : 0.0;
      l1[kt][g] = L10[(int64_t)(16 * tj + lr) * ld + 16 * kt + 4 * g + lk];
      w1[kt][g] = (kt <= tj3) ? W11[(16 * tj3 + lr) * 64 + 16 * kt + 4 * g + lk] : 0.0;
    }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) cin[rr] = Arow[(int64_t)lr * ld + 64 + 16 * tj + lk + 4 * rr];
  // ---- X0^T tile tj = sum_{kt <= tj} W00[tj][kt] A0^T[kt]
  d4 x0, x1;
  {
    d4 p0 = d4{0.0, 0.0, 0.0, 0.0}, p1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
      if (kt <= tj) {
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0[kt][g], bq[kt][g], p0, 0, 0, 0);
          p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0[kt][g + 1], bq[kt][g + 1], p1, 0, 0, 0);
        }
      }
    x0 = p0 + p1;
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) Xs[tj][rr][lane] = x0[rr];
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) Arow[(int64_t)lr * ld + 16 * tj + lk + 4 * rr] = x0[rr];   // every wave has read A0 by now
  // ---- T^T tile tj = A1^T[tj] - sum_kt L10[tj][kt] X0^T[kt]
  {
    d4 p0 = d4{0.0, 0.0, 0.0, 0.0}, p1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(l1[kt][g], Xs[kt][g][lane], p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(l1[kt][g + 1], Xs[kt][g + 1][lane], p1, 0, 0, 0);
      }
    const d4 s = p0 + p1;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Ts[tj][rr][lane] = cin[rr] - s[rr];
  }
  __syncthreads();
  // ---- X1^T tile tj3 = sum_{kt <= tj3} W11[tj3][kt] T^T[kt]
  {
    d4 p0 = d4{0.0, 0.0, 0.0, 0.0}, p1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
      if (kt <= tj3) {
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1[kt][g], Ts[kt][g][lane], p0, 0, 0, 0);
          p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1[kt][g + 1], Ts[kt][g + 1][lane], p1, 0, 0, 0);
        }
      }
    x1 = p0 + p1;
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) Arow[(int64_t)lr * ld + 64 + 16 * tj3 + lk + 4 * rr] = x1[rr];
}

// The NEXT diagonal block (nt x nt 16x16 tiles, nt = 8 or 4) brought up to date right after the panel solve:
// C -= X X^T with X = the solved panel rows of that block (k = 128).  One 16x16 tile (linear lower index t) per 256-thread group,
// k split over the four waves, partial sums combined through LDS in a fixed order — 36 small groups instead of a 128^3 product on
// the diagonal workgroup's single CU (8.5 us there).  `lds`: C128_UPD_LDS_DOUBLES doubles; one barrier.
constexpr int C128_UPD_LDS_DOUBLES = 4 * 4 * 64;
__device__ __forceinline__ void chol128_diag_update_body(double* L, const int64_t ld, const int kb, const int t, double* lds, const int tid) {
  typedef double (*part_t)[4][64];
  part_t Ps = reinterpret_cast<part_t>(lds);
  int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  while (ti * (ti + 1) / 2 > t) --ti;
  const int tj = t - ti * (ti + 1) / 2;
  const int lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t c0 = (int64_t)kb * 64;
  const double* X = L + (c0 + 128) * ld + c0;
  double* C = L + (c0 + 128) * ld + (c0 + 128);
  // k = 32 w + 8 h + 2 lk + e for step (h, e): any assignment of the four k's of a step to the four lane groups works as
  // long as A and B agree, and this one lets a lane fetch its operands with 16-byte loads
  double2 av[4], bv[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    av[h] = *reinterpret_cast<const double2*>(X + (int64_t)(16 * ti + lr) * ld + 32 * w + 8 * h + 2 * lk);
    bv[h] = *reinterpret_cast<const double2*>(X + (int64_t)(16 * tj + lr) * ld + 32 * w + 8 * h + 2 * lk);
  }
  d4 p0 = d4{0.0, 0.0, 0.0, 0.0}, p1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h].x, bv[h].x, p0, 0, 0, 0);
    p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h].y, bv[h].y, p1, 0, 0, 0);
  }
  const d4 p = p0 + p1;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) Ps[w][rr][lane] = p[rr];
  __syncthreads();
  const double s = ((Ps[0][w][lane] + Ps[1][w][lane]) + Ps[2][w][lane]) + Ps[3][w][lane];
  double* cp = C + (int64_t)(16 * ti + lk + 4 * w) * ld + 16 * tj + lr;
  *cp -= s;
}

}  // namespace gpbo
