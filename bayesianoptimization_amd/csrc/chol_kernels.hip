// Blocked Cholesky, round-3 schedule (gfx950): 128-column steps inside 512-column outer panels.
//
// What it replaces: cholesky(K, lower=True) -> LAPACK dpotrf   (sklearn _gpr.py:349, via scipy.linalg.cholesky)
//
// Why 128: the factorisation is a latency chain, not a flop problem (round 2: 0.68 us per column at every N, the
// matrix pipe 2-5 % busy on the chain kernels).  fp64 runs at 32 flop/clk/SIMD on VALU and MFMA alike, so ONE compute
// unit needs 1.7 us for a 64^3 product: whatever sits on the chain must be small, and everything else must be wide.
// Per 128 columns the chain is now
//     diagonal block (ONE workgroup, no launch inside: potf2 of 128x64 -> SYRK -> potf2 of 64x64 -> the two 64x64 inverses)
//  -> panel solve of ALL rows below (16 rows per workgroup, three chained 16-row MFMA products, no explicit 128x128 inverse)
//  -> rank-128 update of the NEXT diagonal block only (36 small workgroups)
//  -> [next diagonal block || rest of the in-panel update] in one launch
// i.e. three launches per 128 columns where round 2 had two per 64, and the diagonal workgroup no longer applies the
// previous column block's update to its own block by itself (that 64^3 product cost 1.7 us per step on one CU; for a
// 128-block it would be 8.5 us).  The rank-512 trailing update per outer panel is unchanged (gemm128_f64_kernel).
//
// Every kernel here takes a lane index (blockIdx.y / .z) so that gpbo_lml_batch's theta lanes share the launches.
#include <cstdlib>

#include "gemm_tile.h"
#include "gpbo_internal.h"

namespace gpbo {

// ---- LDS layout of the diagonal-block workgroup (doubles) --------------------------------------------------------
constexpr int DS = 80;                                   // stride of the column-major images: a fragment's four k-rows fall 32 banks apart
constexpr int C128_IMG = 64 * DS + 64;                   // one image: 64 columns + room for the last column's marker overflow (factor_quarter)
constexpr int C128_LC0 = 0;                              // L00:  Lc0[j * DS + i] = L[i][j], Lc0[j * DS + 64] = 1 / L[j][j]
constexpr int C128_LCX = C128_IMG;                       // L10 during the first factorisation and the SYRK, then L11
constexpr int C128_WR = 2 * C128_IMG;                    // SYRK exchange [64][81], then the waves' 16x16 diagonal inverses [4][4][272]
constexpr int C128_WR_DOUBLES = 5248;
constexpr int C128_FLAGS = C128_WR + C128_WR_DOUBLES;    // ints: [0] broken hand-off
constexpr int C128_LDS_DOUBLES = C128_FLAGS + 8;
constexpr size_t C128_LDS_BYTES = (size_t)C128_LDS_DOUBLES * sizeof(double);
static_assert(C128_WR_DOUBLES >= 64 * 81 && C128_WR_DOUBLES >= 4 * 4 * 272, "exchange area too small");
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

// What one step costs (scripts/r03_latency_probe.py, one wave on its SIMD): EVERY VALU instruction — v_fma_f64 dependent
// or not, v_readlane_b32, v_cndmask, v_mov — occupies the wave for 4 cycles, v_rsq_f64 16, a dependent v_writelane_b32
// (what an SGPR spill turns into) 28; an LDS write -> read round trip is ~90 cycles.  So the factorisation of a column
// is priced in INSTRUCTIONS: the first version of this routine spent ~110 per column (selects for the diagonal / the
// zero upper part, an exec-masked pivot/counter store, a per-column pivot test, 2 v_readlane per broadcast multiplier,
// spilled SGPRs) = 720 cycles per column; this one ~35.
//
// One quarter (16 columns, wave q) of the right-looking factorisation of a 64-column panel held row-per-lane:
// thread (row i, quarter q) keeps a[0..15] = A[i][16q .. 16q+15]; with FOLLOW a second row (i + 64, the block below the
// diagonal one) rides along in a2 — the same multipliers, no pivots of its own (that is the panel solve of block row 1,
// L10 = A10 L00^-T, done by substitution in the shadow of the factorisation).  The wave first applies the columns left
// of its own as their owners publish them, then factors its 16 columns inside the wave, publishing each column the
// moment it is final.  Every element receives its rank-1 updates in column order whatever the timing, so the result is
// deterministic.
//
// Column image: Lc[j * DS + i] = L[i][j] for i < 64, and Lc[j * DS + 64] = 1 / L[j][j] — the reciprocal pivot doubles as
// the "column j is complete" marker (zero-initialised; written by ALL lanes right behind the column, slots 64..127, no
// exec mask: the overflow lands in the next column's rows 0..47, which are written later and read only after that).
// Elements above the diagonal are NOT zeroed on the way (they never reach the lower triangle); the caller zeroes the
// registers before the global store.  A non-positive pivot gives a non-finite reciprocal (v_rsq_f64 of <= 0) that
// spreads to everything behind it; the caller finds the first one afterwards (LAPACK's info) — no test on the chain.
template <bool FOLLOW>
__device__ __forceinline__ void factor_quarter(double (&a)[16], double (&a2)[16], double* __restrict__ Lc, double* __restrict__ Lc2,
                                               int* broken, const int i, const int q, long long* stamp = nullptr) {
  {
    const int need = 16 * q;
    int k = 0, spins = 0;
    const double* prow0 = Lc + 16 * q;      // L[16q + cc][k] = prow0[k * DS + cc]: the same address for every lane
    while (k < need) {
      // columns are published in order: the marker of column k + 1 vouches for k and k + 1
      const double m2 = Lc[(k + 1) * DS + 64];
      GPBO_LDS_ORDER();
      if (m2 != 0.0) {
        double li[2], li2[2], p[2][16];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          li[u] = Lc[(k + u) * DS + i];
          li2[u] = FOLLOW ? Lc2[(k + u) * DS + i] : 0.0;
#pragma unroll
          for (int cc = 0; cc < 16; ++cc) p[u][cc] = prow0[(k + u) * DS + cc];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int cc = 0; cc < 16; ++cc) {
            a[cc] = fma(-li[u], p[u][cc], a[cc]);
            if (FOLLOW) a2[cc] = fma(-li2[u], p[u][cc], a2[cc]);
          }
        k += 2;
        continue;
      }
      const double m1 = Lc[k * DS + 64];
      GPBO_LDS_ORDER();
      if (m1 != 0.0) {
        const double li = Lc[k * DS + i];
        const double li2 = FOLLOW ? Lc2[k * DS + i] : 0.0;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
          const double p = prow0[k * DS + cc];
          a[cc] = fma(-li, p, a[cc]);
          if (FOLLOW) a2[cc] = fma(-li2, p, a2[cc]);
        }
        k += 1;
        continue;
      }
      if (++spins > (1 << 22)) {   // cannot happen while the owner wave runs; never hang the GPU on a bug
        if (i == 0) *broken = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  double pivsrc = a[0];        // lane j of this holds the pivot of the wave's next column
  double pend[16];             // multipliers of the PREVIOUS column fetched from LDS, applied one column later
#pragma unroll
  for (int cc = 0; cc < 16; ++cc) pend[cc] = 0.0;
  double lprev = 0.0, l2prev = 0.0;
  double* col = Lc + 16 * q * DS + i;        // column 16q + jj of the image: col[jj * DS]
  double* col2 = Lc2 + 16 * q * DS + i;
  const double* bro = Lc + 16 * q * DS + 16 * q;   // bro[jj * DS + cc] = L[16q + cc][16q + jj]
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int j = 16 * q + jj;
    const double piv = readlane_f64(pivsrc, j);
    // 1/sqrt(piv): v_rsq_f64 seed (2^-23) + two Newton steps y <- y + y (1/2 - (piv/2) y^2); the column is scaled by
    // the reciprocal (as LAPACK's dpotf2 does) — the diagonal element too (piv * rs)
    const double h = 0.5 * piv;
    double rs = __builtin_amdgcn_rsq(piv);
    rs = fma(rs, fma(-h, rs * rs, 0.5), rs);
    rs = fma(rs, fma(-h, rs * rs, 0.5), rs);
    const double l = a[jj] * rs;
    // the NEXT pivot first: in lane j + 1 the rank-1 update of element (j+1, j+1) is fma(-l, l, .) — the same bits the
    // general update produces there (all updates of earlier columns are already in a[jj + 1]) — so the chain does not
    // wait for the broadcast of l
    if (jj < 15) pivsrc = fma(-l, l, a[jj + 1]);
    a[jj] = l;
    col[jj * DS] = l;
    col[jj * DS + 64] = rs;
    double l2 = 0.0;
    if (FOLLOW) {
      l2 = a2[jj] * rs;
      a2[jj] = l2;
      col2[jj * DS] = l2;
    }
    GPBO_LDS_ORDER();
    GPBO_SCHED_FENCE();
    // (1) the previous column's far multipliers (cc >= jj + 2), requested one column ago: their LDS latency is covered by
    //     the chain above instead of standing in front of it
    if (jj >= 1) {
#pragma unroll
      for (int cc = jj + 2; cc < 16; ++cc) {
        a[cc] = fma(-lprev, pend[cc], a[cc]);
        if (FOLLOW) a2[cc] = fma(-l2prev, pend[cc], a2[cc]);
      }
    }
    // (2) request this column's far multipliers (cc >= jj + 3): broadcast reads of the column just written
#pragma unroll
    for (int cc = jj + 3; cc < 16; ++cc) pend[cc] = bro[jj * DS + cc];
    lprev = l;
    l2prev = l2;
    // (3) the two nearest columns by v_readlane (needed before their turn comes)
#pragma unroll
    for (int cc = jj + 1; cc < 16 && cc <= jj + 2; ++cc) {
      const double lc = readlane_f64(l, 16 * q + cc);   // L[16q + cc][j]
      a[cc] = fma(-l, lc, a[cc]);
      if (FOLLOW) a2[cc] = fma(-l2, lc, a2[cc]);
    }
    GPBO_SCHED_FENCE();
  }
  if (stamp && i == 0 && q == 0) *stamp = clock64();
}

// The riding rows (block row 1 of a two-block diagonal workgroup) as waves of their own: L10 = A10 L00^-T by substitution,
// wave q the columns 16q .. 16q+15 of rows 64 .. 127 (thread = row), consuming the owners' columns as they are published.
// For a column k left of the wave's quarter the update needs L10[i][k] from the follower wave that owns it: the L10
// image Lc2 carries markers of its own (same convention as the L00 image).  In the wave's own quarter the marker read
// returns the reciprocal pivot itself.  Reads of column j + 1 (marker, then multipliers) are issued before column j is
// processed: if the marker was already set they are valid (LDS serves a wave's reads in order), else they are repeated.
__device__ __forceinline__ void follow_quarter(double (&a2)[16], const double* __restrict__ Lc, double* __restrict__ Lc2, int* broken,
                                               const int i, const int q) {
  const double* prow0 = Lc + 16 * q;      // L00[16q + cc][k] = prow0[k * DS + cc]
  {
    const int need = 16 * q;
    int k = 0, spins = 0;
    while (k < need) {
      const double m4 = Lc2[(k + 3) * DS + 64];
      GPBO_LDS_ORDER();
      if (m4 != 0.0) {
        double li[4], p[4][16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          li[u] = Lc2[(k + u) * DS + i];
#pragma unroll
          for (int cc = 0; cc < 16; ++cc) p[u][cc] = prow0[(k + u) * DS + cc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int cc = 0; cc < 16; ++cc) a2[cc] = fma(-li[u], p[u][cc], a2[cc]);
        k += 4;
        continue;
      }
      const double m1 = Lc2[k * DS + 64];
      GPBO_LDS_ORDER();
      if (m1 != 0.0) {
        const double li = Lc2[k * DS + i];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) a2[cc] = fma(-li, prow0[k * DS + cc], a2[cc]);
        k += 1;
        continue;
      }
      if (++spins > (1 << 22)) {
        if (i == 0) *broken = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  const double* bro = Lc + 16 * q * DS + 16 * q;   // bro[jj * DS + cc] = L00[16q + cc][16q + jj]; bro[jj * DS + 64 - 16q] = its marker
  const double* mk = Lc + 16 * q * DS + 64;         // mk[jj * DS] = 1 / L00[j][j], 0 while column j is not complete
  double* col2 = Lc2 + 16 * q * DS + i;
  double rs = mk[0], pc[16];
  GPBO_LDS_ORDER();
#pragma unroll
  for (int cc = 1; cc < 16; ++cc) pc[cc] = bro[cc];
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    int spins = 0;
    while (rs == 0.0) {     // not published when the look-ahead read it: poll, then fetch the multipliers again
      if (++spins > (1 << 22)) {
        if (i == 0) *broken = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      rs = mk[jj * DS];
      GPBO_LDS_ORDER();
#pragma unroll
      for (int cc = jj + 1; cc < 16; ++cc) pc[cc] = bro[jj * DS + cc];
    }
    double rsn = 0.0, pn[16];
    if (jj < 15) {          // look-ahead: column j + 1
      rsn = mk[(jj + 1) * DS];
      GPBO_LDS_ORDER();
#pragma unroll
      for (int cc = jj + 2; cc < 16; ++cc) pn[cc] = bro[(jj + 1) * DS + cc];
    }
    const double l2 = a2[jj] * rs;
    a2[jj] = l2;
    col2[jj * DS] = l2;
    col2[jj * DS + 64] = rs;          // marker of the L10 column (all lanes, see factor_quarter)
#pragma unroll
    for (int cc = jj + 1; cc < 16; ++cc) a2[cc] = fma(-l2, pc[cc], a2[cc]);
    rs = rsn;
#pragma unroll
    for (int cc = jj + 2; cc < 16; ++cc) pc[cc] = pn[cc];
  }
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
__device__ __forceinline__ void diag16_inverses(const double* __restrict__ Lc, double* __restrict__ Dk, const int lane) {
  const int b = lane >> 4, c = lane & 15;
  double w[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) w[r] = (r == c) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double wk = w[k] * Lc[(16 * b + k) * DS + 64];
    w[k] = wk;
#pragma unroll
    for (int r = k + 1; r < 16; ++r) w[r] = fma(-Lc[(16 * b + k) * DS + 16 * b + r], wk, w[r]);
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

__device__ __forceinline__ void inverse_wave(const double* Lc, double* Dk, double* Wout, const int q, const int lane) {
  diag16_inverses(Lc, Dk, lane);
  GPBO_LDS_ORDER();   // the wave reads back its own tiles: same-wave LDS accesses are performed in order
  switch (q) {
    case 0: inverse_colblock<0>(Lc, Dk, Wout, lane); break;
    case 1: inverse_colblock<1>(Lc, Dk, Wout, lane); break;
    case 2: inverse_colblock<2>(Lc, Dk, Wout, lane); break;
    default: inverse_colblock<3>(Lc, Dk, Wout, lane); break;
  }
}

// Diagonal block of `nblk` (1 or 2) 64-blocks starting at block kb, all earlier updates applied: factor in place, write
// inv(L_kk) (and inv(L_kk+1)) to dinv.  512 threads = 8 waves, no launch inside, five workgroup barriers:
//   waves 0-3 ("owners", thread = row i of the block being factored, wave = column quarter): potf2 of A00, later of A11
//   waves 4-7 ("followers", thread = row 64 + i): L10 riding along (follow_quarter), later inv(L00) while A11 is factored
// Quarter 3's follower sits in wave 4 — by the usual cyclic wave placement on the SIMD of owner 0, which has been idle
// longest when the chain reaches the last quarter (a wave issues one VALU instruction per 4 cycles whatever it is, so a
// follower sharing the SIMD of the owner that currently carries the chain would take its issue slots).
__device__ __forceinline__ void diag128_body(double* __restrict__ L, const int64_t ld, const int kb, const int nblk,
                                             double* __restrict__ dinv, int* __restrict__ info, double* __restrict__ smem,
                                             long long* __restrict__ stamps) {
  double* Lc0 = smem + C128_LC0;
  double* LcX = smem + C128_LCX;
  double* Wr = smem + C128_WR;
  int* flags = reinterpret_cast<int*>(smem + C128_FLAGS);
  const int tid = threadIdx.x;
  const int i = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool owner = w < 4;
  const int q = owner ? w : 7 - w;          // column quarter
  const bool two = nblk == 2;
  double* A = L + (int64_t)kb * 64 * ld + (int64_t)kb * 64;
  if (stamps && tid == 0) stamps[0] = clock64();
  if (tid < 64) {          // column markers: nothing published yet
    Lc0[tid * DS + 64] = 0.0;
    LcX[tid * DS + 64] = 0.0;
  }
  if (tid == 0) flags[0] = 0;
  double a[16], b[16];     // owners: rows of A00, A11; followers: rows of A10
#pragma unroll
  for (int h = 0; h < 16; ++h) b[h] = 0.0;
  {
    const int64_t row = owner ? i : 64 + i;
    const double2* s0 = reinterpret_cast<const double2*>(A + row * ld + 16 * q);
    if (owner || two) {
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const double2 v = s0[h];
        a[2 * h] = v.x;
        a[2 * h + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int h = 0; h < 16; ++h) a[h] = 0.0;
    }
    if (owner && two) {
      const double2* s2 = reinterpret_cast<const double2*>(A + (int64_t)(64 + i) * ld + 64 + 16 * q);
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const double2 u = s2[h];
        b[2 * h] = u.x;
        b[2 * h + 1] = u.y;
      }
    }
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[1] = clock64();
  // ---- columns 0..63: L00 by the owners, L10 = A10 L00^-T by the followers behind them
  if (owner) {
    factor_quarter<false>(a, b, Lc0, Lc0, &flags[0], i, q, stamps ? stamps + 7 : nullptr);
    // the wave's 16 columns are final: rows straight from registers (128 contiguous bytes per thread, zeros above the diagonal)
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) a[cc] = (i >= 16 * q + cc) ? a[cc] : 0.0;
    double2* d0 = reinterpret_cast<double2*>(A + (int64_t)i * ld + 16 * q);
#pragma unroll
    for (int h = 0; h < 8; ++h) d0[h] = make_double2(a[2 * h], a[2 * h + 1]);
  } else if (two) {
    follow_quarter(a, Lc0, LcX, &flags[0], i, q);
    double2* d1 = reinterpret_cast<double2*>(A + (int64_t)(64 + i) * ld + 16 * q);
#pragma unroll
    for (int h = 0; h < 8; ++h) d1[h] = make_double2(a[2 * h], a[2 * h + 1]);
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[2] = clock64();
  const int lane = i, lr = lane & 15, lk = lane >> 4;
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
    if (owner) {
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) b[cc] -= Wr[i * 81 + 16 * q + cc];
    }
    if (tid < 64) LcX[tid * DS + 64] = 0.0;      // the L10 image makes room for L11: its markers start over
    __syncthreads();   // the exchange area is re-used for the waves' diagonal inverses
    // ---- columns 64..127: L11 by the owners; inv(L00) by the followers meanwhile
    if (owner) {
      factor_quarter<false>(b, a, LcX, LcX, &flags[0], i, q);
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) b[cc] = (i >= 16 * q + cc) ? b[cc] : 0.0;
      double2* d2 = reinterpret_cast<double2*>(A + (int64_t)(64 + i) * ld + 64 + 16 * q);
#pragma unroll
      for (int h = 0; h < 8; ++h) d2[h] = make_double2(b[2 * h], b[2 * h + 1]);
      if (stamps && tid == 0) stamps[4] = clock64();
    } else {
      inverse_wave(Lc0, Wr + (w - 4) * 4 * 272, dinv + (int64_t)kb * 4096, w - 4, lane);
    }
    __syncthreads();
    if (stamps && tid == 0) stamps[5] = clock64();
    if (owner) inverse_wave(LcX, Wr + w * 4 * 272, dinv + (int64_t)(kb + 1) * 4096, w, lane);
  } else {
    if (owner) inverse_wave(Lc0, Wr + w * 4 * 272, dinv + (int64_t)kb * 4096, w, lane);
  }
  if (w == 0) {
    // LAPACK potrf: order of the first non-positive leading minor (the images are complete: the second inverse / the only one
    // started behind a barrier)
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

// One launch = diagonal block(s) kb (workgroup 0) || the 64x64 tiles of the previous column block's in-panel update, two per
// 512-thread workgroup (every tile but the leading skip00 x skip00 ones, which chol128_diag_update_kernel has already
// brought up to date).
__global__ __launch_bounds__(512) void chol128_step_kernel(double* L, int64_t ld, int kb, int nblk, double* dinv, int* info,
                                                            GemmArgs g, int tiles_n, int tiles, int64_t lane_stride, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) double c128_smem[];
  const int zl = (int)blockIdx.y;
  if (blockIdx.x == 0) {
    diag128_body(L + (int64_t)zl * lane_stride, ld, kb, nblk, dinv + (int64_t)zl * lane_stride, info + (int64_t)zl * lane_stride * 2,
                 c128_smem, stamps);
    return;
  }
  const int half = (int)(threadIdx.x >> 8);
  const int t = 2 * ((int)blockIdx.x - 1) + half;
  int bm = t / tiles_n, bn = t - bm * tiles_n;
  const bool mine = t < tiles && !(g.lower_only && bn > bm) && !(bn < g.skip00 && bm < g.skip00);
  if (!mine) { bm = g.m / 64 - 1; bn = 0; }   // a tile that exists (the last row tile): loads only, same barrier count
  GemmArgs h = g;
  h.lower_only = 0; h.skip00 = 0;            // decided above
  gemm_tile_body<true, false>(h, bm, bn, zl, 0, c128_smem + half * GT_LDS_DOUBLES, (int)(threadIdx.x & 255), mine);
}

// Panel solve below a 128-column diagonal block, in place: X = A inv(L_blk)^T with L_blk = [[L00, 0], [L10, L11]], i.e.
//   X0 = A0 W00^T,   X1 = (A1 - X0 L10^T) W11^T        (W00, W11 = the 64x64 inverses from the diagonal workgroup).
// One workgroup = 16 panel rows, worked on TRANSPOSED (X^T = W A^T): the accumulator of each product is the B fragment
// of the next, so the three products chain through 8 KiB of LDS exchange only; wave w owns column tile w (tile 3 - w
// in the last product, which balances the triangular k-ranges).  All operand fragments come straight from L2 (the
// 96 KiB of W00 / L10 / W11 are shared by every workgroup of the launch) and are requested up front.
__global__ __launch_bounds__(256) void chol128_panel_kernel(double* L, int64_t ld, int kb, const double* __restrict__ dinv,
                                                             int64_t lane_stride) {
  __shared__ __attribute__((aligned(16))) double Xs[4][4][64];   // [k tile][k group][lane]: B fragments of X0^T
  __shared__ __attribute__((aligned(16))) double Ts[4][4][64];   // ... of T^T
  L += (int64_t)blockIdx.y * lane_stride;
  dinv += (int64_t)blockIdx.y * lane_stride;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t c0 = (int64_t)kb * 64;
  double* Arow = L + (c0 + 128 + (int64_t)blockIdx.x * 16) * ld + c0;       // A[n][k] = Arow[n * ld + k]
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
// C -= X X^T with X = the solved panel rows of that block (k = 128).  One 16x16 tile per workgroup, k split over the
// four waves, partial sums combined through LDS in a fixed order — 36 small workgroups instead of a 128^3 product on
// the diagonal workgroup's single CU (8.5 us there).
__global__ __launch_bounds__(256) void chol128_diag_update_kernel(double* L, int64_t ld, int kb, int64_t lane_stride) {
  __shared__ __attribute__((aligned(16))) double Ps[4][4][64];
  L += (int64_t)blockIdx.y * lane_stride;
  const int t = (int)blockIdx.x;
  int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  while (ti * (ti + 1) / 2 > t) --ti;
  const int tj = t - ti * (ti + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
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

// ---- launchers -------------------------------------------------------------------------------------------------------
static int launch_step(gpbo_ctx* ctx, Model& m, int kb, int nblk, const GemmArgs* upd, long long* stamps) {
  if (!(ctx->func_attrs & ATTR_CHOL128)) {
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(chol128_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)C128_LDS_BYTES));
    ctx->func_attrs |= ATTR_CHOL128;
  }
  GemmArgs g{};
  int tiles = 0, tiles_n = 1;
  if (upd) {
    g = *upd;
    g.lanes = ctx->lanes; g.lane_stride = ctx->lane_stride; g.batch = 1;
    tiles_n = g.n / 64;
    tiles = (g.m / 64) * tiles_n;
    if (g.m / 64 <= g.skip00 && tiles_n <= g.skip00) tiles = 0;   // nothing but the block the diagonal workgroup owns
  }
  chol128_step_kernel<<<dim3((unsigned)(1 + (tiles + 1) / 2), (unsigned)ctx->lanes), dim3(512), C128_LDS_BYTES, ctx->stream>>>(
      m.L, m.NP, kb, nblk, m.dinv, ctx->info_dev, g, tiles_n, tiles, ctx->lane_stride, stamps);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// ---- look-ahead over the outer panels ----------------------------------------------------------------------------
// The rank-`outer` trailing update of panel p only has to reach the NEXT panel's columns before that panel's chain can
// start; the rest of it (T_b) is independent of the chain and runs on a second ("bulk") stream meanwhile:
//     main:  chain(p) | [wait T_b(p-1)] T_a(p) | chain(p+1) | ...
//     bulk:            [wait chain(p)]  T_b(p)
// T_a(p) waits for T_b(p-1) because both accumulate into the next panel's columns and the order of the two rank-512
// contributions is part of the result (bitwise the one-stream factor: same launches, same kernels, same order per tile).
// Round 2 measured this with a plain second stream and dropped it: the bulk GEMM's workgroups hold every CU, so the
// chain's small kernels queue behind them.  Here the bulk stream may be confined to a subset of the CUs
// (hipExtStreamCreateWithCUMask; GPBO_CHOL_LA_CUS = how many of the device's CUs it may use, 0 = no mask), which keeps
// the rest free for the chain at any moment.  GPBO_CHOL_LA=0 turns the look-ahead off (A/B runs).
static int lookahead_min_np() {
  static const int v = [] {
    const char* e = getenv("GPBO_CHOL_LA");
    if (e && e[0] == '0') return 1 << 30;
    const char* f = getenv("GPBO_CHOL_LA_MIN_NP");
    return f ? atoi(f) : 2048;
  }();
  return v;
}

static LookAhead* lookahead_for(gpbo_ctx* ctx, int n_events) {
  LookAhead* la = nullptr;
  for (auto& l : ctx->lookahead)
    if (l.main == ctx->stream) la = &l;
  if (!la) {
    LookAhead l;
    l.main = ctx->stream;
    static const int cus = getenv("GPBO_CHOL_LA_CUS") ? atoi(getenv("GPBO_CHOL_LA_CUS")) : 0;
    hipError_t e = hipErrorUnknown;
    if (cus > 0) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && cus < prop.multiProcessorCount) {
        std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, 0u);
        for (int b = 0; b < cus; ++b) mask[(size_t)b / 32] |= 1u << (b % 32);
        e = hipExtStreamCreateWithCUMask(&l.bulk, (uint32_t)mask.size(), mask.data());
      }
    }
    if (e != hipSuccess) e = hipStreamCreateWithFlags(&l.bulk, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    ctx->lookahead.push_back(l);
    la = &ctx->lookahead.back();
  }
  while ((int)la->ev.size() < n_events) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    la->ev.push_back(ev);
  }
  return la;
}

// Blocked Cholesky of m.L (lower, in place), inverted 64x64 diagonal blocks to m.dinv: 128-column steps inside
// `outer`-column panels (outer a multiple of 128), one rank-`outer` trailing update per panel.
int launch_cholesky128(gpbo_ctx* ctx, Model& m, int outer, long long* stamps) {
  const int nblk = (int)(m.NP / NB);
  const int per_outer = outer / NB;
  const unsigned lanes = (unsigned)ctx->lanes;
  int rc;
  const int n_panels = (nblk + per_outer - 1) / per_outer;
  LookAhead* la = (ctx->lanes == 1 && m.NP >= lookahead_min_np() && n_panels >= 3) ? lookahead_for(ctx, 2 * n_panels) : nullptr;
  bool la_joined = true;
  int pidx = 0;     // outer panel index
  for (int ob = 0; ob < nblk; ob += per_outer) {
    const int oe = (ob + per_outer < nblk) ? ob + per_outer : nblk;
    if ((rc = launch_step(ctx, m, ob, (oe - ob >= 2) ? 2 : 1, nullptr, stamps))) return rc;   // everything before the panel is applied
    for (int kb = ob; kb < oe; kb += 2) {
      const int wb = (oe - kb >= 2) ? 2 : 1;
      const int rem = (int)(m.NP - (int64_t)(kb + wb) * NB);      // rows below the diagonal block
      if (rem == 0) break;
      if (wb != 2) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "cholesky128: a single 64-block can only end the matrix");
      chol128_panel_kernel<<<dim3((unsigned)(rem / 16), lanes), dim3(256), 0, ctx->stream>>>(m.L, m.NP, kb, m.dinv, ctx->lane_stride);
      GPBO_HIP(ctx, hipGetLastError());
      const int next = kb + 2;
      if (next < oe) {
        const int nb2 = (oe - next >= 2) ? 2 : 1;
        const int nt = 4 * nb2;
        chol128_diag_update_kernel<<<dim3((unsigned)(nt * (nt + 1) / 2), lanes), dim3(256), 0, ctx->stream>>>(m.L, m.NP, kb,
                                                                                                            ctx->lane_stride);
        GPBO_HIP(ctx, hipGetLastError());
        double* panel = m.L + (int64_t)next * NB * m.NP + (int64_t)kb * NB;
        GemmArgs s{};    // rank-128 update of the panel's remaining columns; the next diagonal block is already done
        s.m = rem; s.n = (oe - next) * NB; s.k = 2 * NB; s.alpha = -1.0; s.beta = 1.0;
        s.A = panel; s.lda = m.NP; s.B = panel; s.ldb = m.NP; s.b_trans = 1;
        s.C = m.L + (int64_t)next * NB * m.NP + (int64_t)next * NB; s.ldc = m.NP;
        s.lower_only = 1; s.skip00 = nb2;
        if ((rc = launch_step(ctx, m, next, nb2, &s, nullptr))) return rc;
      }
    }
    const int rem2 = (int)(m.NP - (int64_t)oe * NB);
    if (rem2 > 0) {
      const double* P = m.L + (int64_t)oe * NB * m.NP + (int64_t)ob * NB;
      GemmArgs t{};
      t.k = (oe - ob) * NB; t.alpha = -1.0; t.beta = 1.0;
      t.lda = m.NP; t.ldb = m.NP; t.b_trans = 1; t.ldc = m.NP;
      t.batch = 1; t.lower_only = 1;
      const int nw = rem2 < outer ? rem2 : outer;       // the next panel's columns
      if (!la || rem2 <= nw) {
        if (la && pidx > 0) GPBO_HIP(ctx, hipStreamWaitEvent(ctx->stream, la->ev[2 * (pidx - 1) + 1], 0));
        t.m = rem2; t.n = rem2; t.A = P; t.B = P;
        t.C = m.L + (int64_t)oe * NB * m.NP + (int64_t)oe * NB;
        if ((rc = launch_gemm(ctx, t))) return rc;
        la_joined = true;
      } else {
        // chain(p) is complete on main: the bulk stream may read the panel
        GPBO_HIP(ctx, hipEventRecord(la->ev[2 * pidx], ctx->stream));
        if (pidx > 0) GPBO_HIP(ctx, hipStreamWaitEvent(ctx->stream, la->ev[2 * (pidx - 1) + 1], 0));
        t.m = rem2; t.n = nw; t.A = P; t.B = P;
        t.C = m.L + (int64_t)oe * NB * m.NP + (int64_t)oe * NB;
        if ((rc = launch_gemm(ctx, t))) return rc;        // T_a on main
        hipStream_t main_stream = ctx->stream;
        ctx->stream = la->bulk;
        hipError_t e = hipStreamWaitEvent(la->bulk, la->ev[2 * pidx], 0);
        if (e == hipSuccess) {
          t.m = rem2 - nw; t.n = rem2 - nw; t.A = P + (int64_t)nw * m.NP; t.B = t.A;
          t.C = m.L + ((int64_t)oe * NB + nw) * m.NP + ((int64_t)oe * NB + nw);
          rc = launch_gemm(ctx, t);                          // T_b on bulk
          if (rc == GPBO_OK) e = hipEventRecord(la->ev[2 * pidx + 1], la->bulk);
        }
        ctx->stream = main_stream;
        if (rc) return rc;
        GPBO_HIP(ctx, e);
        la_joined = false;
      }
    }
    ++pidx;
  }
  // nothing of the factorisation may still be in flight on the bulk stream when main goes on (it always joined above:
  // the last update has no far part)
  if (la && !la_joined && pidx > 0) GPBO_HIP(ctx, hipStreamWaitEvent(ctx->stream, la->ev[2 * (pidx - 1) + 1], 0));
  return GPBO_OK;
}

}  // namespace gpbo
