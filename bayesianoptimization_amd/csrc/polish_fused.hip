// gpbo_polish_seeds as ONE launch: one workgroup per local search, the evaluations AND the optimiser inside it.
//
// What it replaces: the local-search stage of AcquisitionFunction._smart_minimize (bayes_opt/acquisition.py:322-420) for the
// sizes where it is launch latency and nothing else.  polish.hip advances all runs in lockstep, one batched evaluation per round:
// six dependent launches and a stream synchronisation (41-75 us) for ~2 N^2 flops per run.  Here a run is a workgroup that owns
// its search from the seed to the stopping rule, THREAD = TRAINING POINT (polish_rows_kernel, round 6):
//   * thread i keeps k*_i, v_i = (W k*)_i and u_i = (W^T v)_i of its own point: it walks row i of W = L^-1 for v and column i for u;
//   * NP <= 128: W sits in LDS as a padded square ([NP][NP + 1]: both walks conflict-free).  128 < NP <= 512: W stays in memory and
//     both walks are coalesced — the row walk over a transposed copy made once per fit, the column walk over W itself;
//   * the sums over the points (mu, |v|^2, the 2 d gradient sums) are taken by lane groups of one dimension each and combined in a
//     fixed order; the optimiser (polish_opt.h's steps) runs on wave 0 with a lane per variable, its sums over the variables as
//     DPP row reductions, its two-loop recursion as ONE rolled loop;
//   * four barriers per evaluation.  3.4-3.9 us per evaluation at N <= 64, 5.3-5.7 at 128, 16 / 21 / 31 / 41 at N = 143 / 256 / 384 / 512
//     (one CU streams W at ~30 B per clock), against 14 / 19 / 24-46 for round 5's eight-wave kernel and 41-57 us per lockstep
//     round (profiles/r06_polish_fused_ab.json).
// Deterministic, and NOT the bits of the lockstep path (other summation orders): the evaluation agrees with gpbo_predict_grad to
// ~3e-12 of the values' scale, a whole search ends at the same or a better value ("parity is statistical
// (same or better acquisition value), not bit-wise"); the lockstep path is the checker (tests/test_gpu_polish_fused.py).  Round 5's
// eight-wave kernel — the six kernels' arithmetic phase by phase, bitwise the lockstep path, nine barriers and two passes over W
// at 25 GB/s per evaluation — is in the git history; every size it served is served faster here.
// One model (no constraint slots).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "gpbo_internal.h"
#include "polish_opt.h"

namespace gpbo {

namespace {

struct PolishFusedArgs {
  const double *W, *Wt, *Xs, *alpha, *ls;      // Wt: W transposed, row-major (polish_rows_kernel with W in memory), else null
  int NP, N, d, DP;
  double y_mean, y_std;
  int acq;
  double acq_param, y_max;
  int max_iter, eval_only;             // eval_only = R > 0: R evaluations at the seed, no search (debug entry)
  const double *seeds, *lo, *hi;      // (n_seeds, d), (d,), (d,): device-visible pinned host memory
  double *x_out, *f_out;              // (n_seeds, d), (n_seeds,): pinned
  int *status_out, *iter_out, *eval_out;
  double* dbg;                        // eval_only: per seed [f, mu, sd, 0 | g (d) | dmu (d) | dsd (d)]
  int* negvar;
};

struct DevCdf {
  __device__ __forceinline__ double operator()(double z) const { return 0.5 * erfc(-z * 0.70710678118654752440); }
};
struct DevPdf {
  __device__ __forceinline__ double operator()(double z) const { return exp(-0.5 * z * z) * 0.39894228040143267794; }
};

constexpr int PF_LDS_CAP = 160 * 128 - 8;      // doubles in 160 KiB, less the flag words

__device__ __forceinline__ double pf_lane(double v, int i) {      // v of lane i (i uniform), in every lane
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), i);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), i);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

// ---- the optimiser: polish_opt.h's steps over wave 0, lane i = variable i (d <= 64; lanes >= d carry zeros) ---------------------
// Against the host's arithmetic: (a) a sum over the variables is a DPP butterfly inside the rows of 16
// lanes + the rows' totals (the host's left-to-right chain would be 3 d instructions of v_readlane + add); (b) s.y and y.y of a correction pair are kept from the step that
// stored it — the recursion recomputes them only while some variable sits on a bound (the sums then run over the free variables);
// (c) the CODE is kept short: the two loops of the recursion are ONE rolled loop with one reduction in its body, the new direction
// is formed at one place.  (c) is what the time hangs on: with every loop unrolled and the routines inlined at each call site the
// kernel was 65 KB of instructions, more than the instruction cache two CUs share — in-kernel clocks: 8-9 000 cycles per step AND
// the evaluation beside it 14 000 instead of the 8 200 it takes alone (profiles/r06_polish_fused_ab.json, notes).
template <int CTRL>
__device__ __forceinline__ double pr_dpp(double v) {      // v of the lane the DPP pattern CTRL pairs this one with (inside a row of 16)
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: after the four steps every lane holds its row's total
__device__ __forceinline__ double pr_sum(double v, int d) {      // lanes >= d carry zeros
#pragma clang fp contract(off)
  v += pr_dpp<0xB1>(v);
  v += pr_dpp<0x4E>(v);
  v += pr_dpp<0x141>(v);
  v += pr_dpp<0x140>(v);
  double acc = pf_lane(v, 0);
  if (d > 16) acc += pf_lane(v, 16);
  if (d > 32) acc = (acc + pf_lane(v, 32)) + pf_lane(v, 48);
  return acc;
}
__device__ __forceinline__ double pr_max_abs(double v, int d) {  // max |v_i|, a NaN never wins (polish_opt.h)
  double m = __builtin_fabs(v);
  if (m != m) m = 0.0;
  m = polish_max(m, pr_dpp<0xB1>(m));
  m = polish_max(m, pr_dpp<0x4E>(m));
  m = polish_max(m, pr_dpp<0x141>(m));
  m = polish_max(m, pr_dpp<0x140>(m));
  double acc = pf_lane(m, 0);
  if (d > 16) acc = polish_max(acc, pf_lane(m, 16));
  if (d > 32) acc = polish_max(polish_max(acc, pf_lane(m, 32)), pf_lane(m, 48));
  return acc;
}

struct RowsRun {
  double x, g, xt, dir, lo, hi;        // this lane's variable
  bool freev;
  double f, alpha;
  int hist, head, iter, evals, ls, phase, status;
};
// LDS of the optimiser (doubles): S, Y [LBFGS_M][d] (lane i reads and writes column i) | s.y, y.y over ALL variables, rho, a
// [LBFGS_M each] (written and read by every lane alike: program order within the one wave)
__host__ __device__ inline int pr_opt_doubles(int d) { return 2 * LBFGS_M * d + 4 * LBFGS_M; }

// one answer (ft, this lane's gradient component gt — non-finite components already 0) of the objective: polish_advance
__device__ __forceinline__ void pr_advance(RowsRun& r, double ft, double gt, int d, int lane, double* opt, int max_iter) {
#pragma clang fp contract(off)
  double* S = opt;
  double* Y = S + LBFGS_M * d;
  double* SY = Y + LBFGS_M * d;
  double* YY = SY + LBFGS_M;
  double* RHO = YY + LBFGS_M;
  double* AV = RHO + LBFGS_M;
  const bool mine = lane < d;
  ++r.evals;
  bool fresh = false;      // a new direction is due
  if (r.phase == 0) {
    r.x = r.xt; r.g = gt; r.f = ft;
    if (!__builtin_isfinite(ft)) { r.phase = 2; r.status = 2; return; }
    r.phase = 1;
    fresh = true;
  } else {
    const double sd = r.xt - r.x;
    const double gs = pr_sum(r.g * sd, d), moved = pr_max_abs(sd, d);
    const bool ok = __builtin_isfinite(ft) && ft <= r.f + 1e-4 * gs;
    if (!ok) {
      const bool flat = __builtin_isfinite(ft) && r.ls >= 2 &&
                        __builtin_fabs(ft - r.f) <= POLISH_FTOL * polish_max(polish_max(__builtin_fabs(ft), __builtin_fabs(r.f)), 1.0);
      if (moved == 0.0 || flat) { r.phase = 2; r.status = 1; return; }
      if (++r.ls >= POLISH_MAXLS) { r.phase = 2; r.status = 3; return; }
      double shrink = 0.1;
      if (__builtin_isfinite(ft)) {
        const double curv = ft - r.f - gs;
        shrink = curv > 0.0 ? polish_min(polish_max(-gs / (2.0 * curv), 0.1), 0.5) : 0.5;
      }
      r.alpha *= shrink;
    } else {
      // accepted: the pair (s, y) joins the ring when it is usable
      const double s = sd, y = gt - r.g;
      const double sy = pr_sum(s * y, d), yy = pr_sum(y * y, d);
      if (sy > 2.2e-16 * yy && yy > 0.0) {
        if (mine) {
          S[r.head * d + lane] = s;
          Y[r.head * d + lane] = y;
        }
        SY[r.head] = sy;
        YY[r.head] = yy;
        r.head = (r.head + 1) % LBFGS_M;
        r.hist = (r.hist + 1 < LBFGS_M) ? r.hist + 1 : LBFGS_M;
      }
      const double f_old = r.f;
      r.x = r.xt; r.g = gt; r.f = ft;
      ++r.iter;
      if ((f_old - ft) <= POLISH_FTOL * polish_max(polish_max(__builtin_fabs(f_old), __builtin_fabs(ft)), 1.0)) {
        // (the projected-gradient test comes first in polish_advance: status 0 wins where both hold)
        r.phase = 2;
        r.status = (pr_max_abs(polish_min(polish_max(r.x - r.g, r.lo), r.hi) - r.x, d) <= POLISH_PGTOL) ? 0 : 1;
        return;
      }
      fresh = true;
    }
  }
  if (fresh) {
    if (pr_max_abs(polish_min(polish_max(r.x - r.g, r.lo), r.hi) - r.x, d) <= POLISH_PGTOL) { r.phase = 2; r.status = 0; return; }
    if (r.iter >= max_iter && r.iter > 0) { r.phase = 2; r.status = 2; return; }
    // ---- the new direction: two-loop recursion over the free variables (polish_new_direction)
    r.freev = mine && !((r.x <= r.lo && r.g > 0.0) || (r.x >= r.hi && r.g < 0.0));
    const bool allfree = __ballot(r.freev) == __ballot(mine);
    int usable = 0, used = 0;          // bit t: the t-th newest pair takes part
    double gamma = 1.0;
    for (int t = 0; t < r.hist; ++t) {
      const int k = (r.head - 1 - t + 2 * LBFGS_M) % LBFGS_M;
      double sy = SY[k], yy = YY[k];
      if (!allfree) {
        const double s = mine ? S[k * d + lane] : 0.0, y = mine ? Y[k * d + lane] : 0.0;
        sy = pr_sum(r.freev ? s * y : 0.0, d);
        yy = pr_sum(r.freev ? y * y : 0.0, d);
      }
      if ((sy > 2.2e-16 * yy) && (yy > 0.0)) {
        usable |= 1 << t;
        RHO[k] = 1.0 / sy;
        if (used == 0) gamma = sy / yy;
        ++used;
      }
    }
    double q = r.freev ? r.g : 0.0;
    // j < hist: the t = j-th newest pair, q -= a_t y_t; then q *= gamma; j >= hist: back from the oldest, q += (a_t - b_t) s_t
    for (int j = 0; j < 2 * r.hist; ++j) {
      const bool first = j < r.hist;
      const int t = first ? j : 2 * r.hist - 1 - j;
      if (j == r.hist) q *= gamma;
      if (!((usable >> t) & 1)) continue;
      const int k = (r.head - 1 - t + 2 * LBFGS_M) % LBFGS_M;
      const double s = mine ? S[k * d + lane] : 0.0, y = mine ? Y[k * d + lane] : 0.0;
      const double dot = pr_sum(r.freev ? (first ? s : y) * q : 0.0, d);
      const double c = RHO[k] * dot;
      if (first) {
        AV[k] = c;
        if (r.freev) q -= c * y;
      } else {
        if (r.freev) q += (AV[k] - c) * s;
      }
    }
    if (r.hist == 0) q *= gamma;
    r.dir = r.freev ? -q : 0.0;
    const double gd = pr_sum(r.dir * r.g, d), gn = pr_sum(r.freev ? r.g * r.g : 0.0, d);
    if (!(gd < 0.0) || !__builtin_isfinite(gd)) {     // not a descent direction: steepest descent over the free variables, history dropped
      r.hist = 0;
      used = 0;
      r.dir = r.freev ? -r.g : 0.0;
    }
    r.alpha = (used == 0) ? polish_min(1.0, 1.0 / __builtin_sqrt(polish_max(gn, 1e-300))) : 1.0;
    r.ls = 0;
  }
  r.xt = polish_min(polish_max(r.x + r.alpha * r.dir, r.lo), r.hi);      // the trial point (polish_trial_point)
}

// Wt = W^T (both row-major NP x NP, NP a multiple of 64): one 64x64 tile per 256-thread workgroup through a padded LDS image
__global__ __launch_bounds__(256) void transpose_w_kernel(const double* __restrict__ W, double* __restrict__ Wt, int NP) {
  __shared__ double tile[64][65];
  const int bi = (int)blockIdx.y, bj = (int)blockIdx.x, c = (int)threadIdx.x & 63, r0 = (int)threadIdx.x >> 6;
  for (int r = r0; r < 64; r += 4) tile[r][c] = W[(int64_t)(64 * bi + r) * NP + 64 * bj + c];
  __syncthreads();
  for (int r = r0; r < 64; r += 4) Wt[(int64_t)(64 * bj + r) * NP + 64 * bi + c] = tile[c][r];
}

// ---- thread = training point (NP <= 512) ---------------------------------------------------------------------------------
#ifndef GPBO_PR_INFLIGHT
#define GPBO_PR_INFLIGHT 16
#endif
constexpr int PR_INFLIGHT = GPBO_PR_INFLIGHT;      // 16-byte loads in flight per lane in the walks over W in memory (128 is a multiple)
constexpr int PR_LDS_NP = 128;       // W fits the LDS up to here (a padded square: 132 KB at 128)
constexpr int PR_MAX_NP = 512;       // ... and is streamed from memory above (W for the column walk, its transpose for the row walk): 8 waves,
                                     // 256 VGPRs each (12 waves for NP = 768 would spill the optimiser's registers to scratch)
// LDS (doubles): [W [NP][NP + 1]] | xs [64] | ls [64] | alpha, k*, v [NP each] | (c1, c2) [NP][2] | (v^2, k* alpha) [NP][2] | u [NP] |
// group partials [groups][2 DP + 2] | the optimiser's block (pr_opt_doubles) | X [NP][DP + 1] (when it fits) ; then the flag word
__host__ __device__ inline int pr_groups(int NP, int DP) { return (64 / DP) * (NP >> 6); }
__host__ __device__ inline int pr_lds_base(int NP, int d, int DP, bool wlds) {
  return (wlds ? NP * (NP + 1) : 0) + 128 + 8 * NP + pr_groups(NP, DP) * (2 * DP + 2) + pr_opt_doubles(d);
}
__host__ __device__ inline int pr_xs_stage(int NP, int d, int DP, bool wlds) {
  const int want = NP * (DP + 1);
  return (pr_lds_base(NP, d, DP, wlds) + want <= PF_LDS_CAP) ? want : 0;
}
__host__ __device__ inline int pr_lds_doubles(int NP, int d, int DP, bool wlds) {
  return pr_lds_base(NP, d, DP, wlds) + pr_xs_stage(NP, d, DP, wlds);
}

// WLDS = false (round 6, 128 < NP <= 512): W stays in memory and both walks are COALESCED: the row walk reads the transposed copy
// (Wt[k][i], lanes = consecutive i), the column walk reads W itself (W[i'][k], lanes = consecutive k), two rows (columns) per thread
// with 16-byte loads, PR_INFLIGHT of them in flight per lane.  A CU pulls NP^2 / 2 * 8 B per walk at ~64 B per clock: 0.4 us at NP = 256, 1.7 us
// at 512 — against two passes of the eight-wave kernel at 25 GB/s per CU and against the 41-75 us of a six-launch lockstep round.
template <int KERNEL, bool WLDS>
__global__ __launch_bounds__(WLDS ? PR_LDS_NP : PR_MAX_NP) void polish_rows_kernel(const PolishFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) double pr_smem[];
  const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int sidx = (int)blockIdx.x;
  const int NP = a.NP, N = a.N, d = a.d, DP = a.DP;
  const int WLD = NP + 1;
  double* Wl = pr_smem;                     // [NP][NP + 1]: row walks and column walks both hit 64 different banks
  double* xs = Wl + (WLDS ? NP * WLD : 0);  // [64] the trial point over the length scales, zero padded
  double* ls_s = xs + 64;                   // [64]
  double* al_s = ls_s + 64;                 // [NP] alpha
  double* ks = al_s + NP;                   // [NP] k*
  double* vs = ks + NP;                     // [NP] v = W k*
  double* cc = vs + NP;                     // [NP][2] alpha_k f_k, u_k f_k
  double* pp = cc + 2 * NP;                 // [NP][2] v_k^2, k*_k alpha_k
  double* us = pp + 2 * NP;                 // [NP] u = W^T v (W in memory: the pair threads hand their columns' sums over)
  double* red = us + NP;                    // [groups][2 DP + 2]
  double* opt = red + pr_groups(NP, DP) * (2 * DP + 2);
  double* Xl = opt + pr_opt_doubles(d);
  const int xs_staged = pr_xs_stage(NP, d, DP, WLDS);
  int* flag = (int*)(Xl + xs_staged);
  const double* __restrict__ Xs = xs_staged ? Xl : a.Xs;
  const int xld = xs_staged ? DP + 1 : DP;

  if (WLDS)
    for (int e = tid; e < NP * NP; e += NP) {         // (blockDim.x == NP) coalesced rows of the matrix in memory, zeros above the diagonal included
      const int i = e / NP, k = e - i * NP;
      Wl[i * WLD + k] = a.W[e];
    }
  al_s[tid] = a.alpha[tid];
  if (tid < 64) ls_s[tid] = (tid < d) ? a.ls[tid] : 1.0;
  if (xs_staged)
    for (int e = tid; e < NP * DP; e += NP) {
      const int k = e / DP, t = e - k * DP;
      Xl[k * (DP + 1) + t] = a.Xs[e];
    }
  RowsRun run{};
  if (wave == 0) {
    const bool mine = lane < d;
    run.lo = mine ? a.lo[lane] : 0.0;
    run.hi = mine ? a.hi[lane] : 0.0;
    run.xt = mine ? polish_min(polish_max(a.seeds[(size_t)sidx * d + lane], run.lo), run.hi) : 0.0;      // polish_start
    run.alpha = 1.0;
    run.status = 2;
    if (lane < DP) xs[lane] = mine ? run.xt / a.ls[lane] : 0.0;
    if (lane == 0) flag[0] = 0;
  }
  __syncthreads();

  // the sums over the training points: lane groups of DP lanes (one dimension each), group j takes the points k = j (mod groups)
  const int gpw = 64 / DP, groups = gpw * (NP >> 6);
  const int grp = wave * gpw + lane / DP, gt = lane % DP;
  const int round_cap = 4 * a.max_iter + 64;
  const double al_i = al_s[tid];
  const double* wrow = Wl + tid * WLD;

  for (int round = 0;; ++round) {
    // ---- k*_i and the gradient factor f_i of this thread's point
    double fi;
    {
      const double* xr = Xs + (int64_t)tid * xld;
      double d2 = 0.0;
      for (int t = 0; t < DP; ++t) {
        const double df = xs[t] - xr[t];
        d2 = fma(df, df, d2);
      }
      const double kv = gpbo_kernel_value<KERNEL>(d2);
      if (KERNEL == GPBO_KERNEL_MATERN25) {
        const double sq = gpbo_sqrt_pos(d2) * 2.23606797749978969641;      // sqrt(5) r
        fi = -1.66666666666666666667 * (1.0 + sq) * gpbo_exp_nonpos(-sq);
      } else {
        fi = -kv;
      }
      if (tid >= N) fi = 0.0;          // padding rows carry no gradient
      ks[tid] = kv;
    }
    __syncthreads();
    // ---- v = W k* and u = W^T v, then every point's two gradient weights and its terms of |v|^2 and k* . alpha
    if (WLDS) {
      // thread i: row i of W for v_i (zeros above the diagonal: the whole row), column i for u_i
      double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
      for (int k = 0; k < NP; k += 4) {
        v0 = fma(wrow[k], ks[k], v0);
        v1 = fma(wrow[k + 1], ks[k + 1], v1);
        v2 = fma(wrow[k + 2], ks[k + 2], v2);
        v3 = fma(wrow[k + 3], ks[k + 3], v3);
      }
      const double v = (tid < N) ? (v0 + v1) + (v2 + v3) : 0.0;
      vs[tid] = v;
      pp[2 * tid] = v * v;
      pp[2 * tid + 1] = ks[tid] * al_i;
      __syncthreads();
      double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
      const double* wcol = Wl + tid;
      for (int i = 0; i < NP; i += 4) {
        u0 = fma(wcol[i * WLD], vs[i], u0);
        u1 = fma(wcol[(i + 1) * WLD], vs[i + 1], u1);
        u2 = fma(wcol[(i + 2) * WLD], vs[i + 2], u2);
        u3 = fma(wcol[(i + 3) * WLD], vs[i + 3], u3);
      }
      cc[2 * tid] = al_i * fi;
      cc[2 * tid + 1] = ((u0 + u1) + (u2 + u3)) * fi;
    } else {
      // W in memory: the first NP / 2 threads walk TWO rows (columns) each with 16-byte loads — a wave instruction is 1 KB of one row
      // of Wt (of W) — wave w the rows 128 w .. 128 w + 127, whose columns end at 128 (w + 1) (the columns 128 w .., whose rows
      // start at 128 w): the zeros of the triangle are never loaded.  The sums come back through LDS to the points' own threads.
      typedef double d2 __attribute__((ext_vector_type(2)));
      const int ld2 = NP / 2;
      if (tid < ld2) {
        const d2* __restrict__ wt = reinterpret_cast<const d2*>(a.Wt) + tid;      // Wt[k][2 tid .. 2 tid + 1] = W[2 tid ..][k]
        const int kend = min(NP, 128 * (wave + 1));
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        for (int k = 0; k < kend; k += PR_INFLIGHT) {
          d2 w[PR_INFLIGHT];
#pragma unroll
          for (int e = 0; e < PR_INFLIGHT; ++e) w[e] = wt[(int64_t)(k + e) * ld2];
#pragma unroll
          for (int e = 0; e < PR_INFLIGHT; e += 2) {
            a0 = fma(w[e].x, ks[k + e], a0);
            b0 = fma(w[e].y, ks[k + e], b0);
            a1 = fma(w[e + 1].x, ks[k + e + 1], a1);
            b1 = fma(w[e + 1].y, ks[k + e + 1], b1);
          }
        }
        vs[2 * tid] = (2 * tid < N) ? a0 + a1 : 0.0;
        vs[2 * tid + 1] = (2 * tid + 1 < N) ? b0 + b1 : 0.0;
      }
      __syncthreads();
      pp[2 * tid] = vs[tid] * vs[tid];
      pp[2 * tid + 1] = ks[tid] * al_i;
      if (tid < ld2) {
        const d2* __restrict__ wc = reinterpret_cast<const d2*>(a.W) + tid;       // W[i][2 tid .. 2 tid + 1]
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        for (int i = 128 * wave; i < NP; i += PR_INFLIGHT) {
          d2 w[PR_INFLIGHT];
#pragma unroll
          for (int e = 0; e < PR_INFLIGHT; ++e) w[e] = wc[(int64_t)(i + e) * ld2];
#pragma unroll
          for (int e = 0; e < PR_INFLIGHT; e += 2) {
            a0 = fma(w[e].x, vs[i + e], a0);
            b0 = fma(w[e].y, vs[i + e], b0);
            a1 = fma(w[e + 1].x, vs[i + e + 1], a1);
            b1 = fma(w[e + 1].y, vs[i + e + 1], b1);
          }
        }
        us[2 * tid] = a0 + a1;
        us[2 * tid + 1] = b0 + b1;
      }
      __syncthreads();
      cc[2 * tid] = al_i * fi;
      cc[2 * tid + 1] = us[tid] * fi;
    }
    __syncthreads();
    // ---- the sums over the points: gm_t = sum_k alpha_k f_k (x_t - X_kt), gv_t likewise with u_k; |v|^2 and k* . alpha ride along
    {
      const double xt = xs[gt];
      double gm = 0.0, gv = 0.0, s2 = 0.0, mm = 0.0;
      for (int k = grp; k < NP; k += groups) {
        const double df = xt - Xs[(int64_t)k * xld + gt];
        gm = fma(cc[2 * k], df, gm);
        gv = fma(cc[2 * k + 1], df, gv);
        s2 += pp[2 * k];
        mm += pp[2 * k + 1];
      }
      double* r = red + grp * (2 * DP + 2);
      r[gt] = gm;
      r[DP + gt] = gv;
      if (gt == 0) { r[2 * DP] = s2; r[2 * DP + 1] = mm; }
    }
    __syncthreads();
    // ---- wave 0, lane = variable: the groups in order, mean / deviation / acquisition, the optimiser's step
    if (wave == 0) {
      double sa = 0.0, sb = 0.0, tot0 = 0.0, tot1 = 0.0;
      const int t = lane < DP ? lane : 0;
      for (int j = 0; j < groups; ++j) {
        const double* r = red + j * (2 * DP + 2);
        sa += r[t];
        sb += r[DP + t];
        tot0 += r[2 * DP];
        tot1 += r[2 * DP + 1];
      }
      double var = 1.0 - tot0;
      if (var < 0.0) {
        if (lane == 0) *a.negvar = 1;
        var = 0.0;
      }
      const double sdn = sqrt(var);
      const double sd = sqrt(var * (a.y_std * a.y_std));
      const double mu = a.y_std * tot1 + a.y_mean;
      double dmu = 0.0, dsd = 0.0, g = 0.0;
      double av, ca, cs;
      polish_acq_coeffs(a.acq, a.acq_param, a.y_max, mu, sd, DevCdf(), DevPdf(), av, ca, cs);
      if (lane < d) {
        const double inv_l = 1.0 / ls_s[lane];
        dmu = a.y_std * sa * inv_l;
        dsd = (sdn > 0.0) ? -(a.y_std * sb * inv_l) / sdn : 0.0;       // a clipped (zero) variance has no slope
        g = polish_acq_grad(ca, cs, dmu, dsd);
        if (!__builtin_isfinite(g)) g = 0.0;
      }
      if (a.eval_only) {
        if (round + 1 >= a.eval_only) {
          double* o = a.dbg + (size_t)sidx * (4 + 3 * d);
          if (lane == 0) { o[0] = -av; o[1] = mu; o[2] = sd; o[3] = 0.0; }
          if (lane < d) { o[4 + lane] = g; o[4 + d + lane] = dmu; o[4 + 2 * d + lane] = dsd; }
          if (lane == 0) flag[0] = 1;
        }
      } else {
        pr_advance(run, -av, g, d, lane, opt, a.max_iter);
        if (lane < DP) xs[lane] = (lane < d) ? run.xt / ls_s[lane] : 0.0;
        if (lane == 0) flag[0] = (run.phase == 2 || round + 1 > round_cap) ? 1 : 0;
      }
    }
    __syncthreads();
    if (flag[0]) break;
  }
  if (a.eval_only) return;
  if (wave == 0) {
    if (lane < d) a.x_out[(size_t)sidx * d + lane] = run.x;
    if (lane == 0) {
      a.f_out[sidx] = run.f;
      a.status_out[sidx] = run.phase == 2 ? run.status : 2;
      a.iter_out[sidx] = run.iter;
      a.eval_out[sidx] = run.evals;
    }
  }
}

}  // namespace

// Largest padded size the one launch serves with W in memory (debug build: GPBO_POLISH_FUSED_MAX_NP read per call — the crossover
// against the lockstep rounds, scripts/r06_polish_fused_ab.py: at N = 512 a run of ~50 evaluations already loses to them, 2.46
// against 2.13 ms — and 0 = never, the lockstep path alone: the checker's switch, with GPBO_POLISH_FUSED=0)
int polish_fused_max_np() {
  int v = POLISH_FUSED_NP_DEFAULT;
  if (const char* e = dbg_env("GPBO_POLISH_FUSED_MAX_NP")) v = atoi(e);
  return v > PR_MAX_NP ? PR_MAX_NP : v;
}

// W in LDS for NP <= 128 (whenever the image fits: 1), streamed from memory above (2), 0 = not served
static int polish_rows_mode(const Model& m) {
  const int cap = polish_fused_max_np();
  if (m.NP > cap) return 0;
  if (m.NP <= PR_LDS_NP && (size_t)pr_lds_doubles((int)m.NP, m.d, m.DP, true) * sizeof(double) + 16 <= (size_t)160 * 1024) return 1;
  if (m.NP <= PR_MAX_NP && (size_t)pr_lds_doubles((int)m.NP, m.d, m.DP, false) * sizeof(double) + 16 <= (size_t)160 * 1024) return 2;
  return 0;
}

size_t polish_fused_lds_bytes(const Model& m) {
  return (size_t)pr_lds_doubles((int)m.NP, m.d, m.DP, polish_rows_mode(m) == 1) * sizeof(double) + 16;
}

bool polish_fused_serves(const Model& m) { return polish_rows_mode(m) != 0; }

// pinned block (device-visible): doubles [seeds (S, d) | lo (d) | hi (d) | x (S, d) | f (S) | dbg (S, 4 + 3 d)] then ints
// [status (S) | iter (S) | evals (S)]
size_t polish_fused_pinned_bytes(int n_seeds, int d) {
  return ((size_t)2 * n_seeds * d + 2 * (size_t)d + (size_t)n_seeds + (size_t)n_seeds * (4 + 3 * (size_t)d)) * sizeof(double) +
         (size_t)3 * n_seeds * sizeof(int);
}

int launch_polish_fused(gpbo_ctx* ctx, Model& m, int acq, double acq_param, double y_max, double y_mean, double y_std, const double* seeds,
                        int n_seeds, const double* box_lo, const double* box_hi, int max_iter, int eval_repeat, double* host_block,
                        double* dev_block) {
  const int d = m.d;
  if (!(ctx->func_attrs & ATTR_POLISH_FUSED)) {
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(polish_rows_kernel<GPBO_KERNEL_MATERN25, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(polish_rows_kernel<GPBO_KERNEL_RBF, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(polish_rows_kernel<GPBO_KERNEL_MATERN25, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(polish_rows_kernel<GPBO_KERNEL_RBF, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    ctx->func_attrs |= ATTR_POLISH_FUSED;
  }
  const size_t S = (size_t)n_seeds;
  double* h = host_block;
  std::copy(seeds, seeds + S * d, h);
  std::copy(box_lo, box_lo + d, h + S * d);
  std::copy(box_hi, box_hi + d, h + S * d + d);
  PolishFusedArgs a{};
  a.W = m.W; a.Xs = m.Xs; a.alpha = m.alpha; a.ls = m.ls;
  a.NP = (int)m.NP; a.N = (int)m.N; a.d = d; a.DP = m.DP;
  a.y_mean = y_mean; a.y_std = y_std;
  a.acq = acq; a.acq_param = acq_param; a.y_max = y_max;
  a.max_iter = max_iter; a.eval_only = eval_repeat;
  double* dv = dev_block;
  a.seeds = dv; a.lo = dv + S * d; a.hi = dv + S * d + d;
  a.x_out = dv + S * d + 2 * d;
  a.f_out = a.x_out + S * d;
  a.dbg = a.f_out + S;
  int* iv = (int*)(a.dbg + S * (4 + 3 * (size_t)d));
  a.status_out = iv; a.iter_out = iv + S; a.eval_out = iv + 2 * S;
  a.negvar = ctx->negvar;
  const size_t lds = polish_fused_lds_bytes(m);
  const int rows_mode = polish_rows_mode(m);
  const dim3 grid((unsigned)n_seeds), rows_block((unsigned)m.NP);
  if (rows_mode == 2) {
    // the transposed copy for the row walk lives in the slot's K buffer (a fit assembles K straight into L; gpbo_get_K and the LML
    // path, which write K, invalidate it): made once per fit
    if (!m.wt_valid) {
      transpose_w_kernel<<<dim3((unsigned)(m.NP / 64), (unsigned)(m.NP / 64)), dim3(256), 0, ctx->stream>>>(m.W, m.K, (int)m.NP);
      GPBO_HIP(ctx, hipGetLastError());
      m.wt_valid = true;
    }
    a.Wt = m.K;
    if (m.kernel == GPBO_KERNEL_MATERN25) polish_rows_kernel<GPBO_KERNEL_MATERN25, false><<<grid, rows_block, lds, ctx->stream>>>(a);
    else polish_rows_kernel<GPBO_KERNEL_RBF, false><<<grid, rows_block, lds, ctx->stream>>>(a);
  } else if (rows_mode == 1) {
    if (m.kernel == GPBO_KERNEL_MATERN25) polish_rows_kernel<GPBO_KERNEL_MATERN25, true><<<grid, rows_block, lds, ctx->stream>>>(a);
    else polish_rows_kernel<GPBO_KERNEL_RBF, true><<<grid, rows_block, lds, ctx->stream>>>(a);
  } else {
    GPBO_FAIL(ctx, GPBO_ERR_INVALID, "launch_polish_fused: the model is outside the one-launch path's range");
  }
  GPBO_HIP(ctx, hipGetLastError());
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GPBO_OK;
}

}  // namespace gpbo
