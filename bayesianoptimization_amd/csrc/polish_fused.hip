// gpbo_polish_seeds as ONE launch: one workgroup per local search, the evaluations AND the optimiser inside it.
//
// What it replaces: the local-search stage of AcquisitionFunction._smart_minimize (bayes_opt/acquisition.py:322-420) for the
// sizes where it is launch latency and nothing else.  polish.hip advances all runs in lockstep, one batched evaluation per round:
// six dependent launches and a stream synchronisation (43-50 us at N <= 256) for ~2 N^2 flops per run.  Here a run is a workgroup
// that owns its search from the seed to the stopping rule.  Two kernels, one optimiser (polish_opt.h's, laid over wave 0 with a
// lane per variable):
//
//   polish_rows_kernel (round 6; NP <= 128): THREAD = TRAINING POINT.  W = L^-1 sits in LDS as a padded square ([NP][NP + 1]:
//     a thread walks its row for v = W k*, its column for u = W^T v, both conflict-free), thread i keeps k*_i, v_i, u_i of its own
//     point, the sums over the points (mu, |v|^2, the 2 d gradient sums) are taken by lane groups of one dimension each and
//     combined in a fixed order, the optimiser's sums over the variables are DPP row reductions.  Four barriers of a one- or
//     two-wave workgroup per evaluation: ~2 us per evaluation + step where the kernel below needs 13-19 (profiles/r06_polish_fused_ab.json).
//     Deterministic, and NOT the bits of the lockstep path: the evaluation agrees with gpbo_predict_grad to ~1e-13 of the
//     values' scale (tests/test_gpu_polish_fused.py), a whole search ends at the same or a better value (SURVEY.md section 8 f2:
//     "parity is statistical (same or better acquisition value), not bit-wise"; the lockstep path is the checker).
//   polish_fused_kernel (round 5; 128 < NP <= 256): eight waves, W read twice per evaluation from L2, the arithmetic of
//     launch_posterior_grad_small (posterior_small.hip) phase by phase with the same loops, accumulation orders and reduction
//     trees, the optimiser's sums as the host's left-to-right chains (v_readlane + add): bitwise the lockstep path for UCB.  From
//     N ~ 300 on one CU's load rate makes an evaluation slower than the six launches that spread W over the chip, which is where
//     polish_fused_max_np() stops it.
// One model (no constraint slots).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "gpbo_internal.h"
#include "polish_opt.h"

namespace gpbo {

namespace {

constexpr int PF_THREADS = 512;     // 8 waves, two per SIMD: 256 VGPRs each (16 waves spilled ~100 registers in the v = W k* phase)
constexpr int PF_SPLITS = 16;    // = GRAD_SPLITS of posterior_small.hip (NP < 2048)
constexpr int PF_KSL = 16;       // = GRAD_KSL
constexpr int PF_FIXED = 64 * 3 + 16;   // xs, xt, ls [64 each] + sh[16]
constexpr int PF_XS_STAGE = 8192 + 512;   // training points staged in LDS when NP * (DP + 1) doubles fit in here

struct PolishFusedArgs {
  const double *W, *Xs, *alpha, *ls;
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

// (the row splits of u and, once u is summed, the k-lane partials of the gradient sums share one region)
__host__ __device__ inline int pf_shared_region(int NP) { return PF_SPLITS * NP > PF_KSL * 2 * 256 ? PF_SPLITS * NP : PF_KSL * 2 * 256; }
__host__ __device__ inline int pf_lds_base(int NP, int d, int DP) {
  return PF_FIXED + 5 * NP + pf_shared_region(NP) + PF_KSL * 2 * DP + 2 * LBFGS_M * d;
}
constexpr int PF_LDS_CAP = 160 * 128 - 8;      // doubles in 160 KiB, less the flag words
// (the training points: staged only where the whole image still fits — at NP > 512 the row splits take the room)
__host__ __device__ inline int pf_xs_stage(int NP, int d, int DP) {
  const int want = NP * (DP + 1);
  return (want <= PF_XS_STAGE && pf_lds_base(NP, d, DP) + want <= PF_LDS_CAP) ? want : 0;
}
__host__ __device__ inline int pf_lds_doubles(int NP, int d, int DP) {
  return pf_lds_base(NP, d, DP) + pf_xs_stage(NP, d, DP);
}
__host__ __device__ inline int pf_lds_ints(int) { return 4; }

// ---- the optimiser over wave 0 -----------------------------------------------------------------------------------------
// polish_opt.h's arithmetic with lane i owning variable i (d <= 64 = one wave; lanes >= d carry zeros and are never read).  Every
// sum over the variables is the host's left-to-right chain, formed from v_readlane'd addends (a skipped addend on the host is a
// + 0.0 here: the chains start at + 0.0, so no partial sum is ever -0.0 and x + 0.0 = x).  Scalars are computed by all lanes alike.
// The correction pairs live in LDS ([LBFGS_M][d], lane i reads and writes column i: program order within one wave); the per-pair
// scalars rho_t, a_t and the slot order sit in lanes t of one register each.
struct WaveRun {
  double x, g, xt, dir, q, lo, hi;     // this lane's variable
  bool freev;
  double rho, av;                      // lane t: rho_t, a_t of the t-th usable pair (newest first)
  int order;                           // lane t: its slot
  double f, alpha;
  int hist, head, iter, evals, ls, phase, status;
};

__device__ __forceinline__ double pf_lane(double v, int i) {      // v of lane i (i uniform), in every lane
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), i);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), i);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double pf_sum(double v, int d) {       // ((0 + v_0) + v_1) + ...
#pragma clang fp contract(off)
  double acc = 0.0;
  for (int i = 0; i < d; ++i) acc += pf_lane(v, i);
  return acc;
}
__device__ __forceinline__ double pf_max_abs(double v, int d) {   // m = 0; m = max(m, |v_i|) in order (std::max: a NaN never wins)
  double m = 0.0;
  for (int i = 0; i < d; ++i) m = polish_max(m, __builtin_fabs(pf_lane(v, i)));
  return m;
}

__device__ __forceinline__ double pf_projected_gradient_norm(const WaveRun& r, int d) {
#pragma clang fp contract(off)
  const double t = polish_min(polish_max(r.x - r.g, r.lo), r.hi) - r.x;
  return pf_max_abs(t, d);
}

__device__ __forceinline__ void pf_trial_point(WaveRun& r) {
#pragma clang fp contract(off)
  r.xt = polish_min(polish_max(r.x + r.alpha * r.dir, r.lo), r.hi);
}

__device__ __forceinline__ void pf_new_direction(WaveRun& r, int d, int lane, const double* S, const double* Y) {
#pragma clang fp contract(off)
  const bool mine = lane < d;
  r.freev = mine && !((r.x <= r.lo && r.g > 0.0) || (r.x >= r.hi && r.g < 0.0));
  r.q = r.freev ? r.g : 0.0;
  int used = 0;        // newest first
  double gamma = 1.0;
  for (int t = 0; t < r.hist; ++t) {
    const int k = (r.head - 1 - t + 2 * LBFGS_M) % LBFGS_M;
    const double s = mine ? S[k * d + lane] : 0.0, y = mine ? Y[k * d + lane] : 0.0;
    const double sy = pf_sum(r.freev ? s * y : 0.0, d), yy = pf_sum(r.freev ? y * y : 0.0, d);
    if (!(sy > 2.2e-16 * yy) || !(yy > 0.0)) continue;
    if (used == 0) gamma = sy / yy;
    if (lane == used) {
      r.rho = 1.0 / sy;
      r.order = k;
    }
    ++used;
  }
  for (int t = 0; t < used; ++t) {
    const int k = __builtin_amdgcn_readlane(r.order, t);
    const double s = mine ? S[k * d + lane] : 0.0, y = mine ? Y[k * d + lane] : 0.0;
    const double sq = pf_sum(r.freev ? s * r.q : 0.0, d);
    const double at = pf_lane(r.rho, t) * sq;
    if (lane == t) r.av = at;
    if (r.freev) r.q -= at * y;
  }
  r.q *= gamma;
  for (int t = used - 1; t >= 0; --t) {
    const int k = __builtin_amdgcn_readlane(r.order, t);
    const double s = mine ? S[k * d + lane] : 0.0, y = mine ? Y[k * d + lane] : 0.0;
    const double yq = pf_sum(r.freev ? y * r.q : 0.0, d);
    const double bt = pf_lane(r.rho, t) * yq;
    if (r.freev) r.q += (pf_lane(r.av, t) - bt) * s;
  }
  r.dir = r.freev ? -r.q : 0.0;
  const double gd = pf_sum(r.dir * r.g, d), gn = pf_sum(r.freev ? r.g * r.g : 0.0, d);
  if (!(gd < 0.0) || !__builtin_isfinite(gd)) {     // not a descent direction: steepest descent over the free variables, history dropped
    r.hist = 0;
    used = 0;
    r.dir = r.freev ? -r.g : 0.0;
  }
  r.alpha = (used == 0) ? polish_min(1.0, 1.0 / __builtin_sqrt(polish_max(gn, 1e-300))) : 1.0;
  r.ls = 0;
}

// polish_advance: one answer (ft, this lane's gradient component gt — non-finite components already 0) of the objective
__device__ __forceinline__ void pf_advance(WaveRun& r, double ft, double gt, int d, int lane, double* S, double* Y, int max_iter) {
#pragma clang fp contract(off)
  const bool mine = lane < d;
  ++r.evals;
  if (r.phase == 0) {
    r.x = r.xt; r.g = gt; r.f = ft;
    if (!__builtin_isfinite(ft)) { r.phase = 2; r.status = 2; return; }
    if (pf_projected_gradient_norm(r, d) <= POLISH_PGTOL) { r.phase = 2; r.status = 0; return; }
    pf_new_direction(r, d, lane, S, Y);
    pf_trial_point(r);
    r.phase = 1;
    return;
  }
  const double sd = r.xt - r.x;
  const double gs = pf_sum(r.g * sd, d), moved = pf_max_abs(sd, d);
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
    pf_trial_point(r);
    return;
  }
  {   // accepted
    const double s = r.xt - r.x, y = gt - r.g;
    if (mine) {
      S[r.head * d + lane] = s;
      Y[r.head * d + lane] = y;
    }
    const double sy = pf_sum(s * y, d), yy = pf_sum(y * y, d);
    if (sy > 2.2e-16 * yy && yy > 0.0) {
      r.head = (r.head + 1) % LBFGS_M;
      r.hist = (r.hist + 1 < LBFGS_M) ? r.hist + 1 : LBFGS_M;
    }
  }
  const double f_old = r.f;
  r.x = r.xt; r.g = gt; r.f = ft;
  ++r.iter;
  if (pf_projected_gradient_norm(r, d) <= POLISH_PGTOL) { r.phase = 2; r.status = 0; return; }
  if ((f_old - ft) <= POLISH_FTOL * polish_max(polish_max(__builtin_fabs(f_old), __builtin_fabs(ft)), 1.0)) { r.phase = 2; r.status = 1; return; }
  if (r.iter >= max_iter) { r.phase = 2; r.status = 2; return; }
  pf_new_direction(r, d, lane, S, Y);
  pf_trial_point(r);
}

template <int KERNEL>
__global__ __launch_bounds__(PF_THREADS) void polish_fused_kernel(const PolishFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) double pf_smem[];
  const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int sidx = (int)blockIdx.x;
  const int NP = a.NP, N = a.N, d = a.d, DP = a.DP;
  const double* __restrict__ W = a.W;

  double* xs = pf_smem;                 // [64] the trial point / length scale, zero padded
  double* xt_s = xs + 64;               // [64] the trial point
  double* ls_s = xt_s + 64;             // [64] length scales
  double* sh = ls_s + 64;               // [16] 0..3: v.v by wave, 4..7: k*.alpha by wave, 8: f, 9: mu, 10: sd
  double* ks = sh + 16;                 // [NP]
  double* fs = ks + NP;
  double* vv = fs + NP;
  double* uu = vv + NP;
  double* al_s = uu + NP;               // [NP] alpha
  double* partial = al_s + NP;          // [PF_SPLITS][NP], then (u summed)
  double* gs = partial;                 // [PF_KSL][2][256]
  double* gpart = partial + pf_shared_region(NP);   // [PF_KSL][2][DP]
  double* Sh = gpart + PF_KSL * 2 * DP;              // [LBFGS_M][d] correction pairs
  double* Yh = Sh + LBFGS_M * d;
  double* Xl = Yh + LBFGS_M * d;                     // [NP][DP + 1] training points (when staged)
  const int xs_staged = pf_xs_stage(NP, d, DP);
  int* flag = (int*)(Xl + xs_staged);
  auto w_mem = [&](int i, int k) -> double { return W[(int64_t)i * NP + k]; };
  // the training points: LDS rows of DP + 1 (conflict-free for the row walk of P1 and the dimension walk of P5), else global rows
  const double* __restrict__ Xs = xs_staged ? Xl : a.Xs;
  const int xld = xs_staged ? DP + 1 : DP;

  for (int k = tid; k < NP; k += PF_THREADS) al_s[k] = a.alpha[k];
  if (tid < 64) ls_s[tid] = (tid < d) ? a.ls[tid] : 1.0;
  if (xs_staged)
    for (int e = tid; e < NP * DP; e += PF_THREADS) {
      const int k = e / DP, t = e - k * DP;
      Xl[k * (DP + 1) + t] = a.Xs[e];
    }
  WaveRun run{};
  if (wave == 0) {
    const bool mine = lane < d;
    run.lo = mine ? a.lo[lane] : 0.0;
    run.hi = mine ? a.hi[lane] : 0.0;
    run.x = 0.0; run.g = 0.0; run.dir = 0.0; run.q = 0.0; run.freev = false; run.rho = 0.0; run.av = 0.0; run.order = 0;
    run.xt = mine ? polish_min(polish_max(a.seeds[(size_t)sidx * d + lane], run.lo), run.hi) : 0.0;      // polish_start
    run.f = 0.0; run.alpha = 1.0;
    run.hist = 0; run.head = 0; run.iter = 0; run.evals = 0; run.ls = 0; run.phase = 0; run.status = 2;
    if (mine) xt_s[lane] = run.xt;
    // the trial point over the length scales (prescale_elem), zero padded: written here and after every step of the optimiser
    if (lane < DP) xs[lane] = mine ? run.xt / a.ls[lane] : 0.0;
    if (lane == 0) flag[0] = 0;
  }
  __syncthreads();

  const int t256 = tid & 255, grp = tid >> 8;
  const int gt_dim = t256 % DP, kl = t256 / DP, nkl = 256 / DP;
  const int per = (NP + PF_KSL - 1) / PF_KSL;
  const int round_cap = 4 * a.max_iter + 64;

  for (int round = 0;; ++round) {
    // ---- P1: k* and the gradient factor f (kstar_grad_small_kernel)
    for (int k = tid; k < NP; k += PF_THREADS) {
      const double* xr = Xs + (int64_t)k * xld;
      double d2 = 0.0;
      for (int t = 0; t < DP; ++t) {
        const double df = xs[t] - xr[t];
        d2 = fma(df, df, d2);
      }
      const double kv = gpbo_kernel_value<KERNEL>(d2);
      double f;
      if (KERNEL == GPBO_KERNEL_MATERN25) {
        const double s = gpbo_sqrt_pos(d2) * 2.23606797749978969641;      // sqrt(5) r
        f = -1.66666666666666666667 * (1.0 + s) * gpbo_exp_nonpos(-s);
      } else {
        f = -kv;
      }
      ks[k] = kv;
      fs[k] = (k < N) ? f : 0.0;      // padding rows carry no gradient
    }
    __syncthreads();
    // ---- P2: v = W k* (gemv_small_v_kernel<8, 2>: rows in pairs, lane = k mod 64, ascending k, xor tree).  Four pairs and two
    // k-steps per turn: 16 loads in flight per lane.  Loads past a pair's last k (or past the last pair) are clamped into the
    // matrix and not accumulated.
    auto phase_v = [&](auto w_at) {
      const int npairs = NP >> 1;
      for (int p0 = wave * 4; p0 < npairs; p0 += (PF_THREADS / 64) * 4) {
        double acc[4][2];
        int kmax[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[q][0] = 0.0; acc[q][1] = 0.0;
          kmax[q] = (p0 + q < npairs) ? min(NP, 2 * (p0 + q) + 2) : 0;
        }
        const int kend = min(NP, 2 * min(p0 + 3, npairs - 1) + 2);
        for (int kb = 0; kb < kend; kb += 128) {
          double w[4][2][2], kv[2];
          int kk[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            kk[e] = kb + 64 * e + lane;
            const int kc = min(kk[e], NP - 1);
            kv[e] = ks[kc];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int i0 = 2 * min(p0 + q, npairs - 1);
              w[q][0][e] = w_at(i0, kc);
              w[q][1][e] = w_at(i0 + 1, kc);
            }
          }
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (kk[e] < kmax[q]) {
                acc[q][0] = fma(w[q][0][e], kv[e], acc[q][0]);
                acc[q][1] = fma(w[q][1][e], kv[e], acc[q][1]);
              }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            double v = acc[q][r];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            const int i = 2 * (p0 + q) + r;
            if (lane == 0 && p0 + q < npairs) vv[i] = (i < N) ? v : 0.0;
          }
      }
    };
    phase_v(w_mem);
    __syncthreads();
    // ---- P3: the row splits of u = W^T v (gemvt_small_kernel<4>: per 64-column block the rows below it in PF_SPLITS chunks, four
    // row lanes i = r0 + ig (mod 4) each summed by itself and added in order)
    auto phase_u = [&](auto w_at) {
    for (int it = tid; it < PF_SPLITS * NP; it += PF_THREADS) {
      const int sp = it / NP, j = it - sp * NP;
      const int j0 = j & ~63;
      const int chunk = (NP - j0 + PF_SPLITS - 1) / PF_SPLITS;
      const int r0 = j0 + sp * chunk, r1 = min(NP, r0 + chunk);
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      for (int ib = r0; ib < r1; ib += 16) {      // 16 rows in flight (rows past the chunk: clamped into the matrix, not accumulated)
        double w[16], vi[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int iq = min(ib + q, NP - 1);
          w[q] = w_at(iq, j);
          vi[q] = vv[iq];
        }
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
          if (ib + q < r1) a0 = fma(w[q], vi[q], a0);
          if (ib + q + 1 < r1) a1 = fma(w[q + 1], vi[q + 1], a1);
          if (ib + q + 2 < r1) a2 = fma(w[q + 2], vi[q + 2], a2);
          if (ib + q + 3 < r1) a3 = fma(w[q + 3], vi[q + 3], a3);
        }
      }
      double sum = a0;
      sum += a1;
      sum += a2;
      sum += a3;
      partial[sp * NP + j] = sum;
    }
    };
    phase_u(w_mem);
    __syncthreads();
    // ---- P4: u_k = the splits in order (grad_small_kernel's inner sum)
    for (int k = tid; k < NP; k += PF_THREADS) {
      double u = 0.0;
#pragma unroll
      for (int sp = 0; sp < PF_SPLITS; ++sp) u += partial[sp * NP + k];
      uu[k] = u;
    }
    __syncthreads();
    // ---- P5: the two k-sums per dimension (grad_small_kernel: PF_KSL slices of train points, 256 threads = DP dimensions x k-lanes per
    // slice, the k-lanes added in order)
    for (int sl = grp; sl < PF_KSL; sl += PF_THREADS / 256) {
      const int k0 = sl * per, k1 = min(NP, k0 + per);
      const double xt = xs[gt_dim];
      double gm = 0.0, gv = 0.0;
      for (int k = k0 + kl; k < k1; k += nkl) {
        const double f = fs[k];
        const double df = (xt - Xs[(int64_t)k * xld + gt_dim]) * f;
        gm = fma(al_s[k], df, gm);
        gv = fma(uu[k], df, gv);
      }
      gs[(sl * 2 + 0) * 256 + t256] = gm;
      gs[(sl * 2 + 1) * 256 + t256] = gv;
    }
    __syncthreads();
    for (int it = tid; it < PF_KSL * DP; it += PF_THREADS) {
      const int sl = it / DP, t = it - sl * DP;
      double sa = 0.0, sb = 0.0;
      for (int q = 0; q < nkl; ++q) {
        sa += gs[(sl * 2 + 0) * 256 + q * DP + t];
        sb += gs[(sl * 2 + 1) * 256 + q * DP + t];
      }
      gpart[(sl * 2 + 0) * DP + t] = sa;
      gpart[(sl * 2 + 1) * DP + t] = sb;
    }
    // ---- P6: mean, variance and the slices in order (grad_final_kernel)
    if (tid < 256) {
      double s2 = 0.0, mm = 0.0;
      for (int i = tid; i < NP; i += 256) {
        const double vi = vv[i];
        s2 = fma(vi, vi, s2);
        mm = fma(ks[i], al_s[i], mm);
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) s2 += __shfl_xor(s2, off);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) mm += __shfl_xor(mm, off);
      if (lane == 0) {
        sh[wave] = s2;
        sh[4 + wave] = mm;
      }
    }
    __syncthreads();
    // ---- P7 and the optimiser's step: wave 0, lane = variable
    if (wave == 0) {
      const double tot0 = ((sh[0] + sh[1]) + sh[2]) + sh[3];
      const double tot1 = ((sh[4] + sh[5]) + sh[6]) + sh[7];
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
      polish_acq_coeffs(a.acq, a.acq_param, a.y_max, mu, sd, DevCdf(), DevPdf(), av, ca, cs);   // (polish_opt.h: the host's expressions)
      if (lane < d) {
        double sa = 0.0, sb = 0.0;
        for (int sl = 0; sl < PF_KSL; ++sl) {
          sa += gpart[(sl * 2 + 0) * DP + lane];
          sb += gpart[(sl * 2 + 1) * DP + lane];
        }
        const double inv_l = 1.0 / ls_s[lane];
        dmu = a.y_std * sa * inv_l;
        // d sd / d x = y_std * (-2 b / l) / (2 sqrt(var_n)); a clipped (zero) variance has no slope
        dsd = (sdn > 0.0) ? -(a.y_std * sb * inv_l) / sdn : 0.0;
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
        pf_advance(run, -av, g, d, lane, Sh, Yh, a.max_iter);
        if (lane < d) xt_s[lane] = run.xt;
        if (lane < DP) xs[lane] = (lane < d) ? run.xt / ls_s[lane] : 0.0;       // (P1 and P5 of this round are behind the barrier above)
        if (lane == 0) flag[0] = (run.phase == 2 || round + 1 > round_cap) ? 1 : 0;      // (the cap cannot bind: a run is bounded by max_iter * MAXLS)
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

// ---- the optimiser of polish_rows_kernel: the same steps (polish_opt.h), other arithmetic and a SMALL body -------------------
// Lane i owns variable i on wave 0, as above.  What differs: (a) a sum over the variables is a DPP butterfly inside the rows of 16
// lanes + the rows' totals (the chain above: 3 d instructions); (b) s.y and y.y of a correction pair are kept from the step that
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

// ---- thread = training point (NP <= 128) ---------------------------------------------------------------------------------
constexpr int PR_MAX_NP = 128;
// LDS (doubles): W [NP][NP + 1] | xs [64] | ls [64] | alpha, k*, v [NP each] | (c1, c2) [NP][2] | (v^2, k* alpha) [NP][2] |
// group partials [PR_MAX_GROUPS][2 DP + 2] | the optimiser's block (pr_opt_doubles) | X [NP][DP + 1] (when it fits) ; then the flag word
constexpr int PR_MAX_GROUPS = 32;
__host__ __device__ inline int pr_lds_base(int NP, int d, int DP) {
  return NP * (NP + 1) + 128 + 7 * NP + PR_MAX_GROUPS * (2 * DP + 2) + pr_opt_doubles(d);
}
__host__ __device__ inline int pr_xs_stage(int NP, int d, int DP) {
  const int want = NP * (DP + 1);
  return (pr_lds_base(NP, d, DP) + want <= PF_LDS_CAP) ? want : 0;
}
__host__ __device__ inline int pr_lds_doubles(int NP, int d, int DP) { return pr_lds_base(NP, d, DP) + pr_xs_stage(NP, d, DP); }

template <int KERNEL>
__global__ __launch_bounds__(PR_MAX_NP) void polish_rows_kernel(const PolishFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) double pr_smem[];
  const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int sidx = (int)blockIdx.x;
  const int NP = a.NP, N = a.N, d = a.d, DP = a.DP;
  const int WLD = NP + 1;
  double* Wl = pr_smem;                     // [NP][NP + 1]: row walks and column walks both hit 64 different banks
  double* xs = Wl + NP * WLD;               // [64] the trial point over the length scales, zero padded
  double* ls_s = xs + 64;                   // [64]
  double* al_s = ls_s + 64;                 // [NP] alpha
  double* ks = al_s + NP;                   // [NP] k*
  double* vs = ks + NP;                     // [NP] v = W k*
  double* cc = vs + NP;                     // [NP][2] alpha_k f_k, u_k f_k
  double* pp = cc + 2 * NP;                 // [NP][2] v_k^2, k*_k alpha_k
  double* red = pp + 2 * NP;                // [groups][2 DP + 2]
  double* opt = red + PR_MAX_GROUPS * (2 * DP + 2);
  double* Xl = opt + pr_opt_doubles(d);
  const int xs_staged = pr_xs_stage(NP, d, DP);
  int* flag = (int*)(Xl + xs_staged);
  const double* __restrict__ Xs = xs_staged ? Xl : a.Xs;
  const int xld = xs_staged ? DP + 1 : DP;

  for (int e = tid; e < NP * NP; e += NP) {           // (blockDim.x == NP) coalesced rows of the matrix in memory, zeros above the diagonal included
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
    // ---- v_i = sum_k W[i][k] k*_k (zeros above the diagonal: the whole row)
    {
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
    }
    __syncthreads();
    // ---- u_k = sum_i W[i][k] v_i for this thread's column, then its two gradient weights
    {
      double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
      const double* wcol = Wl + tid;
      for (int i = 0; i < NP; i += 4) {
        u0 = fma(wcol[i * WLD], vs[i], u0);
        u1 = fma(wcol[(i + 1) * WLD], vs[i + 1], u1);
        u2 = fma(wcol[(i + 2) * WLD], vs[i + 2], u2);
        u3 = fma(wcol[(i + 3) * WLD], vs[i + 3], u3);
      }
      const double u = (u0 + u1) + (u2 + u3);
      cc[2 * tid] = al_i * fi;
      cc[2 * tid + 1] = u * fi;
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

// Largest padded size the one-launch local search serves (0: never).  (Debug build: GPBO_POLISH_FUSED_MAX_NP read per call, for
// the A/B tests and the crossover measurement.)
int polish_fused_max_np() {
  int v = POLISH_FUSED_NP_DEFAULT;
  if (const char* e = dbg_env("GPBO_POLISH_FUSED_MAX_NP")) v = atoi(e);
  if (v > POLISH_FUSED_NP_CAP) v = POLISH_FUSED_NP_CAP;
  return v;
}

// thread = training point (polish_rows_kernel) for NP <= 128 whenever its LDS image fits; GPBO_POLISH_ROWS=0 (debug build, read per
// call): the eight-wave kernel there too (A/B, and the bitwise tests of that kernel at small sizes)
static bool polish_rows_serves(const Model& m) {
  const char* e = dbg_env("GPBO_POLISH_ROWS");
  if (e && e[0] == '0') return false;
  return m.NP <= PR_MAX_NP && (size_t)pr_lds_doubles((int)m.NP, m.d, m.DP) * sizeof(double) + 16 <= (size_t)160 * 1024;
}

size_t polish_fused_lds_bytes(const Model& m) {
  if (polish_rows_serves(m)) return (size_t)pr_lds_doubles((int)m.NP, m.d, m.DP) * sizeof(double) + 16;
  return (size_t)pf_lds_doubles((int)m.NP, m.d, m.DP) * sizeof(double) + (size_t)pf_lds_ints(m.d) * sizeof(int);
}

bool polish_fused_serves(const Model& m) {
  return m.NP <= polish_fused_max_np() && m.NP < 2048 && polish_fused_lds_bytes(m) <= (size_t)160 * 1024;
}

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
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(polish_fused_kernel<GPBO_KERNEL_MATERN25>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(polish_fused_kernel<GPBO_KERNEL_RBF>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(polish_rows_kernel<GPBO_KERNEL_MATERN25>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(polish_rows_kernel<GPBO_KERNEL_RBF>),
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
  if (polish_rows_serves(m)) {
    if (m.kernel == GPBO_KERNEL_MATERN25)
      polish_rows_kernel<GPBO_KERNEL_MATERN25><<<dim3((unsigned)n_seeds), dim3((unsigned)m.NP), lds, ctx->stream>>>(a);
    else
      polish_rows_kernel<GPBO_KERNEL_RBF><<<dim3((unsigned)n_seeds), dim3((unsigned)m.NP), lds, ctx->stream>>>(a);
  } else if (m.kernel == GPBO_KERNEL_MATERN25)
    polish_fused_kernel<GPBO_KERNEL_MATERN25><<<dim3((unsigned)n_seeds), dim3(PF_THREADS), lds, ctx->stream>>>(a);
  else
    polish_fused_kernel<GPBO_KERNEL_RBF><<<dim3((unsigned)n_seeds), dim3(PF_THREADS), lds, ctx->stream>>>(a);
  GPBO_HIP(ctx, hipGetLastError());
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GPBO_OK;
}

}  // namespace gpbo
