// The optimiser of gpbo_polish_seeds: the host's walk of it (polish.hip: the runs advance in lockstep, a batched device
// evaluation per round) and everything the device's walk shares with it — constants, min / max, the acquisition's value and
// gradient coefficients.  polish_fused.hip (one workgroup per run, evaluations and optimiser in one launch) restates the same
// algorithm with lane i of a wave owning variable i (pr_advance there follows polish_new_direction / polish_advance here
// decision by decision); since round 6 its sums over the variables are wave reductions, not this file's left-to-right chains, so
// the two walks agree to rounding, not to the bit: tests/test_gpu_polish_fused.py holds whole searches to "the same optimum or a
// better one" and single evaluations to their error bar.
//
// A projected L-BFGS (two-loop recursion over the free variables, backtracking on the projected path with an Armijo test on
// the actual displacement), NOT a transcription of L-BFGS-B: no generalised Cauchy point, no subspace minimisation.  It keeps
// L-BFGS-B's stopping rule as SciPy configures it for `minimize` (m = 10 corrections, projected-gradient tolerance 1e-5,
// relative reduction 1e7 * eps, 20 line-search steps, 15000 iterations), and its iterates are always inside the box.  What it
// replaces: the optimiser inside AcquisitionFunction._smart_minimize (bayes_opt/acquisition.py:364-374,
// scipy.optimize.minimize(method="L-BFGS-B")); parity is statistical.
//
// Every function is plain sequential arithmetic on arrays the caller owns, in one fixed order and with floating-point
// contraction OFF: a host run is reproducible to the bit whatever the compiler (x86-64 has no fused multiply-add in its
// baseline).  (A first device version ran these very functions on one thread over LDS: correct, and 27 us per evaluation of pure
// LDS latency; round 5's lane-parallel version kept this file's summation order with v_readlane chains, 6 us per step; round 6's
// uses DPP row sums — docs/LAB_NOTEBOOK.md §9.7, §10.4.)
// First step: as L-BFGS-B, 1 / |g| without curvature information, 1 otherwise.  Over the 660 runs of the sweep
// (profiles/r06_polish_sweep.json) a run is 23.2 evaluations for 18.5 iterations: the line search costs 3.7 evaluations a run in
// all, which bounds what any other first-step rule could save.
#pragma once
#include <cstddef>

#include "gpbo_internal.h"

namespace gpbo {

constexpr int LBFGS_M = 10;
constexpr double POLISH_PGTOL = 1e-5;
constexpr double POLISH_FTOL = 1e7 * 2.220446049250313e-16;
constexpr int POLISH_MAXLS = 20;

#ifdef __HIPCC__
#define GPBO_HD __host__ __device__ __forceinline__
#else
#define GPBO_HD inline
#endif

struct PolishRun {
  int d = 0;
  double *x = nullptr, *g = nullptr, *xt = nullptr, *dir = nullptr, *q = nullptr;   // d each
  double *S = nullptr, *Y = nullptr;                                                  // LBFGS_M rows of d
  double *a = nullptr, *rho = nullptr;                                                // LBFGS_M each
  int *freev = nullptr, *order = nullptr;                                             // d; LBFGS_M
  double f = 0.0, alpha = 1.0;
  int hist = 0, head = 0;       // pairs stored, next slot
  int iter = 0, evals = 0, ls = 0;
  int phase = 0;                // 0: first evaluation pending, 1: line search, 2: finished
  int status = 2;               // 0: projected gradient, 1: relative reduction / no further progress, 2: iteration limit,
                                // 3: line search exhausted (SciPy: ABNORMAL_TERMINATION_IN_LNSRCH, success = False)
};
// storage a run needs: polish_run_doubles(d) doubles then polish_run_ints(d) ints
GPBO_HD int polish_run_doubles(int d) { return 5 * d + 2 * LBFGS_M * d + 2 * LBFGS_M; }
GPBO_HD int polish_run_ints(int d) { return d + LBFGS_M; }
GPBO_HD void polish_run_bind(PolishRun& r, int d, double* dbl, int* ints) {
  r.d = d;
  r.x = dbl; r.g = dbl + d; r.xt = dbl + 2 * d; r.dir = dbl + 3 * d; r.q = dbl + 4 * d;
  r.S = dbl + 5 * d; r.Y = r.S + LBFGS_M * d;
  r.a = r.Y + LBFGS_M * d; r.rho = r.a + LBFGS_M;
  r.freev = ints; r.order = ints + d;
}

// std::min / std::max as the host code always used them (the second argument wins only when strictly better)
GPBO_HD double polish_min(double a, double b) { return (b < a) ? b : a; }
GPBO_HD double polish_max(double a, double b) { return (a < b) ? b : a; }

// max_i |P(x - g)_i - x_i|
GPBO_HD double polish_projected_gradient_norm(const PolishRun& r, const double* lo, const double* hi) {
#pragma clang fp contract(off)
  double m = 0.0;
  for (int i = 0; i < r.d; ++i) {
    const double t = polish_min(polish_max(r.x[i] - r.g[i], lo[i]), hi[i]) - r.x[i];
    m = polish_max(m, __builtin_fabs(t));
  }
  return m;
}

// dir = -H g over the free variables (a variable sitting on a bound with the gradient pushing outwards stays there).
// The correction pairs are restricted to the CURRENT free set before they are used (components of fixed variables are
// dropped from s and y, a pair whose restricted curvature s.y is not positive is skipped): without that the pairs of
// an earlier active set steer the step and the run needs 1.5-2x the iterations (measured against SciPy on C2 / C3).
GPBO_HD void polish_new_direction(PolishRun& r, const double* lo, const double* hi) {
#pragma clang fp contract(off)
  const int d = r.d;
  for (int i = 0; i < d; ++i)
    r.freev[i] = !((r.x[i] <= lo[i] && r.g[i] > 0.0) || (r.x[i] >= hi[i] && r.g[i] < 0.0));
  for (int i = 0; i < d; ++i) r.q[i] = r.freev[i] ? r.g[i] : 0.0;
  int used = 0;        // newest first
  double gamma = 1.0;
  for (int t = 0; t < r.hist; ++t) {
    const int k = (r.head - 1 - t + 2 * LBFGS_M) % LBFGS_M;
    const double* s = r.S + (size_t)k * d;
    const double* y = r.Y + (size_t)k * d;
    double sy = 0.0, yy = 0.0;
    for (int i = 0; i < d; ++i)
      if (r.freev[i]) { sy += s[i] * y[i]; yy += y[i] * y[i]; }
    if (!(sy > 2.2e-16 * yy) || !(yy > 0.0)) continue;
    if (used == 0) gamma = sy / yy;
    r.rho[used] = 1.0 / sy;
    r.order[used++] = k;
  }
  for (int t = 0; t < used; ++t) {
    const double* s = r.S + (size_t)r.order[t] * d;
    const double* y = r.Y + (size_t)r.order[t] * d;
    double sq = 0.0;
    for (int i = 0; i < d; ++i) if (r.freev[i]) sq += s[i] * r.q[i];
    r.a[t] = r.rho[t] * sq;
    for (int i = 0; i < d; ++i) if (r.freev[i]) r.q[i] -= r.a[t] * y[i];
  }
  for (int i = 0; i < d; ++i) r.q[i] *= gamma;
  for (int t = used - 1; t >= 0; --t) {
    const double* s = r.S + (size_t)r.order[t] * d;
    const double* y = r.Y + (size_t)r.order[t] * d;
    double yq = 0.0;
    for (int i = 0; i < d; ++i) if (r.freev[i]) yq += y[i] * r.q[i];
    const double b = r.rho[t] * yq;
    for (int i = 0; i < d; ++i) if (r.freev[i]) r.q[i] += (r.a[t] - b) * s[i];
  }
  double gd = 0.0, gn = 0.0;
  for (int i = 0; i < d; ++i) {
    r.dir[i] = r.freev[i] ? -r.q[i] : 0.0;
    gd += r.dir[i] * r.g[i];
    if (r.freev[i]) gn += r.g[i] * r.g[i];
  }
  if (!(gd < 0.0) || !__builtin_isfinite(gd)) {     // not a descent direction: steepest descent over the free variables, history dropped
    r.hist = 0;
    used = 0;
    for (int i = 0; i < d; ++i) r.dir[i] = r.freev[i] ? -r.g[i] : 0.0;
  }
  // L-BFGS-B takes a unit step except when it has no curvature information, where it starts from 1 / |d|
  r.alpha = (used == 0) ? polish_min(1.0, 1.0 / __builtin_sqrt(polish_max(gn, 1e-300))) : 1.0;
  r.ls = 0;
}

GPBO_HD void polish_trial_point(PolishRun& r, const double* lo, const double* hi) {
#pragma clang fp contract(off)
  for (int i = 0; i < r.d; ++i) r.xt[i] = polish_min(polish_max(r.x[i] + r.alpha * r.dir[i], lo[i]), hi[i]);
}

// the run's first request: the seed, clipped into the box
GPBO_HD void polish_start(PolishRun& r, const double* seed, const double* lo, const double* hi) {
  const int d = r.d;
  for (int i = 0; i < d; ++i) {
    r.x[i] = 0.0; r.g[i] = 0.0; r.dir[i] = 0.0; r.q[i] = 0.0;
    r.xt[i] = polish_min(polish_max(seed[i], lo[i]), hi[i]);
  }
  for (int i = 0; i < LBFGS_M * d; ++i) { r.S[i] = 0.0; r.Y[i] = 0.0; }
  r.f = 0.0; r.alpha = 1.0;
  r.hist = 0; r.head = 0; r.iter = 0; r.evals = 0; r.ls = 0; r.phase = 0; r.status = 2;
}

// one answer (f_t, g_t at r.xt; non-finite gradient components already replaced by 0) of the objective; leaves the next request in
// r.xt unless the run has finished
GPBO_HD void polish_advance(PolishRun& r, double ft, const double* gt, const double* lo, const double* hi, int max_iter) {
#pragma clang fp contract(off)
  const int d = r.d;
  ++r.evals;
  if (r.phase == 0) {
    for (int i = 0; i < d; ++i) { r.x[i] = r.xt[i]; r.g[i] = gt[i]; }
    r.f = ft;
    if (!__builtin_isfinite(ft)) { r.phase = 2; r.status = 2; return; }
    if (polish_projected_gradient_norm(r, lo, hi) <= POLISH_PGTOL) { r.phase = 2; r.status = 0; return; }
    polish_new_direction(r, lo, hi);
    polish_trial_point(r, lo, hi);
    r.phase = 1;
    return;
  }
  double gs = 0.0, moved = 0.0;       // g . (x_t - x): the Armijo test on the displacement the projection left
  for (int i = 0; i < d; ++i) {
    const double s = r.xt[i] - r.x[i];
    gs += r.g[i] * s;
    moved = polish_max(moved, __builtin_fabs(s));
  }
  const bool ok = __builtin_isfinite(ft) && ft <= r.f + 1e-4 * gs;
  if (!ok) {
    // no further progress along this path (x stays): the step no longer moves x, the line search is exhausted, or — after
    // two shrinks — the values differ by less than the relative-reduction tolerance, i.e. the test is deciding on rounding
    const bool flat = __builtin_isfinite(ft) && r.ls >= 2 &&
                      __builtin_fabs(ft - r.f) <= POLISH_FTOL * polish_max(polish_max(__builtin_fabs(ft), __builtin_fabs(r.f)), 1.0);
    // (an exhausted line search is SciPy's "ABNORMAL" termination, success = False: the reference discards such a run,
    //  acquisition.py:367 — its own status, so that the caller can do the same)
    if (moved == 0.0 || flat) { r.phase = 2; r.status = 1; return; }
    if (++r.ls >= POLISH_MAXLS) { r.phase = 2; r.status = 3; return; }
    // the minimiser of the parabola through f, its slope and f_t, kept inside [0.1, 0.5] of the step that failed
    double shrink = 0.1;
    if (__builtin_isfinite(ft)) {
      const double curv = ft - r.f - gs;
      shrink = curv > 0.0 ? polish_min(polish_max(-gs / (2.0 * curv), 0.1), 0.5) : 0.5;
    }
    r.alpha *= shrink;
    polish_trial_point(r, lo, hi);
    return;
  }
  // accepted
  {
    double* s = r.S + (size_t)r.head * d;
    double* y = r.Y + (size_t)r.head * d;
    double sy = 0.0, yy = 0.0;
    for (int i = 0; i < d; ++i) {
      s[i] = r.xt[i] - r.x[i];
      y[i] = gt[i] - r.g[i];
      sy += s[i] * y[i];
      yy += y[i] * y[i];
    }
    if (sy > 2.2e-16 * yy && yy > 0.0) {          // L-BFGS-B's curvature test (repeated on the free set when the pair is used)
      r.head = (r.head + 1) % LBFGS_M;
      r.hist = (r.hist + 1 < LBFGS_M) ? r.hist + 1 : LBFGS_M;
    }
  }
  const double f_old = r.f;
  for (int i = 0; i < d; ++i) { r.x[i] = r.xt[i]; r.g[i] = gt[i]; }
  r.f = ft;
  ++r.iter;
  if (polish_projected_gradient_norm(r, lo, hi) <= POLISH_PGTOL) { r.phase = 2; r.status = 0; return; }
  if ((f_old - ft) <= POLISH_FTOL * polish_max(polish_max(__builtin_fabs(f_old), __builtin_fabs(ft)), 1.0)) { r.phase = 2; r.status = 1; return; }
  if (r.iter >= max_iter) { r.phase = 2; r.status = 2; return; }
  polish_new_direction(r, lo, hi);
  polish_trial_point(r, lo, hi);
}

// f = -acq(mu, sd) and its gradient from the posterior and ITS gradient, unconstrained (acquisition.py:198-217, 485, 660-661,
// 847-849): g_i = -(ca dmu_i + cs dsd_i).  CDF / PDF are the caller's (std::erfc / std::exp on the host, the device library's on
// the device: EI and POI values agree to rounding between the two sides, UCB bit for bit).
template <class Cdf, class Pdf>
GPBO_HD void polish_acq_coeffs(int acq, double acq_param, double y_max, double mu, double sd, Cdf&& cdf_of, Pdf&& pdf_of, double& a,
                               double& ca, double& cs) {
#pragma clang fp contract(off)
  if (acq == GPBO_ACQ_UCB) {
    a = mu + acq_param * sd; ca = 1.0; cs = acq_param;
  } else {
    const double aa = mu - y_max - acq_param;
    const double z = aa / sd;
    const double cdf = cdf_of(z), pdf = pdf_of(z);
    if (acq == GPBO_ACQ_EI) { a = aa * cdf + sd * pdf; ca = cdf; cs = pdf; }
    else { a = cdf; ca = pdf / sd; cs = -pdf * z / sd; }
  }
}
GPBO_HD double polish_acq_grad(double ca, double cs, double dmu, double dsd) {
#pragma clang fp contract(off)
  return -(ca * dmu + cs * dsd);
}

}  // namespace gpbo
