// gpbo_polish_seeds — the local-search stage of a suggest() behind the C ABI.
//
// What it replaces: AcquisitionFunction._smart_minimize for an all-continuous space (bayes_opt/acquisition.py:322-420):
// one scipy.optimize.minimize(acq, x_seed, bounds=..., method="L-BFGS-B") per seed, every function value a
// GaussianProcessRegressor.predict call and every gradient d + 1 of them (finite differences).  Round 2 moved the
// evaluations to the device (one batched launch per lockstep round, analytic gradient) and found the time unchanged:
// 4.4 of 5.6 ms at N = 512 were SciPy's setulb and Python between the launches.  Here the whole stage is one C call:
// all seeds advance in lockstep, each round is ONE batched value-and-gradient evaluation on the device
// (launch_posterior_grad per model: mu, sd and their input gradients from one k*) and a few hundred
// flops of optimiser arithmetic per seed on the host.
//
// The optimiser is a projected L-BFGS (two-loop recursion over the free variables, backtracking on the projected path
// with an Armijo test on the actual displacement), NOT a transcription of L-BFGS-B: no generalised Cauchy point, no
// subspace minimisation.  It keeps L-BFGS-B's stopping rule as SciPy configures it for `minimize` (m = 10 corrections,
// projected-gradient tolerance 1e-5, relative reduction 1e7 * eps, 20 line-search steps, 15000 iterations), and its
// iterates are always inside the box.  Parity is statistical: the acquisition value at the returned
// point is compared with the reference's (tests/test_gpu_seams.py, scripts/r03_polish_modes.py), not the path.
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "gpbo_internal.h"

namespace gpbo {

namespace {

constexpr int LBFGS_M = 10;
constexpr double PGTOL = 1e-5;
constexpr double FTOL = 1e7 * 2.220446049250313e-16;
constexpr int MAXLS = 20;

inline double norm_cdf(double z) { return 0.5 * std::erfc(-z * 0.70710678118654752440); }
inline double norm_pdf(double z) { return std::exp(-0.5 * z * z) * 0.39894228040143267794; }

struct Run {
  int d = 0;
  std::vector<double> x, g, xt, dir, S, Y;       // S, Y: LBFGS_M rows of d
  double f = 0.0, alpha = 1.0;
  int hist = 0, head = 0;       // pairs stored, next slot
  int iter = 0, evals = 0, ls = 0;
  int phase = 0;                // 0: first evaluation pending, 1: line search, 2: finished
  int status = 2;               // 0: projected gradient, 1: relative reduction / no further progress, 2: iteration limit,
                                // 3: line search exhausted (SciPy: ABNORMAL_TERMINATION_IN_LNSRCH, success = False)
};

// max_i |P(x - g)_i - x_i|
double projected_gradient_norm(const Run& r, const double* lo, const double* hi) {
  double m = 0.0;
  for (int i = 0; i < r.d; ++i) {
    const double t = std::min(std::max(r.x[i] - r.g[i], lo[i]), hi[i]) - r.x[i];
    m = std::max(m, std::fabs(t));
  }
  return m;
}

// dir = -H g over the free variables (a variable sitting on a bound with the gradient pushing outwards stays there).
// The correction pairs are restricted to the CURRENT free set before they are used (components of fixed variables are
// dropped from s and y, a pair whose restricted curvature s.y is not positive is skipped): without that the pairs of
// an earlier active set steer the step and the run needs 1.5-2x the iterations (measured against SciPy on C2 / C3).
void new_direction(Run& r, const double* lo, const double* hi) {
  const int d = r.d;
  std::vector<char> freev((size_t)d);
  for (int i = 0; i < d; ++i)
    freev[i] = !((r.x[i] <= lo[i] && r.g[i] > 0.0) || (r.x[i] >= hi[i] && r.g[i] < 0.0));
  std::vector<double> q((size_t)d);
  for (int i = 0; i < d; ++i) q[i] = freev[i] ? r.g[i] : 0.0;
  double a[LBFGS_M], rho[LBFGS_M];
  int order[LBFGS_M], used = 0;        // newest first
  double gamma = 1.0;
  for (int t = 0; t < r.hist; ++t) {
    const int k = (r.head - 1 - t + 2 * LBFGS_M) % LBFGS_M;
    const double* s = &r.S[(size_t)k * d];
    const double* y = &r.Y[(size_t)k * d];
    double sy = 0.0, yy = 0.0;
    for (int i = 0; i < d; ++i)
      if (freev[i]) { sy += s[i] * y[i]; yy += y[i] * y[i]; }
    if (!(sy > 2.2e-16 * yy) || !(yy > 0.0)) continue;
    if (used == 0) gamma = sy / yy;
    rho[used] = 1.0 / sy;
    order[used++] = k;
  }
  for (int t = 0; t < used; ++t) {
    const double* s = &r.S[(size_t)order[t] * d];
    const double* y = &r.Y[(size_t)order[t] * d];
    double sq = 0.0;
    for (int i = 0; i < d; ++i) if (freev[i]) sq += s[i] * q[i];
    a[t] = rho[t] * sq;
    for (int i = 0; i < d; ++i) if (freev[i]) q[i] -= a[t] * y[i];
  }
  for (int i = 0; i < d; ++i) q[i] *= gamma;
  for (int t = used - 1; t >= 0; --t) {
    const double* s = &r.S[(size_t)order[t] * d];
    const double* y = &r.Y[(size_t)order[t] * d];
    double yq = 0.0;
    for (int i = 0; i < d; ++i) if (freev[i]) yq += y[i] * q[i];
    const double b = rho[t] * yq;
    for (int i = 0; i < d; ++i) if (freev[i]) q[i] += (a[t] - b) * s[i];
  }
  double gd = 0.0, gn = 0.0;
  for (int i = 0; i < d; ++i) {
    r.dir[i] = freev[i] ? -q[i] : 0.0;
    gd += r.dir[i] * r.g[i];
    if (freev[i]) gn += r.g[i] * r.g[i];
  }
  if (!(gd < 0.0) || !std::isfinite(gd)) {     // not a descent direction: steepest descent over the free variables, history dropped
    r.hist = 0;
    used = 0;
    for (int i = 0; i < d; ++i) r.dir[i] = freev[i] ? -r.g[i] : 0.0;
  }
  // L-BFGS-B takes a unit step except when it has no curvature information, where it starts from 1 / |d|
  r.alpha = (used == 0) ? std::min(1.0, 1.0 / std::sqrt(std::max(gn, 1e-300))) : 1.0;
  r.ls = 0;
}

void trial_point(Run& r, const double* lo, const double* hi) {
  for (int i = 0; i < r.d; ++i) r.xt[i] = std::min(std::max(r.x[i] + r.alpha * r.dir[i], lo[i]), hi[i]);
}

// one answer (f_t, g_t at r.xt) of the objective; leaves the next request in r.xt unless the run has finished
void advance(Run& r, double ft, const double* gt, const double* lo, const double* hi, int max_iter) {
  const int d = r.d;
  ++r.evals;
  if (r.phase == 0) {
    r.x = r.xt;
    r.f = ft;
    std::copy(gt, gt + d, r.g.begin());
    if (!std::isfinite(ft)) { r.phase = 2; r.status = 2; return; }
    if (projected_gradient_norm(r, lo, hi) <= PGTOL) { r.phase = 2; r.status = 0; return; }
    new_direction(r, lo, hi);
    trial_point(r, lo, hi);
    r.phase = 1;
    return;
  }
  double gs = 0.0, moved = 0.0;       // g . (x_t - x): the Armijo test on the displacement the projection left
  for (int i = 0; i < d; ++i) {
    const double s = r.xt[i] - r.x[i];
    gs += r.g[i] * s;
    moved = std::max(moved, std::fabs(s));
  }
  const bool ok = std::isfinite(ft) && ft <= r.f + 1e-4 * gs;
  if (!ok) {
    // no further progress along this path (x stays): the step no longer moves x, the line search is exhausted, or — after
    // two shrinks — the values differ by less than the relative-reduction tolerance, i.e. the test is deciding on rounding
    const bool flat = std::isfinite(ft) && r.ls >= 2 &&
                      std::fabs(ft - r.f) <= FTOL * std::max(std::max(std::fabs(ft), std::fabs(r.f)), 1.0);
    // (an exhausted line search is SciPy's "ABNORMAL" termination, success = False: the reference discards such a run,
    //  acquisition.py:367 — its own status, so that the caller can do the same)
    if (moved == 0.0 || flat) { r.phase = 2; r.status = 1; return; }
    if (++r.ls >= MAXLS) { r.phase = 2; r.status = 3; return; }
    // the minimiser of the parabola through f, its slope and f_t, kept inside [0.1, 0.5] of the step that failed
    double shrink = 0.1;
    if (std::isfinite(ft)) {
      const double curv = ft - r.f - gs;
      shrink = curv > 0.0 ? std::min(std::max(-gs / (2.0 * curv), 0.1), 0.5) : 0.5;
    }
    r.alpha *= shrink;
    trial_point(r, lo, hi);
    return;
  }
  // accepted
  {
    double* s = &r.S[(size_t)r.head * d];
    double* y = &r.Y[(size_t)r.head * d];
    double sy = 0.0, yy = 0.0;
    for (int i = 0; i < d; ++i) {
      s[i] = r.xt[i] - r.x[i];
      y[i] = gt[i] - r.g[i];
      sy += s[i] * y[i];
      yy += y[i] * y[i];
    }
    if (sy > 2.2e-16 * yy && yy > 0.0) {          // L-BFGS-B's curvature test (repeated on the free set when the pair is used)
      r.head = (r.head + 1) % LBFGS_M;
      r.hist = std::min(r.hist + 1, LBFGS_M);
    }
  }
  const double f_old = r.f;
  r.x = r.xt;
  r.f = ft;
  std::copy(gt, gt + d, r.g.begin());
  ++r.iter;
  if (projected_gradient_norm(r, lo, hi) <= PGTOL) { r.phase = 2; r.status = 0; return; }
  if ((f_old - ft) <= FTOL * std::max(std::max(std::fabs(f_old), std::fabs(ft)), 1.0)) { r.phase = 2; r.status = 1; return; }
  if (r.iter >= max_iter) { r.phase = 2; r.status = 2; return; }
  new_direction(r, lo, hi);
  trial_point(r, lo, hi);
}

// All runs in lockstep: `eval(batch, live, f, g)` evaluates the objective and its gradient at the `live` trial points of the
// round (rows of `batch`, d columns) in one go and returns a status code; the runs that are still alive get their answers
// and leave their next request.  Shared by gpbo_polish_seeds (device evaluator) and gpbo_debug_minimize_box (host callback:
// the optimiser alone, testable without a GPU).
template <class Eval>
int lockstep_minimize(Eval&& eval, const double* seeds, int n_seeds, int d, const double* box_lo, const double* box_hi, int max_iter,
                      double* x_out, double* f_out, int* status_out, int* n_rounds_out, int* n_iter_out, int* n_eval_out) {
  std::vector<Run> runs((size_t)n_seeds);
  for (int s = 0; s < n_seeds; ++s) {
    Run& r = runs[s];
    r.d = d;
    r.x.assign((size_t)d, 0.0); r.g.assign((size_t)d, 0.0); r.xt.assign((size_t)d, 0.0); r.dir.assign((size_t)d, 0.0);
    r.S.assign((size_t)LBFGS_M * d, 0.0); r.Y.assign((size_t)LBFGS_M * d, 0.0);
    for (int i = 0; i < d; ++i) r.xt[i] = std::min(std::max(seeds[(size_t)s * d + i], box_lo[i]), box_hi[i]);
  }
  std::vector<double> batch((size_t)n_seeds * d), fv((size_t)n_seeds), gv((size_t)n_seeds * d);
  std::vector<int> who((size_t)n_seeds);
  int rounds = 0;
  for (;;) {
    int live = 0;
    for (int s = 0; s < n_seeds; ++s)
      if (runs[s].phase != 2) {
        std::copy(runs[s].xt.begin(), runs[s].xt.end(), batch.begin() + (size_t)live * d);
        who[live++] = s;
      }
    if (live == 0) break;
    ++rounds;
    const int rc = eval(batch.data(), live, fv.data(), gv.data());
    if (rc) return rc;
    for (int t = 0; t < live; ++t) {
      double* g = &gv[(size_t)t * d];
      for (int i = 0; i < d; ++i) if (!std::isfinite(g[i])) g[i] = 0.0;
      advance(runs[who[t]], fv[t], g, box_lo, box_hi, max_iter);
    }
    if (rounds > 4 * max_iter + 64) break;      // cannot happen (every run is bounded by max_iter * MAXLS); never spin
  }
  for (int s = 0; s < n_seeds; ++s) {
    std::copy(runs[s].x.begin(), runs[s].x.end(), x_out + (size_t)s * d);
    f_out[s] = runs[s].f;
    status_out[s] = runs[s].phase == 2 ? runs[s].status : 2;
    if (n_iter_out) n_iter_out[s] = runs[s].iter;
    if (n_eval_out) n_eval_out[s] = runs[s].evals;
  }
  if (n_rounds_out) *n_rounds_out = rounds;
  return GPBO_OK;
}

}  // namespace

}  // namespace gpbo

using namespace gpbo;

#ifdef GPBO_DEBUG
extern "C" int gpbo_debug_minimize_box(gpbo_fg_callback fg, void* user, const double* seeds, int n_seeds, int d, const double* box_lo,
                                       const double* box_hi, int max_iter, double* x_out, double* f_out, int* status_out,
                                       int* n_rounds_out, int* n_iter_out, int* n_eval_out) {
  if (!fg || !seeds || !box_lo || !box_hi || !x_out || !f_out || !status_out || n_seeds < 1 || n_seeds > GPBO_MAX_SEEDS || d < 1 ||
      d > GPBO_MAX_DIM)
    return GPBO_ERR_INVALID;
  for (int i = 0; i < d; ++i)
    if (!(box_lo[i] < box_hi[i])) return GPBO_ERR_INVALID;
  if (max_iter < 1) max_iter = 15000;
  return lockstep_minimize([&](const double* batch, int live, double* f, double* g) { return fg(batch, live, d, f, g, user); }, seeds,
                           n_seeds, d, box_lo, box_hi, max_iter, x_out, f_out, status_out, n_rounds_out, n_iter_out, n_eval_out);
}
#endif  // GPBO_DEBUG

extern "C" int gpbo_polish_seeds(gpbo_ctx* ctx, int acq, double acq_param, double y_max, int n_constraints, const double* lb,
                                 const double* ub, const double* y_mean, const double* y_std, const double* seeds, int n_seeds,
                                 int d, const double* box_lo, const double* box_hi, int max_iter, double* x_out, double* f_out,
                                 int* status_out, int* n_rounds_out, int* n_iter_out, int* n_eval_out) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (n_seeds < 1 || n_seeds > GPBO_MAX_SEEDS) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "polish_seeds: n_seeds out of range [1, 64]");
  if (!seeds || !box_lo || !box_hi || !x_out || !f_out || !status_out || !y_mean || !y_std)
    GPBO_FAIL(ctx, GPBO_ERR_INVALID, "polish_seeds: NULL argument");
  if (acq != GPBO_ACQ_UCB && acq != GPBO_ACQ_EI && acq != GPBO_ACQ_POI) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "polish_seeds: unknown acquisition");
  if (n_constraints < 0 || n_constraints >= GPBO_MAX_MODELS || (n_constraints > 0 && (!lb || !ub)))
    GPBO_FAIL(ctx, GPBO_ERR_INVALID, "polish_seeds: bad constraint arguments");
  if (max_iter < 1) max_iter = 15000;
  for (int j = 0; j <= n_constraints; ++j) {
    if (!ctx->models[j].fitted) GPBO_FAIL(ctx, GPBO_ERR_STATE, "polish_seeds: model slot has not been fitted");
    if (ctx->models[j].d != d) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "polish_seeds: d differs from the fitted model's");
    if (ctx->pending_info[j]) GPBO_FAIL(ctx, GPBO_ERR_STATE, "polish_seeds: a fit of this slot is still in flight (gpbo_fit_wait)");
  }
  for (int i = 0; i < d; ++i)
    if (!(box_lo[i] < box_hi[i])) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "polish_seeds: every bound needs lo < hi");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));

  const int n_models = 1 + n_constraints;
  // pinned block, device-visible: [the round's points (n_seeds, d)] then per model [dmu | dsd | mu | sd] of a round (pitch = the
  // round's live runs).  The scaling kernel reads the points there and the last kernel of a model's evaluation writes its results
  // there: a round is six launches and ONE stream synchronisation, no copy nodes (round 5; until then: H2D copy + event, D2H copy).
  const size_t per_model = (size_t)n_seeds * (2 + 2 * (size_t)d);
  const size_t pts = (size_t)n_seeds * (size_t)d;
  const size_t need = (pts + per_model * (size_t)n_models) * sizeof(double);
  if ((int64_t)need > ctx->cap_polish_pinned) {
    if (ctx->polish_pinned) GPBO_HIP(ctx, hipHostFree(ctx->polish_pinned));
    ctx->polish_pinned = nullptr;
    ctx->cap_polish_pinned = 0;
    GPBO_HIP(ctx, hipHostMalloc(&ctx->polish_pinned, need, hipHostMallocDefault));
    GPBO_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->polish_pinned_dev, ctx->polish_pinned, 0));
    ctx->cap_polish_pinned = (int64_t)need;
  }
  double* pts_h = (double*)ctx->polish_pinned;
  double* land = pts_h + pts;
  const double* pts_d = (const double*)ctx->polish_pinned_dev;
  double* land_d = (double*)ctx->polish_pinned_dev + pts;
  const bool timing0 = ctx->no_timing;

  auto eval = [&](const double* batch, const int live, double* fv, double* gv) -> int {
    // ---- one batched evaluation: posterior + input gradient of every model at the live runs' trial points
    int rc = GPBO_OK;
    memcpy(pts_h, batch, (size_t)live * d * sizeof(double));
    ctx->no_timing = true;      // (no event records between the launches of a round)
    for (int j = 0; j < n_models && rc == GPBO_OK; ++j) {
      Model& m = ctx->models[j];
      double *dmu_dev = nullptr, *dsd_dev = nullptr;
      rc = launch_posterior_grad(ctx, m, live, y_mean[j], y_std[j], &dmu_dev, &dsd_dev, nullptr, pts_d, land_d + per_model * (size_t)j);
    }
    ctx->no_timing = timing0;
    if (rc) return rc;
    GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // ---- f = -acq [* prod_j p_j] and its gradient (acquisition.py:198-217, 485, 660-661, 847-849; constraint.py:199-221)
    for (int t = 0; t < live; ++t) {
      const double* o0 = land;
      const double* dmu = o0 + (size_t)t * d;
      const double* dsd = o0 + (size_t)live * d + (size_t)t * d;
      const double mu = o0[2 * (size_t)live * d + t], sd = o0[2 * (size_t)live * d + live + t];
      double a, ca, cs;      // acq value; d acq = ca * dmu + cs * dsd
      if (acq == GPBO_ACQ_UCB) {
        a = mu + acq_param * sd; ca = 1.0; cs = acq_param;
      } else {
        const double aa = mu - y_max - acq_param;
        const double z = aa / sd;
        const double cdf = norm_cdf(z), pdf = norm_pdf(z);
        if (acq == GPBO_ACQ_EI) { a = aa * cdf + sd * pdf; ca = cdf; cs = pdf; }
        else { a = cdf; ca = pdf / sd; cs = -pdf * z / sd; }
      }
      double f = -a;
      double* g = gv + (size_t)t * d;
      for (int i = 0; i < d; ++i) g[i] = -(ca * dmu[i] + cs * dsd[i]);
      if (n_constraints > 0) {
        double p = 1.0;
        std::vector<double> pj((size_t)n_constraints), dp((size_t)n_constraints * d, 0.0);
        for (int j = 0; j < n_constraints; ++j) {
          const double* o = land + per_model * (size_t)(1 + j);
          const double* dcm = o + (size_t)t * d;
          const double* dcs = o + (size_t)live * d + (size_t)t * d;
          const double cm = o[2 * (size_t)live * d + t], csd = o[2 * (size_t)live * d + live + t];
          double pv = 0.0;
          for (int side = 0; side < 2; ++side) {
            const double bound = side == 0 ? ub[j] : lb[j];
            const double sign = side == 0 ? 1.0 : -1.0;
            if (std::isinf(bound)) { if (bound > 0) pv += sign; continue; }   // Phi(+inf) = 1, Phi(-inf) = 0, for either side (constraint.py:202-207)
            const double z = (bound - cm) / csd;
            pv += sign * norm_cdf(z);
            const double w = sign * norm_pdf(z) / csd;
            for (int i = 0; i < d; ++i) dp[(size_t)j * d + i] += w * (-dcm[i] - z * dcs[i]);
          }
          pj[j] = pv;
          p *= pv;
        }
        for (int i = 0; i < d; ++i) {
          double sum = 0.0;
          for (int j = 0; j < n_constraints; ++j) {
            double others = 1.0;
            for (int k = 0; k < n_constraints; ++k) if (k != j) others *= pj[k];
            sum += dp[(size_t)j * d + i] * others;
          }
          g[i] = g[i] * p + f * sum;
        }
        f *= p;
      }
      fv[t] = f;
    }
    return GPBO_OK;
  };
  return lockstep_minimize(eval, seeds, n_seeds, d, box_lo, box_hi, max_iter, x_out, f_out, status_out, n_rounds_out, n_iter_out,
                           n_eval_out);
}
