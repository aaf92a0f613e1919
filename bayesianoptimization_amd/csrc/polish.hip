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
// Rounds 5-6: for one model of NP <= 384 the whole stage is ONE LAUNCH instead (polish_fused.hip: a workgroup per run with a
// thread per training point, evaluations and optimiser inside it; results agree with the rounds here to rounding, not to the
// bit): 3.4-31 us per evaluation against 42-61 us per round, and no lockstep — a default suggest() spends 0.38 instead of 1.13 ms
// in its local searches at N <= 143.  The rounds below serve everything else (larger models, constraint slots) and are the one
// launch's checker in the tests.
//
// The optimiser (polish_opt.h) is a projected L-BFGS (two-loop recursion over the free variables, backtracking on the projected path
// with an Armijo test on the actual displacement), NOT a transcription of L-BFGS-B: no generalised Cauchy point, no
// subspace minimisation.  It keeps L-BFGS-B's stopping rule as SciPy configures it for `minimize` (m = 10 corrections,
// projected-gradient tolerance 1e-5, relative reduction 1e7 * eps, 20 line-search steps, 15000 iterations), and its
// iterates are always inside the box.  Parity is statistical: the acquisition value at the returned
// point is compared with the reference's (tests/test_gpu_seams.py, scripts/archive/r03_polish_modes.py), not the path.
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "gpbo_internal.h"
#include "polish_opt.h"

namespace gpbo {

namespace {

inline double norm_cdf(double z) { return 0.5 * std::erfc(-z * 0.70710678118654752440); }
inline double norm_pdf(double z) { return std::exp(-0.5 * z * z) * 0.39894228040143267794; }

// All runs in lockstep: `eval(batch, live, f, g)` evaluates the objective and its gradient at the `live` trial points of the
// round (rows of `batch`, d columns) in one go and returns a status code; the runs that are still alive get their answers
// and leave their next request.  Shared by gpbo_polish_seeds (device evaluator) and gpbo_debug_minimize_box (host callback:
// the optimiser alone, testable without a GPU).
template <class Eval>
int lockstep_minimize(Eval&& eval, const double* seeds, int n_seeds, int d, const double* box_lo, const double* box_hi, int max_iter,
                      double* x_out, double* f_out, int* status_out, int* n_rounds_out, int* n_iter_out, int* n_eval_out) {
  // (the optimiser itself: polish_opt.h; the one-launch device path restates it lane-parallel and is held to its bits)
  std::vector<PolishRun> runs((size_t)n_seeds);
  std::vector<double> store((size_t)n_seeds * polish_run_doubles(d));
  std::vector<int> istore((size_t)n_seeds * polish_run_ints(d));
  for (int s = 0; s < n_seeds; ++s) {
    polish_run_bind(runs[s], d, store.data() + (size_t)s * polish_run_doubles(d), istore.data() + (size_t)s * polish_run_ints(d));
    polish_start(runs[s], seeds + (size_t)s * d, box_lo, box_hi);
  }
  std::vector<double> batch((size_t)n_seeds * d), fv((size_t)n_seeds), gv((size_t)n_seeds * d);
  std::vector<int> who((size_t)n_seeds);
  int rounds = 0;
  for (;;) {
    int live = 0;
    for (int s = 0; s < n_seeds; ++s)
      if (runs[s].phase != 2) {
        std::copy(runs[s].xt, runs[s].xt + d, batch.begin() + (size_t)live * d);
        who[live++] = s;
      }
    if (live == 0) break;
    ++rounds;
    const int rc = eval(batch.data(), live, fv.data(), gv.data());
    if (rc) return rc;
    for (int t = 0; t < live; ++t) {
      double* g = &gv[(size_t)t * d];
      for (int i = 0; i < d; ++i) if (!std::isfinite(g[i])) g[i] = 0.0;
      polish_advance(runs[who[t]], fv[t], g, box_lo, box_hi, max_iter);
    }
    if (rounds > 4 * max_iter + 64) break;      // cannot happen (every run is bounded by max_iter * MAXLS); never spin
  }
  for (int s = 0; s < n_seeds; ++s) {
    std::copy(runs[s].x, runs[s].x + d, x_out + (size_t)s * d);
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
  if (max_iter > 100000000) max_iter = 100000000;      // (4 * max_iter + 64 rounds is an int)
  return lockstep_minimize([&](const double* batch, int live, double* f, double* g) { return fg(batch, live, d, f, g, user); }, seeds,
                           n_seeds, d, box_lo, box_hi, max_iter, x_out, f_out, status_out, n_rounds_out, n_iter_out, n_eval_out);
}
#endif  // GPBO_DEBUG

#ifdef GPBO_DEBUG
// The one-launch path's objective at each of n points, evaluated `repeat` times in one launch (timing): out (n, 4 + 3 d) =
// [f, mu, sd, 0 | g | dmu | dsd].
extern "C" int gpbo_debug_polish_eval(gpbo_ctx* ctx, int acq, double acq_param, double y_max, double y_mean, double y_std,
                                      const double* points, int n, int d, int repeat, double* out) {
  if (!ctx || !points || !out || n < 1 || n > GPBO_MAX_SEEDS || repeat < 1) return GPBO_ERR_INVALID;
  Model& m = ctx->models[0];
  if (!m.fitted || m.d != d) GPBO_FAIL(ctx, GPBO_ERR_STATE, "debug_polish_eval: slot 0 is not fitted for this d");
  if (!polish_fused_serves(m)) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "debug_polish_eval: the model is outside the one-launch path's range");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  const size_t need = polish_fused_pinned_bytes(n, d);
  if ((int64_t)need > ctx->cap_polish_pinned) {
    if (ctx->polish_pinned) GPBO_HIP(ctx, hipHostFree(ctx->polish_pinned));
    ctx->polish_pinned = nullptr;
    ctx->cap_polish_pinned = 0;
    GPBO_HIP(ctx, hipHostMalloc(&ctx->polish_pinned, need, hipHostMallocDefault));
    GPBO_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->polish_pinned_dev, ctx->polish_pinned, 0));
    ctx->cap_polish_pinned = (int64_t)need;
  }
  std::vector<double> lo((size_t)d, -1e300), hi((size_t)d, 1e300);
  const int rc = launch_polish_fused(ctx, m, acq, acq_param, y_max, y_mean, y_std, points, n, lo.data(), hi.data(), 1, repeat,
                                     (double*)ctx->polish_pinned, (double*)ctx->polish_pinned_dev);
  if (rc) return rc;
  const size_t S = (size_t)n;
  const double* dbg = (const double*)ctx->polish_pinned + 2 * S * d + 2 * (size_t)d + S;
  std::copy(dbg, dbg + S * (4 + 3 * (size_t)d), out);
  return GPBO_OK;
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
  if (max_iter > 100000000) max_iter = 100000000;      // (4 * max_iter + 64 rounds is an int)
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
  // (NP <= polish_fused_max_np(), one model: the runs as ONE launch — polish_fused.hip; GPBO_POLISH_FUSED=0, debug build: never)
  const char* pf_env = dbg_env("GPBO_POLISH_FUSED");
  const bool fused = n_constraints == 0 && polish_fused_serves(ctx->models[0]) && !(pf_env && pf_env[0] == '0');
  const size_t need = std::max((pts + per_model * (size_t)n_models) * sizeof(double), fused ? polish_fused_pinned_bytes(n_seeds, d) : (size_t)0);
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

  if (fused) {
    const int rc = launch_polish_fused(ctx, ctx->models[0], acq, acq_param, y_max, y_mean[0], y_std[0], seeds, n_seeds, box_lo, box_hi,
                                       max_iter, 0, (double*)ctx->polish_pinned, (double*)ctx->polish_pinned_dev);
    if (rc) return rc;
    const size_t S = (size_t)n_seeds;
    const double* xo = (const double*)ctx->polish_pinned + S * d + 2 * (size_t)d;
    const double* fo = xo + S * d;
    const int* io = (const int*)(fo + S + S * (4 + 3 * (size_t)d));
    std::copy(xo, xo + S * d, x_out);
    std::copy(fo, fo + S, f_out);
    int rounds = 0;
    for (int s = 0; s < n_seeds; ++s) {
      status_out[s] = io[s];
      if (n_iter_out) n_iter_out[s] = io[S + s];
      if (n_eval_out) n_eval_out[s] = io[2 * S + s];
      rounds = std::max(rounds, io[2 * S + s]);
    }
    if (n_rounds_out) *n_rounds_out = rounds;      // (what the lockstep path counts: batched evaluations = the longest run's)
    return GPBO_OK;
  }

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
      polish_acq_coeffs(acq, acq_param, y_max, mu, sd, norm_cdf, norm_pdf, a, ca, cs);
      double f = -a;
      double* g = gv + (size_t)t * d;
      for (int i = 0; i < d; ++i) g[i] = polish_acq_grad(ca, cs, dmu[i], dsd[i]);
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
