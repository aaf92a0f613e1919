"""Acquisition policies whose random-search stage runs fused on the GPU (seam B1).

Drop-in for `BayesianOptimization(acquisition_function=...)`.  The reference driver touches an acquisition object
only through `suggest(gp=, target_space=, fit_gp=True, random_state=)` (bayes_opt/bayesian_optimization.py:329-331),
`_fit_gp(gp, space)` (:236) and `get/set_acquisition_params` (:430, :471); subclasses and notebooks additionally
rely on the override points `base_acq`, `_get_acq`, `_acq_min`, `_random_sample_minimize`, `_smart_minimize`
(bayes_opt/acquisition.py:75-420).  This module keeps those names, argument meanings, error types/messages, the
`i` counter, kappa/xi decay and — crucially — the order in which the shared RandomState is consumed, but is written
independently of the reference and does not import it, so it also runs where bayes_opt is not installed.

Device path: when the target GP (and every constraint GP) is a `HipGPR` on one engine, everything after candidate
sampling in the random stage — posterior mean/std of M candidates, -acq(x) [* p_constraint], argmin, min,
argsort[:k] (acquisition.py:311-317) — is one sequence of HIP kernels; only the arg-best record and the k seeds
return to the host.  Any other `gp` object (duck-typed mocks, plain sklearn estimators) goes through the `_get_acq`
closure on the host, as in the reference.  The local-search stage stays a host L-BFGS-B / differential evolution,
with its finite-difference gradient evaluated as one device batch per iteration (`_fd_value_and_grad`).
"""
from __future__ import annotations

import abc
import contextlib
import threading
import warnings

import numpy as np
from scipy.optimize import minimize
from scipy.special import ndtr

from . import engine as E
from . import lbfgsb_lockstep
from .float_space import ensure_rng
from .lockstep import Lockstep as _Lockstep
from .gpr import HipGPR

try:  # raise the reference's own exception classes when it is installed, so `except` clauses keep working
    from bayes_opt.exception import (ConstraintNotSupportedError, NoValidPointRegisteredError,
                                     TargetSpaceEmptyError)
except Exception:  # pragma: no cover - exercised on boxes without bayes_opt
    class BayesianOptimizationError(Exception):
        """Base class (bayes_opt/exception.py:14)."""

    class ConstraintNotSupportedError(BayesianOptimizationError):
        """bayes_opt/exception.py:22."""

    class NoValidPointRegisteredError(BayesianOptimizationError):
        """bayes_opt/exception.py:26."""

    class TargetSpaceEmptyError(BayesianOptimizationError):
        """bayes_opt/exception.py:30."""

_SQRT_2PI = np.sqrt(2.0 * np.pi)
_FD_EPS = 1e-8                       # scipy.optimize._lbfgsb_py._minimize_lbfgsb: eps=1e-8 -> abs_step
_SQRT_EPS = np.finfo(np.float64).eps ** 0.5
_MAX_DEVICE_SEEDS = 64               # GPBO_MAX_SEEDS


def _norm_pdf(x):
    return np.exp(-(x**2) / 2.0) / _SQRT_2PI  # scipy/stats/_continuous_distns.py:360-362


def _fd_value_and_grad(acq, bounds):
    """value-and-gradient callable reproducing, in ONE batched acq() call per iteration, the forward
    differences SciPy's L-BFGS-B forms one point at a time when `jac` is not given
    (scipy/optimize/_numdiff.py: approx_derivative(method="2-point", abs_step=1e-8, bounds) ->
    _adjust_scheme_to_bounds(..., "1-sided") -> _dense_difference): same steps, same divisions, so the
    optimiser sees the same (f, g) and walks the same path as the reference's
    `minimize(acq, x_try, bounds=..., method="L-BFGS-B")` (bayes_opt/acquisition.py:365-366)."""
    lb, ub = bounds[:, 0].astype(float), bounds[:, 1].astype(float)
    unbounded = bool(np.all((lb == -np.inf) & (ub == np.inf)))

    def fun(x):
        x0 = np.asarray(x, dtype=np.float64)
        d = x0.shape[0]
        sign_x0 = (x0 >= 0).astype(float) * 2 - 1
        h = np.full(d, _FD_EPS)
        h = np.where((x0 + h) - x0 == 0, _SQRT_EPS * sign_x0 * np.maximum(1.0, np.abs(x0)), h)
        if not unbounded:
            room_below, room_above = x0 - lb, ub - x0
            trial = x0 + h
            outside = (trial < lb) | (trial > ub)
            fits = np.abs(h) <= np.maximum(room_below, room_above)
            h = h.copy()
            h[outside & fits] *= -1
            go_up = (room_above >= room_below) & ~fits
            h[go_up] = room_above[go_up]
            go_down = (room_above < room_below) & ~fits
            h[go_down] = -room_below[go_down]
        pts = np.empty((d + 1, d))
        pts[0] = x0
        pts[1:] = x0 + np.diag(h)
        vals = np.asarray(acq(pts), dtype=np.float64)
        steps = pts[1:][np.arange(d), np.arange(d)] - x0
        return vals[0], (vals[1:] - vals[0]) / steps

    return fun


def _polish_in_lockstep(acq, x_seeds, box):
    """[OptimizeResult per seed] of `minimize(fd(acq), seed, jac=True, bounds=box, method="L-BFGS-B")`, the
    runs advanced together through `_Lockstep`."""
    n = len(x_seeds)
    hub = _Lockstep(acq, n)
    results: list = [None] * n
    errors: list = [None] * n

    def run(idx, start):
        try:
            fun = _fd_value_and_grad(lambda pts: hub.ask(idx, pts), box)
            results[idx] = minimize(fun, start, jac=True, bounds=box, method="L-BFGS-B")
        except _Lockstep.Abort:
            pass
        except BaseException as exc:
            errors[idx] = exc
        finally:
            hub.retire(idx)

    threads = [threading.Thread(target=run, args=(i, np.array(x, dtype=np.float64)), daemon=True)
               for i, x in enumerate(x_seeds)]
    for t in threads:
        t.start()
    try:
        hub.serve()
    finally:
        for t in threads:
            t.join()
    for exc in errors:
        if exc is not None:
            raise exc
    return results


def _reference_stream_on_device(chain, space, random_state, n_random) -> bool:
    """May the device reproduce `space.random_sample(n_random, random_state)`?  Yes when that call is one
    `random_state.uniform(lo_j, hi_j, n_random)` per column in `space.bounds` order (target_space.py:593-600 with only
    FloatParameters, parameter.py:86-87) on an MT19937 RandomState, the GPs take the raw coordinates, and the batch
    is big enough to be worth a launch."""
    if chain[0].transform is not None or n_random * space.bounds.shape[0] < 4096:
        return False
    if not isinstance(random_state, np.random.RandomState) or random_state.get_state()[0] != "MT19937":
        return False
    config = getattr(space, "_params_config", None)
    if config is not None and not all(type(p).__name__ == "FloatParameter" for p in config.values()):
        return False
    sampler = getattr(type(space), "random_sample", None)
    return getattr(sampler, "__module__", None) in ("bayes_opt.target_space", "bayesianoptimization_amd.float_space")


_PARAM_KINDS = {"FloatParameter": 0, "IntParameter": 1, "CategoricalParameter": 2}


def _mixed_groups_on_device(chain, space, random_state, n_random):
    """Column groups [(kind, col0, ncols, lo, hi, param)] if the device may assemble `space.random_sample(n_random,
    random_state)` AND apply `space.kernel_transform` itself (second half), else None.  Yes when: the space
    is the reference's TargetSpace (or this package's stand-in) holding only the reference's three parameter kinds with at
    least one non-float among them, sampled by the stock per-parameter loop (target_space.py:593-600) on an MT19937
    RandomState; every GP's input transform IS this space's kernel_transform (what accelerate() installs); the engine is a
    single device; and the batch is worth a launch."""
    eng = chain[0]._engine()
    if not getattr(eng, "mixed_device_sampling", False) or not hasattr(eng, "generate_candidates_mixed"):
        return None
    config = getattr(space, "_params_config", None)
    masks = getattr(space, "masks", None)
    if config is None or masks is None or n_random * space.bounds.shape[0] < 4096:
        return None
    if not isinstance(random_state, np.random.RandomState) or random_state.get_state()[0] != "MT19937":
        return None
    if getattr(getattr(type(space), "random_sample", None), "__module__", None) not in ("bayes_opt.target_space",
                                                                                         "bayesianoptimization_amd.float_space"):
        return None
    for model in chain:
        t = model.transform
        if t is None or getattr(t, "__self__", None) is not space or getattr(t, "__name__", "") != "kernel_transform":
            return None
    groups, col = [], 0
    for key in getattr(space, "_keys", list(config)):
        p = config[key]
        kind = _PARAM_KINDS.get(type(p).__name__)
        if kind is None or type(p).__module__ not in ("bayes_opt.parameter", "bayesianoptimization_amd.float_space"):
            return None
        dim = int(p.dim)
        if not np.array_equal(np.flatnonzero(masks[key]), np.arange(col, col + dim)):
            return None
        b = np.atleast_2d(np.asarray(p.bounds, dtype=np.float64))
        groups.append((kind, col, dim, b[:, 0].copy(), b[:, 1].copy(), p))
        col += dim
    if col != space.bounds.shape[0] or all(g[0] == 0 for g in groups) or col > 64:
        return None
    return groups


def _fused_models(gp, constraint):
    """[target, constraint GPs...] if all are HipGPRs on one engine in slots 0..n — and on the device path (a model whose
    kernel the engine does not evaluate runs scikit-learn's code, HipGPR._fit_on_host) — else None."""
    if not isinstance(gp, HipGPR) or gp.slot != 0 or gp._host_mode:
        return None
    chain = [gp]
    members = [] if constraint is None else getattr(constraint, "_model", None)
    if members is None or len(members) + 1 > 8:
        return None
    for slot, model in enumerate(members, start=1):
        if not isinstance(model, HipGPR) or model.slot != slot or model._host_mode or model._engine() is not gp._engine():
            return None
        chain.append(model)
    return chain


class AcquisitionFunction(abc.ABC):
    """Interface of bayes_opt.acquisition.AcquisitionFunction (bayes_opt/acquisition.py:56-420)."""

    #: default number of random candidates; the reference hard-codes 10_000 (acquisition.py:120)
    default_n_random = 10_000
    #: with engine-backed GPs, form L-BFGS-B's finite-difference gradient in one batched device call per
    #: iteration (d + 1 points) instead of d + 1 single-point calls; same numbers, ~d times fewer launches
    batched_fd = True
    #: with `batched_fd`, advance the L-BFGS-B runs of all seeds together so that each iteration of ALL runs is
    #: one device batch (n_seeds * (d + 1) points) instead of one batch per run.  True: SciPy's reverse-communication
    #: routine driven directly on one thread (lbfgsb_lockstep.py) when the installed SciPy is the one it was written
    #: against, else one thread per run around the public `minimize` (`_Lockstep`); "threads" forces the latter
    lockstep = True
    #: where the random-stage candidates are drawn when the GPs live on the engine:
    #:   "auto"    (default) on the device in INDEX-PARITY mode — gpbo_generate_candidates_mt19937 walks the caller's
    #:             MT19937 RandomState and returns it advanced, so candidates, suggestion and every later draw are
    #:             bit for bit the reference's — whenever that applies (all-float space without input transform,
    #:             legacy RandomState, >= 4096 values); otherwise on the host
    #:   False     always space.random_sample() on the host + upload
    #:   True      THROUGHPUT MODE: a Philox generator on the device; NOT the reference's stream (two 31-bit integers
    #:             are drawn from it as the seed), so suggestions differ
    device_sampling = "auto"
    #: local-search gradient.  False (default): the reference's finite differences (d + 1 predicts
    #: per evaluation, batched) — L-BFGS-B sees the same (f, g) as under bayes_opt and walks the same iterates.
    #: True: the exact gradient from ONE point per evaluation (gpbo_predict_grad: u = W^T W k*, chain rule through the
    #: acquisition and the constraint probabilities) — d + 1 times fewer posterior points per evaluation; the iterates
    #: are no longer SciPy's finite-difference ones, so parity with the reference is statistical (equal or better
    #: acquisition value), not bit-wise.  Needs engine-backed GPs without input transform and a stock policy.
    analytic_gradient = False
    #: the local-search stage.  "auto" (default since round 4) / True: ONE library call whenever it
    #: applies — engine-backed GPs without input transform, a stock policy, a non-degenerate box, <= 64 seeds
    #: (gpbo_polish_seeds: projected L-BFGS on the host side of the C ABI, batched value + analytic gradient on the device,
    #: SciPy's stopping rule; no SciPy, no Python between the rounds) — otherwise the reference-shaped path below.  Not the
    #: reference's iterates: parity is statistical, the acquisition value at the returned point (the 66-problem x 10-seed
    #: sweep of tests/test_gpu_polish.py, profiles/r04_polish_sweep.json, is what the default rests on).
    #: False: SciPy's L-BFGS-B over (batched) finite differences, iterate for iterate the reference's local searches
    #: (bayes_opt/acquisition.py:364-374).
    device_polish = "auto"
    _acq_kind: int | None = None     # engine acquisition id of the stock policies; None = host formula only

    def __init__(self, random_state=None) -> None:
        if random_state is not None:
            warnings.warn(
                "Providing a random_state to an acquisition function during initialization is deprecated "
                "and will be ignored. The random_state is instead provided automatically during the "
                "suggest() call.", DeprecationWarning, stacklevel=2)
        self.i = 0
        self._fused = None

    # ---- interface ----------------------------------------------------------------------------------
    @abc.abstractmethod
    def base_acq(self, *args, **kwargs):
        """Acquisition value from the posterior mean and standard deviation."""

    def _acq_param(self) -> float:
        raise NotImplementedError

    def get_acquisition_params(self):
        raise NotImplementedError(
            "Custom AcquisitionFunction subclasses must implement their own get_acquisition_params method.")

    def set_acquisition_params(self, params):
        raise NotImplementedError(
            "Custom AcquisitionFunction subclasses must implement their own set_acquisition_params method.")

    def _fit_gp(self, gp, target_space) -> None:
        """Fit the target GP, then the constraint GPs, warnings silenced (acquisition.py:79-86).  With a constraint the
        device factorisations of the 1 + n_constraints GPs are enqueued side by side (GpEngine.overlapped_fits: each is a
        latency-bound chain, so they overlap) and waited for when the block ends — a non-positive-definite kernel
        matrix raises np.linalg.LinAlgError here, still inside suggest(), as in the reference."""
        cons = target_space.constraint
        eng = None
        if cons is not None and hasattr(gp, "_engine"):
            try:
                eng = gp._engine()
            except Exception:  # noqa: BLE001
                eng = None
        overlap = eng.overlapped_fits() if hasattr(eng, "overlapped_fits") else contextlib.nullcontext()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with overlap:
                gp.fit(target_space.params, target_space.target)
                if cons is not None:
                    cons.fit(target_space.params, target_space._constraint_values)

    # ---- suggest ------------------------------------------------------------------------------------
    def suggest(self, gp, target_space, n_random: int | None = None, n_smart: int = 10, fit_gp: bool = True,
                random_state=None):
        """Next point to probe (acquisition.py:116-169); `n_random=None` -> `default_n_random` (10_000 there)."""
        rng = ensure_rng(random_state)
        if len(target_space) == 0:
            raise TargetSpaceEmptyError(
                "Cannot suggest a point without previous samples. Use "
                " target_space.random_sample() to generate a point and "
                " target_space.probe(*) to evaluate it.")
        self.i += 1
        if fit_gp:
            self._fit_gp(gp=gp, target_space=target_space)
        objective = self._get_acq(gp=gp, constraint=target_space.constraint)
        self._fused = None if self._acq_kind is None else _fused_models(gp, target_space.constraint)
        self._fused_constraint = target_space.constraint if self._fused is not None and len(self._fused) > 1 else None
        try:
            return self._acq_min(objective, target_space, n_smart=n_smart, random_state=rng,
                                 n_random=self.default_n_random if n_random is None else n_random)
        finally:
            self._fused = None
            self._fused_constraint = None

    def _get_acq(self, gp, constraint=None):
        """Host objective x -> -acq(x) [* p_constraint(x)] for (M,d) or (d,) input (acquisition.py:171-219)."""
        ndim = gp.X_train_.shape[1]
        # engine-backed models: finite float batches skip sklearn's per-call input validation and the warnings
        # bookkeeping (the values are the same; anything else takes the reference-shaped path below and fails there)
        trusted = isinstance(gp, HipGPR) and (constraint is None or hasattr(constraint, "_predict_trusted"))

        def objective(x):
            batch = x.reshape(-1, ndim)
            if trusted and batch.dtype == np.float64 and np.isfinite(batch).all():
                mean, std = gp._posterior_trusted(batch)
                with np.errstate(all="ignore"):
                    values = -1 * self.base_acq(mean, std)
                    return values if constraint is None else values * constraint._predict_trusted(batch)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                mean, std = gp.predict(batch, return_std=True)
                if constraint is None:
                    return -1 * self.base_acq(mean, std)
                feasible = constraint.predict(batch)
            return -1 * self.base_acq(mean, std) * feasible

        return objective

    def base_acq_grad(self, mean, std, dmean, dstd):
        """(acq, d acq / d x) from the posterior and its input gradient; stock policies override."""
        raise NotImplementedError

    def _value_and_grad(self, models, constraint):
        """Batch objective X (S, d) -> (-acq(x) [* p_c(x)], its gradient (S, d)) on the engine, one point per
        evaluation (the analytic counterpart of `_get_acq`, acquisition.py:171-219)."""
        target = models[0]

        def fun(X):
            X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, target.X_train_.shape[1])
            mean, std, dmean, dstd = target._posterior_grad_trusted(X)
            with np.errstate(all="ignore"):
                a, da = self.base_acq_grad(mean, std, dmean, dstd)
                f, g = -a, -da
                if constraint is not None:
                    p = np.ones_like(mean)
                    dlogs = []                      # (p_j, d p_j / d x) per constraint
                    for gp, lo, hi in zip(models[1:], constraint._lb, constraint._ub):
                        cm, cs, dcm, dcs = gp._posterior_grad_trusted(X)
                        pj = np.zeros_like(cm)
                        dpj = np.zeros_like(dcm)
                        for bound, sign in ((hi, 1.0), (lo, -1.0)):
                            if not np.isfinite(bound):       # cdf(+inf) = 1, cdf(-inf) = 0, for either side (constraint.py:202-207)
                                pj = pj + (sign if bound == np.inf else 0.0)
                                continue
                            z = (bound - cm) / cs
                            pj = pj + sign * ndtr(z)
                            dpj = dpj + sign * (_norm_pdf(z) / cs)[:, None] * (-dcm - z[:, None] * dcs)
                        dlogs.append((pj, dpj))
                        p = p * pj
                    gp_sum = np.zeros_like(g)
                    for j, (pj, dpj) in enumerate(dlogs):
                        others = np.ones_like(pj)
                        for i, (pi, _) in enumerate(dlogs):
                            if i != j:
                                others = others * pi
                        gp_sum = gp_sum + dpj * others[:, None]
                    g = g * p[:, None] + f[:, None] * gp_sum
                    f = f * p
            return f, np.where(np.isfinite(g), g, 0.0)

        return fun

    def _acq_min(self, acq, space, random_state, n_random: int = 10_000, n_smart: int = 10):
        """Random stage, then local searches from its best points; the better of the two (acquisition.py:221-272)."""
        if n_random == 0 and n_smart == 0:
            raise ValueError("Either n_random or n_smart needs to be greater than 0.")
        x_rand, f_rand, seeds = self._random_sample_minimize(acq, space, random_state,
                                                             n_random=max(n_random, n_smart), n_x_seeds=n_smart)
        if not n_smart:
            return x_rand
        x_loc, f_loc = self._smart_minimize(acq, space, x_seeds=seeds, random_state=random_state)
        return x_loc if f_rand > f_loc else x_rand

    def _random_sample_minimize(self, acq, space, random_state, n_random: int, n_x_seeds: int = 0):
        """(x_min, min_acq, x_seeds) over `n_random` uniform candidates (acquisition.py:274-320)."""
        if n_random == 0:
            return None, np.inf, space.random_sample(n_x_seeds, random_state=random_state)
        chain = getattr(self, "_fused", None) if n_x_seeds <= _MAX_DEVICE_SEEDS else None
        if chain is not None and self.device_sampling is True and chain[0].transform is None:
            seed = int(random_state.randint(0, 2**31 - 1)) | (int(random_state.randint(0, 2**31 - 1)) << 31)
            return self._device_minimize(chain, space, None, n_x_seeds, n_random=n_random, seed=seed)
        if chain is not None and self.device_sampling == "auto" and _reference_stream_on_device(chain, space,
                                                                                               random_state, n_random):
            return self._device_minimize(chain, space, None, n_x_seeds, n_random=n_random, stream=random_state)
        if chain is not None and self.device_sampling == "auto":
            groups = _mixed_groups_on_device(chain, space, random_state, n_random)
            if groups is not None:      # float + int / categorical parameters: assembled and transformed on the device
                return self._device_minimize(chain, space, None, n_x_seeds, n_random=n_random, stream=random_state, groups=groups)
        x_tries = space.random_sample(n_random, random_state=random_state)
        if chain is not None:
            return self._device_minimize(chain, space, x_tries, n_x_seeds)
        values = acq(x_tries)
        seeds = x_tries[np.argsort(values)[:n_x_seeds]] if n_x_seeds != 0 else []
        return x_tries[values.argmin()], values.min(), seeds

    def _device_minimize(self, models, space, x_tries, n_x_seeds, n_random=None, seed=None, stream=None, groups=None):
        """The body of the random stage after sampling, on the GPU (kernels K5-K8 of SURVEY.md §2.1).
        x_tries=None: the candidates are generated on the device too — from `stream`'s MT19937 state (the
        reference's candidates) or, with `seed`, by the Philox throughput generator."""
        target = models[0]
        eng = target._engine()
        if x_tries is None and stream is not None and groups is not None:
            # a mixed space: float columns from the device generator, int / categorical columns drawn on the host at the
            # right position of the same stream, then space.kernel_transform on the device (the GPs' own host transform
            # is skipped: nothing M-sized crosses PCIe but the 8 B per candidate of each non-float column)
            eng.generate_candidates_mixed(n_random, groups, stream)
            eng.transform_candidates(groups)
        elif x_tries is None and stream is not None:
            eng.generate_candidates_like(n_random, space.bounds[:, 0], space.bounds[:, 1], stream)
        elif x_tries is None:
            eng.generate_candidates(n_random, space.bounds[:, 0], space.bounds[:, 1], seed)
        else:
            eng.set_candidates(target._tx(x_tries))
        for model in models:
            model.posterior_resident()
        lb, ub = (space.constraint._lb, space.constraint._ub) if len(models) > 1 else (None, None)
        y_max = getattr(self, "y_max", None)
        best, best_val, picks, _vals, _ = eng.acq_argbest(self._acq_kind, self._acq_param(),
                                                           0.0 if y_max is None else y_max, lb, ub,
                                                           k_seeds=n_x_seeds)
        picks = picks[picks >= 0]
        if x_tries is None:
            rows = eng.get_candidate_rows(np.concatenate([[best], picks]), space.bounds.shape[0])
            return rows[0], best_val, (rows[1:] if n_x_seeds else [])
        return x_tries[best], best_val, (x_tries[picks] if n_x_seeds else [])

    # ---- local search -------------------------------------------------------------------------------
    def _smart_minimize(self, acq, space, x_seeds, random_state):
        """Local refinement of the seeds (acquisition.py:322-420): L-BFGS-B per seed when every parameter is
        continuous, otherwise differential evolution (+ an L-BFGS-B polish of the continuous coordinates);
        returns (x clipped to the bounds, value), or (NaNs, inf) if nothing converged."""
        is_cont = space.continuous_dimensions
        box = space.bounds[is_cont]
        if all(is_cont):
            winner = self._polish_seeds(acq, x_seeds, box)
        else:
            winner = self._evolve_mixed(acq, space, x_seeds, random_state, is_cont, box)
        if winner is None:
            return np.clip(np.full(space.bounds.shape[0], np.nan), space.bounds[:, 0], space.bounds[:, 1]), np.inf
        x_best, f_best = winner
        return np.clip(x_best, space.bounds[:, 0], space.bounds[:, 1]), f_best

    @staticmethod
    def _objective_on_device(chain) -> bool:
        """True when every evaluation of the local searches' objective is a device call (a real engine behind the fused
        models): L-BFGS-B's own tiny BLAS calls may then stay off the host's thread pool (lbfgsb_lockstep.blas_single_thread)."""
        try:
            return chain is not None and type(chain[0]._engine()).__module__.startswith("bayesianoptimization_amd.")
        except Exception:  # noqa: BLE001
            return False

    def _polish_seeds(self, acq, x_seeds, box):
        chain = getattr(self, "_fused", None)
        if (self.device_polish and chain is not None and len(x_seeds) > 0 and chain[0].transform is None
                and not np.any(box[:, 0] == box[:, 1]) and hasattr(chain[0]._engine(), "polish_seeds")
                and len(x_seeds) <= _MAX_DEVICE_SEEDS):
            cons = getattr(self, "_fused_constraint", None)
            for model in chain:
                model._ensure_resident()
            lb, ub = (cons._lb, cons._ub) if cons is not None and len(chain) > 1 else (None, None)
            xs, fs, status, _ = chain[0]._engine().polish_seeds(
                self._acq_kind, self._acq_param(), getattr(self, "y_max", None), lb, ub,
                [float(m._y_train_mean) for m in chain], [float(m._y_train_std) for m in chain], np.asarray(x_seeds), box)
            ok = np.flatnonzero((status < 2) & np.isfinite(fs))         # SciPy's res.success (2: iteration limit, 3: line search exhausted)
            if ok.size == 0:
                return None
            k = ok[np.argmin(fs[ok])]                                  # the first of equal minima, as the reference's scan
            return xs[k], fs[k]
        if (self.analytic_gradient and chain is not None and len(x_seeds) > 0 and chain[0].transform is None
                and not np.any(box[:, 0] == box[:, 1])):
            cons = getattr(self, "_fused_constraint", None)
            fg = self._value_and_grad(chain, cons)
            winner = None
            if lbfgsb_lockstep.driver_available():
                outcomes = lbfgsb_lockstep.minimize_many_with_grad(fg, x_seeds, box, single_thread_blas=self._objective_on_device(chain))       # one device call per round
            else:
                outcomes = (minimize(lambda x: tuple(v[0] for v in fg(x[None])), s0, jac=True, bounds=box, method="L-BFGS-B")
                            for s0 in x_seeds)
            for res in outcomes:
                if res.success and (winner is None or np.squeeze(res.fun) < winner[1]):
                    winner = (res.x, np.squeeze(res.fun))
            return winner
        batched = self.batched_fd and getattr(self, "_fused", None) is not None
        value_and_grad = _fd_value_and_grad(acq, box) if batched else None
        winner = None
        if batched and self.lockstep and len(x_seeds) > 1:
            if self.lockstep != "threads" and lbfgsb_lockstep.driver_available() and not np.any(box[:, 0] == box[:, 1]):
                outcomes = lbfgsb_lockstep.minimize_many(acq, x_seeds, box, single_thread_blas=self._objective_on_device(getattr(self, "_fused", None)))     # SciPy's setulb driven directly
            else:
                outcomes = _polish_in_lockstep(acq, x_seeds, box)               # public API only: one thread per run
        elif batched:
            outcomes = (minimize(value_and_grad, start, jac=True, bounds=box, method="L-BFGS-B") for start in x_seeds)
        else:
            outcomes = (minimize(acq, start, bounds=box, method="L-BFGS-B") for start in x_seeds)
        for res in outcomes:
            if res.success and (winner is None or np.squeeze(res.fun) < winner[1]):
                winner = (res.x, np.squeeze(res.fun))
        return winner

    def _evolve_mixed(self, acq, space, x_seeds, random_state, is_cont, box):
        import scipy
        from packaging import version
        from scipy.optimize._differentialevolution import DifferentialEvolutionSolver  # as acquisition.py:32

        population = space.random_sample(15 * len(space.bounds), random_state=random_state)
        n_keep = min(len(x_seeds), len(population))
        if n_keep > 0:
            population[:n_keep] = x_seeds[:n_keep]
        rng_kw = "seed" if version.parse(scipy.__version__) < version.parse("1.15.0") else "rng"
        solver = DifferentialEvolutionSolver(func=acq, bounds=space.bounds, polish=False, init=population,
                                             **{rng_kw: random_state})
        found = solver.solve()
        if not found.success:
            raise RuntimeError(f"Differential evolution optimization failed. Message: {found.message}")
        x_best, f_best = found.x, np.squeeze(found.fun)
        if any(is_cont):
            frozen = x_best.copy()

            def along_continuous(z, frozen=frozen):
                frozen[is_cont] = z
                return acq(frozen)

            res = minimize(along_continuous, x_best[is_cont], bounds=box)
            if res.success and np.squeeze(res.fun) < f_best:
                frozen[is_cont] = res.x
                x_best, f_best = frozen, np.squeeze(res.fun)
        return x_best, f_best


class _DecayingParameter:
    """kappa / xi schedule shared by the stock policies (acquisition.py:541-554, 721-734, 910-923)."""

    _param_name = ""

    def _init_decay(self, exploration_decay, exploration_decay_delay):
        if exploration_decay is not None and not (0 < exploration_decay <= 1):
            raise ValueError("exploration_decay must be greater than 0 and less than or equal to 1.")
        if exploration_decay_delay is not None and (
                not isinstance(exploration_decay_delay, int) or exploration_decay_delay < 0):
            raise ValueError("exploration_decay_delay must be an integer greater than or equal to 0.")
        self.exploration_decay = exploration_decay
        self.exploration_decay_delay = exploration_decay_delay

    def decay_exploration(self) -> None:
        """Called at the end of every suggest(): multiply the parameter by the decay once the delay has passed."""
        if self.exploration_decay is None:
            return
        if self.exploration_decay_delay is None or self.exploration_decay_delay <= self.i:
            setattr(self, self._param_name, getattr(self, self._param_name) * self.exploration_decay)

    def get_acquisition_params(self):
        return {self._param_name: getattr(self, self._param_name), "exploration_decay": self.exploration_decay,
                "exploration_decay_delay": self.exploration_decay_delay}

    def set_acquisition_params(self, params):
        setattr(self, self._param_name, params[self._param_name])
        self.exploration_decay = params["exploration_decay"]
        self.exploration_decay_delay = params["exploration_decay_delay"]

    def _acq_param(self):
        return float(getattr(self, self._param_name))


class UpperConfidenceBound(_DecayingParameter, AcquisitionFunction):
    """UCB(x) = mu(x) + kappa sigma(x)  (bayes_opt/acquisition.py:423-600)."""

    _acq_kind = E.UCB
    _param_name = "kappa"

    def __init__(self, kappa: float = 2.576, exploration_decay=None, exploration_decay_delay=None,
                 random_state=None) -> None:
        if kappa < 0:
            raise ValueError("kappa must be greater than or equal to 0.")
        self._init_decay(exploration_decay, exploration_decay_delay)
        AcquisitionFunction.__init__(self, random_state=random_state)
        self.kappa = kappa

    def base_acq(self, mean, std):
        return mean + self.kappa * std

    def base_acq_grad(self, mean, std, dmean, dstd):
        return mean + self.kappa * std, dmean + self.kappa * dstd

    def suggest(self, gp, target_space, n_random=None, n_smart: int = 10, fit_gp: bool = True, random_state=None):
        if target_space.constraint is not None:
            raise ConstraintNotSupportedError(
                f"Received constraints, but acquisition function {type(self)} "
                "does not support constrained optimization.")
        x = super().suggest(gp=gp, target_space=target_space, n_random=n_random, n_smart=n_smart, fit_gp=fit_gp,
                            random_state=random_state)
        self.decay_exploration()
        return x


class _ImprovementBased(_DecayingParameter, AcquisitionFunction):
    """Shared by POI and EI: they need y_max = the best feasible, in-bounds observation."""

    _param_name = "xi"

    def __init__(self, xi: float, exploration_decay=None, exploration_decay_delay=None, random_state=None) -> None:
        self._init_decay(exploration_decay, exploration_decay_delay)
        AcquisitionFunction.__init__(self, random_state=random_state)
        self.xi = xi
        self.y_max = None

    def _need_y_max(self):
        if self.y_max is None:
            raise ValueError("y_max is not set. If you are calling this method outside "
                             "of suggest(), ensure y_max is set, or set it manually.")

    def suggest(self, gp, target_space, n_random=None, n_smart: int = 10, fit_gp: bool = True, random_state=None):
        incumbent = target_space._target_max()
        if incumbent is None and not target_space.empty:
            raise NoValidPointRegisteredError(
                "Cannot suggest a point without an allowed point. Use "
                "target_space.random_sample() to generate a point until "
                " at least one point that satisfies the constraints is found.")
        self.y_max = incumbent
        x = super().suggest(gp=gp, target_space=target_space, n_random=n_random, n_smart=n_smart, fit_gp=fit_gp,
                            random_state=random_state)
        self.decay_exploration()
        return x


class ProbabilityOfImprovement(_ImprovementBased):
    """POI(x) = Phi((mu - y_max - xi)/sigma)  (bayes_opt/acquisition.py:603-776)."""

    _acq_kind = E.POI

    def base_acq(self, mean, std):
        self._need_y_max()
        with np.errstate(divide="ignore", invalid="ignore"):
            return ndtr((mean - self.y_max - self.xi) / std)

    def base_acq_grad(self, mean, std, dmean, dstd):
        self._need_y_max()
        z = (mean - self.y_max - self.xi) / std
        return ndtr(z), (_norm_pdf(z) / std)[:, None] * (dmean - z[:, None] * dstd)


class ExpectedImprovement(_ImprovementBased):
    """EI(x) = a Phi(z) + sigma phi(z), a = mu - y_max - xi, z = a/sigma  (bayes_opt/acquisition.py:779-949)."""

    _acq_kind = E.EI

    def base_acq(self, mean, std):
        self._need_y_max()
        with np.errstate(divide="ignore", invalid="ignore"):
            a = mean - self.y_max - self.xi
            z = a / std
            return a * ndtr(z) + std * _norm_pdf(z)

    def base_acq_grad(self, mean, std, dmean, dstd):
        # d/dx [a Phi(z) + s phi(z)] = Phi(z) da + phi(z) ds   (the z-terms cancel)
        self._need_y_max()
        a = mean - self.y_max - self.xi
        z = a / std
        cdf, pdf = ndtr(z), _norm_pdf(z)
        return a * cdf + std * pdf, cdf[:, None] * dmean + pdf[:, None] * dstd


class GPHedge(AcquisitionFunction):
    """Portfolio of acquisition policies (bayes_opt/acquisition.py:1181-1360; Brochu et al., arXiv:1009.5419): every
    base policy nominates a point, one nominee is drawn with probability softmax(gains), and at the next step the gains
    grow by the new posterior mean at the previous nominees.  Same interface, state (`gains`, `previous_candidates`),
    parameter dict and RandomState consumption as the reference class.

    `share_candidates`:
      False (default)  the reference's arithmetic: base policy i draws its own `n_random // n_acq` candidates and runs
                       its own random + local stages through `base.suggest(..., fit_gp=False)` (:1306-1316) — with
                       engine-backed GPs each of those is the fused device stage; nominees and RandomState position are
                       the reference's.
      True             ONE candidate set of `n_random` points and ONE posterior pass (`gpbo_posterior` per model) serve
                       all base policies; each policy then costs only its acquisition + arg-best pass
                       (`gpbo_acq_argbest`, HBM-bound, ~0.1 ms per 2^20 candidates) and its local search.  n_acq times
                       fewer posterior flops for n_acq times MORE candidates per policy; the candidate stream differs
                       from the reference's (one draw of n_random instead of n_acq draws of n_random // n_acq), so this
                       mode is labelled non-parity.  Needs engine-backed GPs and stock base policies; otherwise the
                       per-policy path runs.
    """

    def __init__(self, base_acquisitions, random_state=None, share_candidates: bool = False) -> None:
        super().__init__(random_state)
        self.base_acquisitions = list(base_acquisitions)
        self.n_acq = len(self.base_acquisitions)
        self.gains = np.zeros(self.n_acq)
        self.previous_candidates = None
        self.share_candidates = share_candidates

    def base_acq(self, *args, **kwargs):
        raise TypeError("GPHedge base acquisition function is ambiguous."
                        " You may use self.base_acquisitions[i].base_acq(mean, std)"
                        " to get the base acquisition function for the i-th acquisition.")

    @staticmethod
    def _softmax_cumsum(g):
        z = np.exp(g - np.max(g))          # scipy.special.softmax
        return np.cumsum(z / z.sum())

    def _sample_idx_from_softmax_gains(self, random_state) -> int:
        return int(np.argmax(random_state.rand() <= self._softmax_cumsum(self.gains)))    # first True

    def _update_gains(self, gp) -> None:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            self.gains += gp.predict(self.previous_candidates)
        self.previous_candidates = None

    def _nominate_shared(self, chain, gp, target_space, n_random, n_smart, rng):
        """All base policies over ONE resident candidate set and ONE posterior pass."""
        eng = chain[0]._engine()
        k = n_smart // self.n_acq
        n = max(n_random, k, 1)
        if _reference_stream_on_device(chain, target_space, rng, n):
            eng.generate_candidates_like(n, target_space.bounds[:, 0], target_space.bounds[:, 1], rng)
        else:
            eng.set_candidates(chain[0]._tx(target_space.random_sample(n, random_state=rng)))
        for model in chain:
            model.posterior_resident()                       # ONE pass per GP, shared by every policy below
        lb, ub = (target_space.constraint._lb, target_space.constraint._ub) if len(chain) > 1 else (None, None)
        d = target_space.bounds.shape[0]
        picks_all = []
        for base in self.base_acquisitions:                  # selection only: mu / sd stay on the device
            if isinstance(base, _ImprovementBased):
                base.y_max = target_space._target_max()
                if base.y_max is None and not target_space.empty:
                    raise NoValidPointRegisteredError("Cannot suggest a point without an allowed point.")
            y_max = getattr(base, "y_max", None)
            best, best_val, picks, _v, _ = eng.acq_argbest(base._acq_kind, base._acq_param(), 0.0 if y_max is None else y_max,
                                                           lb, ub, k_seeds=min(k, _MAX_DEVICE_SEEDS))
            picks = picks[picks >= 0]
            rows = eng.get_candidate_rows(np.concatenate([[int(best)], picks]), d)   # now: a local search re-uses the buffer
            picks_all.append((float(best_val), rows))
        nominees = []
        for base, (best_val, rows) in zip(self.base_acquisitions, picks_all):
            x_rand, seeds = rows[0], rows[1:]
            base.i += 1
            x = x_rand
            if k:
                base._fused, base._fused_constraint = chain, (target_space.constraint if len(chain) > 1 else None)
                try:
                    acq = base._get_acq(gp=gp, constraint=target_space.constraint)
                    x_loc, f_loc = base._smart_minimize(acq, target_space, x_seeds=seeds, random_state=rng)
                finally:
                    base._fused = base._fused_constraint = None
                if best_val > f_loc:
                    x = x_loc
            base.decay_exploration()
            nominees.append(x)
        return nominees

    def suggest(self, gp, target_space, n_random: int = 10_000, n_smart: int = 10, fit_gp: bool = True, random_state=None):
        if len(target_space) == 0:
            raise TargetSpaceEmptyError(
                "Cannot suggest a point without previous samples. Use "
                " target_space.random_sample() to generate a point and "
                " target_space.probe(*) to evaluate it.")
        self.i += 1
        rng = ensure_rng(random_state)
        if fit_gp:
            self._fit_gp(gp=gp, target_space=target_space)
        if self.previous_candidates is not None:
            self._update_gains(gp)
        chain = _fused_models(gp, target_space.constraint) if self.share_candidates else None
        shared = (chain is not None and all(getattr(b, "_acq_kind", None) is not None for b in self.base_acquisitions)
                  and not (target_space.constraint is not None
                           and any(isinstance(b, UpperConfidenceBound) for b in self.base_acquisitions)))
        if shared:
            x_max = self._nominate_shared(chain, gp, target_space, n_random, n_smart, rng)
        else:
            x_max = [base.suggest(gp=gp, target_space=target_space, n_random=n_random // self.n_acq,
                                  n_smart=n_smart // self.n_acq, fit_gp=False, random_state=rng)
                     for base in self.base_acquisitions]
        self.previous_candidates = np.array(x_max)
        idx = self._sample_idx_from_softmax_gains(random_state=rng)
        if not getattr(target_space, "_allow_duplicate_points", True) and x_max[idx] in target_space:
            fresh = [t for t, x in enumerate(x_max) if x not in target_space]
            if fresh:      # every nominee a duplicate: keep the draw (the caller decides what a duplicate means)
                idx = fresh[int(np.argmax(rng.rand() <= self._softmax_cumsum(self.gains[fresh])))]
        return x_max[idx]

    def get_acquisition_params(self):
        return {"base_acquisitions_params": [b.get_acquisition_params() for b in self.base_acquisitions],
                "gains": self.gains.tolist(),
                "previous_candidates": None if self.previous_candidates is None else self.previous_candidates.tolist()}

    def set_acquisition_params(self, params):
        for base, p in zip(self.base_acquisitions, params["base_acquisitions_params"]):
            base.set_acquisition_params(p)
        self.gains = np.array(params["gains"])
        pc = params["previous_candidates"]
        self.previous_candidates = None if pc is None else np.array(pc)
