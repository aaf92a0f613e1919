"""Acquisition functions with the random-search stage fused on the GPU (seam B1).

Drop-in for `BayesianOptimization(acquisition_function=...)`: the driver calls exactly
`acq.suggest(gp=, target_space=, fit_gp=True, random_state=)` (bayes_opt/bayesian_optimization.py:329-331),
`acq._fit_gp(gp, space)` (:236) and `get/set_acquisition_params` (:430, :471).  The classes mirror the
reference's public names, arguments, error behaviour and RandomState consumption order
(bayes_opt/acquisition.py:56-420 base class; :423-600 UCB; :603-776 POI; :779-949 EI); they do not
import bayes_opt, so they also run on a box where it is not installed.

What moves to the device: when `gp` (and every constraint GP) is a `HipGPR` on one engine, the whole
`_random_sample_minimize` body after candidate sampling (acquisition.py:311-317) — posterior mean/std
for M candidates, -acq(x) [* p_constraint], argmin, min, argsort[:k] — runs as HIP kernels and only
the arg-best record and the k seeds come back.  For any other `gp` object (duck-typed mocks, plain
sklearn estimators) the same host orchestration evaluates the `_get_acq` closure, as the reference does.
"""
from __future__ import annotations

import abc
import warnings

import numpy as np
from scipy.optimize import minimize
from scipy.special import ndtr

from . import engine as E
from .gpr import HipGPR
from .space import ensure_rng

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


def _norm_pdf(x):
    return np.exp(-(x**2) / 2.0) / _SQRT_2PI  # scipy/stats/_continuous_distns.py:360-362


_FD_EPS = 1e-8                       # scipy.optimize._lbfgsb_py._minimize_lbfgsb: eps=1e-8 -> abs_step
_SQRT_EPS = np.finfo(np.float64).eps ** 0.5


def _fd_value_and_grad(acq, bounds):
    """value-and-gradient callable reproducing, in ONE batched acq() call per iteration, the forward
    differences SciPy's L-BFGS-B forms one point at a time when `jac` is not given
    (scipy/optimize/_numdiff.py: approx_derivative(method="2-point", abs_step=1e-8, bounds) ->
    _adjust_scheme_to_bounds(..., "1-sided") -> _dense_difference): same steps, same divisions, so the
    optimiser sees the same (f, g) and walks the same path as the reference's
    `minimize(acq, x_try, bounds=..., method="L-BFGS-B")` (bayes_opt/acquisition.py:365-366)."""
    lb, ub = bounds[:, 0].astype(float), bounds[:, 1].astype(float)

    def fun(x):
        x0 = np.asarray(x, dtype=np.float64)
        d = x0.shape[0]
        sign_x0 = (x0 >= 0).astype(float) * 2 - 1
        h = np.full(d, _FD_EPS)
        dx = (x0 + h) - x0
        h = np.where(dx == 0, _SQRT_EPS * sign_x0 * np.maximum(1.0, np.abs(x0)), h)
        if not np.all((lb == -np.inf) & (ub == np.inf)):
            lower_dist, upper_dist = x0 - lb, ub - x0
            xs = x0 + h
            violated = (xs < lb) | (xs > ub)
            fitting = np.abs(h) <= np.maximum(lower_dist, upper_dist)
            h = h.copy()
            h[violated & fitting] *= -1
            forward = (upper_dist >= lower_dist) & ~fitting
            h[forward] = upper_dist[forward]
            backward = (upper_dist < lower_dist) & ~fitting
            h[backward] = -lower_dist[backward]
        pts = np.empty((d + 1, d))
        pts[0] = x0
        pts[1:] = x0 + np.diag(h)
        ys = np.asarray(acq(pts), dtype=np.float64)
        dxs = pts[1:][np.arange(d), np.arange(d)] - x0
        return ys[0], (ys[1:] - ys[0]) / dxs

    return fun


def _fused_models(gp, constraint):
    """[target, constraint GPs...] if all are HipGPRs on one engine in slots 0..n, else None."""
    if not isinstance(gp, HipGPR) or gp.slot != 0:
        return None
    models = [gp]
    if constraint is not None:
        cms = getattr(constraint, "_model", None)
        if cms is None or len(cms) + 1 > 8:
            return None
        for j, m in enumerate(cms):
            if not isinstance(m, HipGPR) or m._engine() is not gp._engine() or m.slot != j + 1:
                return None
            models.append(m)
    return models


class AcquisitionFunction(abc.ABC):
    """Base class (bayes_opt/acquisition.py:56-420)."""

    #: default number of random candidates; the reference hard-codes 10_000 (acquisition.py:120)
    default_n_random = 10_000
    #: with engine-backed GPs, form L-BFGS-B's finite-difference gradient in one batched device call per
    #: iteration (d + 1 points) instead of d + 1 single-point calls; same numbers, ~d times fewer launches
    batched_fd = True
    #: THROUGHPUT MODE (off by default): draw the random-stage candidates on the device with a Philox generator
    #: instead of space.random_sample(); removes the host sampling and the upload but is NOT the reference's
    #: RandomState stream (two 31-bit integers are drawn from it as the device seed), so suggestions differ
    device_sampling = False
    _acq_kind: int | None = None

    def __init__(self, random_state=None) -> None:
        if random_state is not None:
            warnings.warn(
                "Providing a random_state to an acquisition function during initialization is deprecated "
                "and will be ignored. The random_state is instead provided automatically during the "
                "suggest() call.", DeprecationWarning, stacklevel=2)
        self.i = 0

    @abc.abstractmethod
    def base_acq(self, *args, **kwargs):
        """Provide access to the base acquisition function."""

    def _acq_param(self) -> float:
        raise NotImplementedError

    def _fit_gp(self, gp, target_space) -> None:  # acquisition.py:79-86
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gp.fit(target_space.params, target_space.target)
            if target_space.constraint is not None:
                target_space.constraint.fit(target_space.params, target_space._constraint_values)

    def get_acquisition_params(self):
        raise NotImplementedError("Custom AcquisitionFunction subclasses must implement their own get_acquisition_params method.")

    def set_acquisition_params(self, params):
        raise NotImplementedError("Custom AcquisitionFunction subclasses must implement their own set_acquisition_params method.")

    def suggest(self, gp, target_space, n_random: int | None = None, n_smart: int = 10, fit_gp: bool = True,
                random_state=None):
        """acquisition.py:116-169.  `n_random=None` means `self.default_n_random` (10_000 in the reference)."""
        if n_random is None:
            n_random = self.default_n_random
        random_state = ensure_rng(random_state)
        if len(target_space) == 0:
            raise TargetSpaceEmptyError(
                "Cannot suggest a point without previous samples. Use "
                " target_space.random_sample() to generate a point and "
                " target_space.probe(*) to evaluate it.")
        self.i += 1
        if fit_gp:
            self._fit_gp(gp=gp, target_space=target_space)
        acq = self._get_acq(gp=gp, constraint=target_space.constraint)
        self._fused = _fused_models(gp, target_space.constraint) if self._acq_kind is not None else None
        try:
            return self._acq_min(acq, target_space, n_random=n_random, n_smart=n_smart, random_state=random_state)
        finally:
            self._fused = None

    def _get_acq(self, gp, constraint=None):  # acquisition.py:171-219
        dim = gp.X_train_.shape[1]
        if constraint is not None:
            def acq(x):
                x = x.reshape(-1, dim)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    mean, std = gp.predict(x, return_std=True)
                    p_constraints = constraint.predict(x)
                return -1 * self.base_acq(mean, std) * p_constraints
        else:
            def acq(x):
                x = x.reshape(-1, dim)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    mean, std = gp.predict(x, return_std=True)
                return -1 * self.base_acq(mean, std)
        return acq

    def _acq_min(self, acq, space, random_state, n_random: int = 10_000, n_smart: int = 10):  # :221-272
        if n_random == 0 and n_smart == 0:
            raise ValueError("Either n_random or n_smart needs to be greater than 0.")
        x_min_r, min_acq_r, x_seeds = self._random_sample_minimize(
            acq, space, random_state, n_random=max(n_random, n_smart), n_x_seeds=n_smart)
        if n_smart:
            x_min_s, min_acq_s = self._smart_minimize(acq, space, x_seeds=x_seeds, random_state=random_state)
            if min_acq_r > min_acq_s:
                return x_min_s
        return x_min_r

    def _random_sample_minimize(self, acq, space, random_state, n_random: int, n_x_seeds: int = 0):  # :274-320
        if n_random == 0:
            return None, np.inf, space.random_sample(n_x_seeds, random_state=random_state)
        fused = getattr(self, "_fused", None)
        if (fused is not None and n_x_seeds <= 64 and self.device_sampling and fused[0].transform is None):
            seed = int(random_state.randint(0, 2**31 - 1)) | (int(random_state.randint(0, 2**31 - 1)) << 31)
            return self._device_minimize(fused, space, None, n_x_seeds, n_random=n_random, seed=seed)
        x_tries = space.random_sample(n_random, random_state=random_state)
        if fused is not None and n_x_seeds <= 64:
            return self._device_minimize(fused, space, x_tries, n_x_seeds)
        ys = acq(x_tries)
        x_min = x_tries[ys.argmin()]
        min_acq = ys.min()
        if n_x_seeds != 0:
            idxs = np.argsort(ys)[:n_x_seeds]
            x_seeds = x_tries[idxs]
        else:
            x_seeds = []
        return x_min, min_acq, x_seeds

    def _device_minimize(self, models, space, x_tries, n_x_seeds, n_random=None, seed=None):
        """The body of _random_sample_minimize after sampling, on the GPU (kernels K5-K8 of SURVEY.md §2.1).
        x_tries=None: the candidates are generated on the device too (device_sampling)."""
        gp = models[0]
        eng = gp._engine()
        if x_tries is None:
            eng.generate_candidates(n_random, space.bounds[:, 0], space.bounds[:, 1], seed)
        else:
            eng.set_candidates(gp._tx(x_tries))
        for m in models:
            m.posterior_resident()
        lb = ub = None
        if len(models) > 1:
            lb, ub = space.constraint._lb, space.constraint._ub
        y_max = getattr(self, "y_max", None)
        bi, bv, si, _sv, _ = eng.acq_argbest(self._acq_kind, self._acq_param(), 0.0 if y_max is None else y_max,
                                             lb, ub, k_seeds=n_x_seeds)
        si = si[si >= 0]
        if x_tries is None:
            rows = eng.get_candidate_rows(np.concatenate([[bi], si]), space.bounds.shape[0])
            return rows[0], bv, (rows[1:] if n_x_seeds else [])
        x_min = x_tries[bi]
        x_seeds = x_tries[si] if n_x_seeds else []
        return x_min, bv, x_seeds

    def _smart_minimize(self, acq, space, x_seeds, random_state):  # acquisition.py:322-420
        continuous_dimensions = space.continuous_dimensions
        continuous_bounds = space.bounds[continuous_dimensions]
        min_acq = None
        x_min = None
        if all(continuous_dimensions):
            batched = self.batched_fd and getattr(self, "_fused", None) is not None
            fg = _fd_value_and_grad(acq, continuous_bounds) if batched else None
            for x_try in x_seeds:
                if batched:
                    res = minimize(fg, x_try, jac=True, bounds=continuous_bounds, method="L-BFGS-B")
                else:
                    res = minimize(acq, x_try, bounds=continuous_bounds, method="L-BFGS-B")
                if not res.success:
                    continue
                if min_acq is None or np.squeeze(res.fun) < min_acq:
                    x_min = res.x
                    min_acq = np.squeeze(res.fun)
        else:
            from scipy.optimize._differentialevolution import DifferentialEvolutionSolver  # as acquisition.py:32

            xinit = space.random_sample(15 * len(space.bounds), random_state=random_state)
            if len(x_seeds) > 0:
                n_seeds = min(len(x_seeds), len(xinit))
                xinit[:n_seeds] = x_seeds[:n_seeds]
            import scipy
            from packaging import version

            de_parameters = {"func": acq, "bounds": space.bounds, "polish": False, "init": xinit}
            if version.parse(scipy.__version__) < version.parse("1.15.0"):
                de_parameters["seed"] = random_state
            else:
                de_parameters["rng"] = random_state
            de = DifferentialEvolutionSolver(**de_parameters)
            res_de = de.solve()
            if not res_de.success:
                raise RuntimeError(f"Differential evolution optimization failed. Message: {res_de.message}")
            x_min = res_de.x
            min_acq = np.squeeze(res_de.fun)
            if any(continuous_dimensions):
                x_try = x_min.copy()

                def continuous_acq(x, x_try=x_try):
                    x_try[continuous_dimensions] = x
                    return acq(x_try)

                res = minimize(continuous_acq, x_min[continuous_dimensions], bounds=continuous_bounds)
                if res.success and np.squeeze(res.fun) < min_acq:
                    x_try[continuous_dimensions] = res.x
                    x_min = x_try
                    min_acq = np.squeeze(res.fun)
        if min_acq is None:
            min_acq = np.inf
            x_min = np.array([np.nan] * space.bounds.shape[0])
        return np.clip(x_min, space.bounds[:, 0], space.bounds[:, 1]), min_acq


def _check_decay(exploration_decay, exploration_decay_delay):
    if exploration_decay is not None and not (0 < exploration_decay <= 1):
        raise ValueError("exploration_decay must be greater than 0 and less than or equal to 1.")
    if exploration_decay_delay is not None and (
            not isinstance(exploration_decay_delay, int) or exploration_decay_delay < 0):
        raise ValueError("exploration_decay_delay must be an integer greater than or equal to 0.")


class UpperConfidenceBound(AcquisitionFunction):
    """UCB(x) = mu(x) + kappa sigma(x)  (bayes_opt/acquisition.py:423-600)."""

    _acq_kind = E.UCB

    def __init__(self, kappa: float = 2.576, exploration_decay=None, exploration_decay_delay=None,
                 random_state=None) -> None:
        if kappa < 0:
            raise ValueError("kappa must be greater than or equal to 0.")
        _check_decay(exploration_decay, exploration_decay_delay)
        super().__init__(random_state=random_state)
        self.kappa = kappa
        self.exploration_decay = exploration_decay
        self.exploration_decay_delay = exploration_decay_delay

    def _acq_param(self):
        return float(self.kappa)

    def base_acq(self, mean, std):
        return mean + self.kappa * std

    def suggest(self, gp, target_space, n_random=None, n_smart: int = 10, fit_gp: bool = True, random_state=None):
        if target_space.constraint is not None:
            raise ConstraintNotSupportedError(
                f"Received constraints, but acquisition function {type(self)} "
                "does not support constrained optimization.")
        x_max = super().suggest(gp=gp, target_space=target_space, n_random=n_random, n_smart=n_smart,
                                fit_gp=fit_gp, random_state=random_state)
        self.decay_exploration()
        return x_max

    def decay_exploration(self) -> None:
        if self.exploration_decay is not None and (
                self.exploration_decay_delay is None or self.exploration_decay_delay <= self.i):
            self.kappa = self.kappa * self.exploration_decay

    def get_acquisition_params(self):
        return {"kappa": self.kappa, "exploration_decay": self.exploration_decay,
                "exploration_decay_delay": self.exploration_decay_delay}

    def set_acquisition_params(self, params):
        self.kappa = params["kappa"]
        self.exploration_decay = params["exploration_decay"]
        self.exploration_decay_delay = params["exploration_decay_delay"]


class _ImprovementBased(AcquisitionFunction):
    def __init__(self, xi: float, exploration_decay=None, exploration_decay_delay=None, random_state=None) -> None:
        _check_decay(exploration_decay, exploration_decay_delay)
        super().__init__(random_state=random_state)
        self.xi = xi
        self.exploration_decay = exploration_decay
        self.exploration_decay_delay = exploration_decay_delay
        self.y_max = None

    def _acq_param(self):
        return float(self.xi)

    def _check_y_max(self):
        if self.y_max is None:
            raise ValueError("y_max is not set. If you are calling this method outside "
                             "of suggest(), ensure y_max is set, or set it manually.")

    def suggest(self, gp, target_space, n_random=None, n_smart: int = 10, fit_gp: bool = True, random_state=None):
        y_max = target_space._target_max()
        if y_max is None and not target_space.empty:
            raise NoValidPointRegisteredError(
                "Cannot suggest a point without an allowed point. Use "
                "target_space.random_sample() to generate a point until "
                " at least one point that satisfies the constraints is found.")
        self.y_max = y_max
        x_max = super().suggest(gp=gp, target_space=target_space, n_random=n_random, n_smart=n_smart,
                                fit_gp=fit_gp, random_state=random_state)
        self.decay_exploration()
        return x_max

    def decay_exploration(self) -> None:
        if self.exploration_decay is not None and (
                self.exploration_decay_delay is None or self.exploration_decay_delay <= self.i):
            self.xi = self.xi * self.exploration_decay

    def get_acquisition_params(self):
        return {"xi": self.xi, "exploration_decay": self.exploration_decay,
                "exploration_decay_delay": self.exploration_decay_delay}

    def set_acquisition_params(self, params):
        self.xi = params["xi"]
        self.exploration_decay = params["exploration_decay"]
        self.exploration_decay_delay = params["exploration_decay_delay"]


class ProbabilityOfImprovement(_ImprovementBased):
    """POI(x) = Phi((mu - y_max - xi)/sigma)  (bayes_opt/acquisition.py:603-776)."""

    _acq_kind = E.POI

    def base_acq(self, mean, std):
        self._check_y_max()
        with np.errstate(divide="ignore", invalid="ignore"):
            z = (mean - self.y_max - self.xi) / std
            return ndtr(z)


class ExpectedImprovement(_ImprovementBased):
    """EI(x) = a Phi(z) + sigma phi(z), a = mu - y_max - xi, z = a/sigma  (bayes_opt/acquisition.py:779-949)."""

    _acq_kind = E.EI

    def base_acq(self, mean, std):
        self._check_y_max()
        with np.errstate(divide="ignore", invalid="ignore"):
            a = mean - self.y_max - self.xi
            z = a / std
            return a * ndtr(z) + std * _norm_pdf(z)
