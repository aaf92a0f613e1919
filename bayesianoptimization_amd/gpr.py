"""HipGPR — the sklearn-estimator duck type of `optimizer._gp`, backed by the HIP engine.

Seam B2 of SURVEY.md §8b.  bayes_opt touches only a handful of members of its
`GaussianProcessRegressor` (bayes_opt/acquisition.py:84,195,205,216; bayes_opt/constraint.py:148-151,
192,200,213,238-242; bayes_opt/bayesian_optimization.py:124-130,238,403-407,440-443,490):
`fit`, `predict(return_std/return_cov)`, `X_train_`, `n_features_in_`, `set_params/get_params`,
`kernel`, `kernel_`, `alpha`, `normalize_y`, `n_restarts_optimizer`.  HipGPR subclasses sklearn's
estimator so all of that keeps its meaning, and replaces the two numeric stages:

  * fit: the theta search is sklearn's own code path (L-BFGS-B over `log_marginal_likelihood`, on the
    host — SURVEY.md §8f-1 "next" row), restated here line by line from
    sklearn/gaussian_process/_gpr.py:225-338 so that the shared RandomState is consumed identically;
    the fixed-theta tail (_gpr.py:346-364: K + alpha I, Cholesky, alpha_) runs on the GPU.
  * predict (fitted, mean/std): _gpr.py:443-494 on the GPU.

Supported on the HIP path: Matern(nu=2.5) and RBF (optionally wrapped by bayes_opt's `wrap_kernel`,
optionally scaled by a fixed-at-1 ConstantKernel as in sklearn's default), scalar `alpha`, 1-D targets.
Anything else — `set_gp_params(kernel=Matern(nu=1.5))` on an accelerated optimizer
(bayes_opt/bayesian_optimization.py:403-407), a per-sample `alpha`, several targets — is outside the device path: the
estimator then IS its base class for that fit (scikit-learn's own `fit` / `predict`, the reference's arithmetic and
RandomState consumption bit for bit) and says so with one UserWarning per estimator (SURVEY.md §2 "Third-party
kernels": degrade, do not raise).  That is the caller's own configuration running the reference's code, not a CPU
implementation of the device path: a supported model never leaves the GPU, and fails loudly without one.
"""
from __future__ import annotations

import warnings
import weakref
from operator import itemgetter

import numpy as np
from sklearn.base import clone
from sklearn.gaussian_process import GaussianProcessRegressor
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, Product
from sklearn.utils import check_random_state
from sklearn.utils.validation import validate_data

from ._lib import MAX_DIM
from .engine import MATERN25, GpEngine
from .engine import RBF as K_RBF

_shared_engines: dict = {}
LML_DEVICE_MIN_N = 1      # lml_on_device="auto": device from this many observations (see HipGPR.__init__; measured, round 4)


def shared_engine(device=0) -> GpEngine:
    """One context per device — or one device group per tuple of devices — per process (created on first use; raises
    without a GPU).  `device`: an int, or a sequence of ints for a single-process multi-GPU group (GroupEngine).
    The seams never read `last_timings()`, so this engine does not record HIP event pairs (`set_timing(False)`: a step of
    BASELINE config 1 is 91 instead of 119 us without its eight marker packets); `set_timing(True)` brings them back."""
    key = int(device) if np.isscalar(device) else tuple(int(x) for x in device)
    if key not in _shared_engines:
        if isinstance(key, tuple):
            from .engine import GroupEngine
            eng = GroupEngine(key)
        else:
            eng = GpEngine(key)
        eng.set_timing(False)
        _shared_engines[key] = eng
    return _shared_engines[key]


def describe_kernel(kernel):
    """(kind, length_scale array) for a supported kernel, else raise NotImplementedError.

    Accepts Matern(nu=2.5) / RBF, dynamic subclasses made by bayes_opt.parameter.wrap_kernel
    (parameter.py:457-495; MRO WrappedKernel -> Matern -> RBF), and `ConstantKernel(1, "fixed") * k`
    (sklearn's default kernel, _gpr.py:241-246).
    """
    k = kernel
    if isinstance(k, Product):
        c, inner = k.k1, k.k2
        if not isinstance(c, ConstantKernel):
            c, inner = k.k2, k.k1
        if not (isinstance(c, ConstantKernel) and c.constant_value == 1.0 and c.constant_value_bounds == "fixed"):
            raise NotImplementedError(f"HIP path supports only a fixed unit ConstantKernel factor, got {kernel!r}")
        k = inner
    if isinstance(k, Matern):
        if k.nu != 2.5:
            raise NotImplementedError(f"HIP path supports Matern(nu=2.5) only, got nu={k.nu}")
        kind = MATERN25
    elif isinstance(k, RBF):
        kind = K_RBF
    else:
        raise NotImplementedError(f"HIP path supports Matern(nu=2.5) and RBF kernels, got {type(kernel).__name__}")
    return kind, np.atleast_1d(np.asarray(k.length_scale, dtype=np.float64))


def _bare_length_scale_kernel(kernel) -> bool:
    """True for a bare Matern / RBF (or a wrap_kernel subclass of one) whose length scale is free: its theta IS log(length_scale)
    and its bounds ARE log(length_scale_bounds) (kernels.py: Hyperparameter("length_scale", "numeric", bounds, n_elements)), so the
    fit can read and write them as attributes instead of through Kernel.theta / .bounds / .n_dims — each of which walks
    dir(kernel) and inspect.signature (~0.05 ms a time, eight times per fit: a fifth of a small-N suggest's host time)."""
    cls = type(kernel)
    ok = _BARE_CLASSES.get(cls)
    if ok is None:      # once per class (wrap_kernel makes one class per optimizer): length_scale is its ONLY hyper-parameter
        ok = _BARE_CLASSES[cls] = issubclass(cls, RBF) and [a for a in dir(cls) if a.startswith("hyperparameter_")] == ["hyperparameter_length_scale"]
    return ok and not isinstance(kernel.length_scale_bounds, str)


_BARE_CLASSES = weakref.WeakKeyDictionary()      # (wrap_kernel's dynamic classes die with their optimizer)


def _length_scale_theta(kernel):
    """(theta, bounds) of such a kernel, as Kernel.theta / Kernel.bounds return them (kernels.py:285-343)."""
    ls = np.atleast_1d(np.asarray(kernel.length_scale, dtype=np.float64))
    b = np.atleast_2d(np.asarray(kernel.length_scale_bounds, dtype=np.float64))
    if ls.shape[0] > 1 and b.shape[0] == 1:
        b = np.repeat(b, ls.shape[0], 0)
    return np.log(ls), np.log(b)


def _set_length_scale_theta(kernel, theta):
    """Kernel.theta = theta for such a kernel (kernels.py:309-334: a scalar for one element, an array for several)."""
    theta = np.asarray(theta, dtype=np.float64)
    kernel.length_scale = np.exp(theta) if np.iterable(kernel.length_scale) and len(kernel.length_scale) > 1 else np.exp(theta[0])


class HipGPR(GaussianProcessRegressor):
    """GaussianProcessRegressor whose fixed-theta fit and posterior run on the MI355X engine."""

    def __init__(self, kernel=None, *, alpha=1e-10, optimizer="fmin_l_bfgs_b", n_restarts_optimizer=0,
                 normalize_y=False, copy_X_train=True, n_targets=None, random_state=None,
                 transform=None, engine=None, slot=0, lml_on_device="auto", precision="f64", incremental=True,
                 theta_lockstep=True):
        super().__init__(kernel=kernel, alpha=alpha, optimizer=optimizer,
                         n_restarts_optimizer=n_restarts_optimizer, normalize_y=normalize_y,
                         copy_X_train=copy_X_train, n_targets=n_targets, random_state=random_state)
        self.transform = transform  # host-side input transform (None = identity, the all-float case)
        self.engine = engine
        self.slot = slot
        # theta search: evaluate log_marginal_likelihood(theta, eval_gradient=True) on the GPU (True), with sklearn's host
        # code (False), or "auto" = on the GPU whenever the kernel is one the device evaluates.  Until round 4 "auto" meant
        # N >= 512, a threshold no measurement backed; scripts/archive/r04_lml_crossover.py (profiles/r04_lml_crossover.json, MI355X
        # + 256 host threads) finds no crossover to speak of: one value + gradient on the device costs 0.13 ms at N = 16 ..
        # 64 (host: 0.10 / 0.13 / 0.19 ms), 0.16 vs 0.44 at N = 128, 0.35 vs 10 at N = 512, and the whole default fit (5
        # restarts, lockstep lanes) is faster on the device at EVERY size: 1.0 vs 2.0 ms at N = 16, 3.2 vs 96 at N = 128,
        # 6.3 vs 1990 at N = 512.
        self.lml_on_device = lml_on_device
        # "f64" (reference arithmetic) or "f32": fp64 factorisation, fp32 posterior contraction (engine.F32)
        self.precision = precision
        # grow the device factorisation row by row (gpbo_fit_append, O(N^2) per new observation) when a fit repeats
        # the previous one with observations appended and the kernel hyper-parameters unchanged
        self.incremental = incremental
        # theta search with restarts: advance the independent L-BFGS-B runs together, their LML evaluations side by
        # side on the device (gpbo_lml_batch); same iterates and same RandomState draws as one run after another
        self.theta_lockstep = theta_lockstep

    # -- outside the device path ------------------------------------------------------------------
    #: True after a fit that the device path does not cover: every numeric method is then the base class's
    _host_mode = False

    def _unsupported_reason(self, kernel, y=None, X=None):
        """Why this configuration is outside the device path (None when it is inside)."""
        try:
            describe_kernel(kernel)
        except NotImplementedError as exc:
            return str(exc)
        if np.iterable(self.alpha):
            return "HIP path supports a scalar alpha only"
        if y is not None and np.ndim(y) == 2 and np.shape(y)[1] != 1:
            return "HIP path supports a single target"
        if X is not None:
            # the width the device sees is the TRANSFORMED one: a CategoricalParameter is one-hot in kernel space
            # (bayes_opt/parameter.py:434-449 through target_space.py:340-347), so a few wide categoricals pass GPBO_MAX_DIM
            # where the parameter count does not
            width = self._device_width(X)
            if width is not None and width > MAX_DIM:
                return f"HIP path supports up to {MAX_DIM} dimensions in kernel space, this space has {width}"
        return None

    def _device_width(self, X):
        """Number of columns `gpbo_fit` would receive for X (None when X is not a non-empty 2-D array: sklearn's validation
        will say so)."""
        try:
            X = np.asarray(X, dtype=np.float64)
            if X.ndim != 2 or X.shape[0] == 0:
                return None
            return int(self._tx(X[:1]).shape[1]) if self.transform is not None else int(X.shape[1])
        except (TypeError, ValueError):
            return None

    def _warn_host(self, reason, stacklevel=3):
        if self.__dict__.get("_host_warned") != reason:
            warnings.warn(f"{reason}: this model runs scikit-learn's GaussianProcessRegressor on the host "
                          "(the reference's path), not the HIP engine", UserWarning, stacklevel=stacklevel)
            self._host_warned = reason

    def set_params(self, **params):
        """sklearn's set_params; a kernel outside the device path is announced HERE — `BayesianOptimization.set_gp_params`
        (bayes_opt/bayesian_optimization.py:403-407) ends in this call, whereas the fits run inside suggest() with every
        warning silenced (acquisition.py:79-86)."""
        out = super().set_params(**params)
        if "kernel" in params and self.kernel is not None:
            reason = self._unsupported_reason(self.kernel)
            if reason is not None:
                self._warn_host(reason)
        return out

    def _fit_on_host(self, X, y, reason):
        """scikit-learn's own fit for a model the device path does not cover (one warning per estimator and reason)."""
        self._warn_host(reason, stacklevel=4)
        # The device state goes first (a host model must never read an engine slot through the lazy L_ / alpha_), the flag last:
        # if the base fit raises (NaN input, a kernel matrix that is not positive definite) the estimator is left UNFITTED —
        # no X_train_ of an earlier device fit next to `_host_mode` — and not as a host model without its attributes.
        self._held = None
        for k in ("_kind", "_ls", "_L_cache", "_alpha_cache", "log_marginal_likelihood_value_", "_lml_lazy", "X_train_", "y_train_"):
            self.__dict__.pop(k, None)
        self._host_mode = False
        out = GaussianProcessRegressor.fit(self, X, y)
        self._host_mode = True
        return out

    # -- plumbing ------------------------------------------------------------------------------
    def _engine(self) -> GpEngine:
        if self.engine is None:
            self.engine = shared_engine(0)
        return self.engine

    def _precision_code(self) -> int:
        if self.precision not in ("f64", "f32"):
            raise ValueError("precision must be 'f64' or 'f32'")
        return 1 if self.precision == "f32" else 0

    def _tx(self, X):
        X = np.asarray(X, dtype=np.float64)
        if self.transform is not None:
            X = np.asarray(self.transform(X), dtype=np.float64)
        return np.ascontiguousarray(X)

    @classmethod
    def from_sklearn(cls, gp: GaussianProcessRegressor, transform=None, engine=None, slot=0, precision="f64"):
        """Same hyper-parameters (and the same RandomState object) as an existing estimator."""
        p = gp.get_params(deep=False)
        return cls(kernel=p["kernel"], alpha=p["alpha"], optimizer=p["optimizer"],
                   n_restarts_optimizer=p["n_restarts_optimizer"], normalize_y=p["normalize_y"],
                   copy_X_train=p["copy_X_train"], n_targets=p.get("n_targets"),
                   random_state=p["random_state"], transform=transform, engine=engine, slot=slot, precision=precision)

    # -- log marginal likelihood ---------------------------------------------------------------------
    def _device_lml_ok(self, kernel) -> bool:
        if self.lml_on_device is False or not hasattr(self, "X_train_") or self._host_mode:
            return False
        if self.lml_on_device == "auto" and self.X_train_.shape[0] < LML_DEVICE_MIN_N:
            return False
        if np.iterable(self.alpha):
            return False
        try:
            _, ls = describe_kernel(kernel)
        except NotImplementedError:
            return False
        if _bare_length_scale_kernel(kernel):
            return True
        free = [h for h in kernel.hyperparameters if not h.fixed]
        return len(free) == 1 and free[0].name.endswith("length_scale") and kernel.n_dims == ls.shape[0]

    def log_marginal_likelihood(self, theta=None, eval_gradient=False, clone_kernel=True):
        """sklearn _gpr.py:537-652; with `theta` given and a supported kernel the value and the gradient
        with respect to log(length_scale) are computed on the device (gpbo_lml), otherwise by sklearn."""
        if theta is None or not self._device_lml_ok(self.kernel_):
            return super().log_marginal_likelihood(theta, eval_gradient=eval_gradient, clone_kernel=clone_kernel)
        if clone_kernel:
            kernel = self.kernel_.clone_with_theta(theta)
        else:
            kernel = self.kernel_
            kernel.theta = theta
        kind, ls = describe_kernel(kernel)
        out = self._engine().lml(self._tx(self.X_train_), self.y_train_, kind, ls, float(self.alpha),
                                 eval_gradient=eval_gradient, slot=self.slot)
        self.__dict__.pop("_L_cache", None)      # the slot's factorisation now belongs to this theta
        self.__dict__.pop("_alpha_cache", None)
        if not getattr(self, "_in_fit", False) and hasattr(self, "_kind"):
            # called on a fitted model: gpbo_lml reused the slot's buffers, so restore the fit
            self._held = None
            self._device_fit_tail()
        return out

    def _ensure_resident(self):
        """Make sure the engine slot holds THIS estimator's factorisation before reading it.  Slots are shared state:
        another accelerated optimizer, a clone() of this estimator or an LML evaluation may have refitted the slot since
        our last fit (the reference gives every estimator its own L_/alpha_).  The engine counts every rewrite of a
        slot; a serial other than the one our fit got back means the slot is someone else's -> refit from X_train_."""
        if getattr(self, "_in_fit", False) or not hasattr(self, "_kind"):
            return
        held = self.__dict__.get("_held")
        eng = self._engine()
        if held is None or held["engine"] is not eng or eng.fit_serial(self.slot) != held["serial"]:
            self._held = None
            self._device_fit_tail()

    # lazily fetched parity attributes -------------------------------------------------------------
    @property
    def L_(self):
        """Lower Cholesky factor, Fortran-ordered with a zero upper triangle like scipy's (gp.L_)."""
        if "_L_cache" not in self.__dict__:
            if not hasattr(self, "X_train_"):
                raise AttributeError("L_")
            self._ensure_resident()
            self.__dict__["_L_cache"] = np.asfortranarray(self._engine().get_L(self.X_train_.shape[0], self.slot))
        return self.__dict__["_L_cache"]

    @L_.setter
    def L_(self, v):
        self.__dict__["_L_cache"] = v

    @property
    def alpha_(self):
        if "_alpha_cache" not in self.__dict__:
            if not hasattr(self, "X_train_"):
                raise AttributeError("alpha_")
            self._ensure_resident()
            self.__dict__["_alpha_cache"] = self._engine().get_alpha(self.X_train_.shape[0], self.slot)
        return self.__dict__["_alpha_cache"]

    @alpha_.setter
    def alpha_(self, v):
        self.__dict__["_alpha_cache"] = v

    @property
    def log_marginal_likelihood_value_(self):
        """sklearn sets this in fit() also when no search runs (`log_marginal_likelihood(kernel_.theta)`, _gpr.py:339-342);
        nothing on the suggest() path reads it, so on the fixed-theta path it is evaluated on first access."""
        if "log_marginal_likelihood_value_" not in self.__dict__:
            if not self.__dict__.get("_lml_lazy"):
                raise AttributeError("log_marginal_likelihood_value_")
            self.__dict__["log_marginal_likelihood_value_"] = GaussianProcessRegressor.log_marginal_likelihood(
                self, self.kernel_.theta, clone_kernel=False)
        return self.__dict__["log_marginal_likelihood_value_"]

    @log_marginal_likelihood_value_.setter
    def log_marginal_likelihood_value_(self, v):
        self.__dict__["log_marginal_likelihood_value_"] = v

    # -- fit -------------------------------------------------------------------------------------
    def fit(self, X, y):
        """Fit; mirrors sklearn _gpr.py:225-365 with the fixed-theta tail on the GPU."""
        if self.kernel is None:  # _gpr.py:241-246
            self.kernel_ = ConstantKernel(1.0, constant_value_bounds="fixed") * RBF(1.0, length_scale_bounds="fixed")
        else:
            self.kernel_ = clone(self.kernel)
        reason = self._unsupported_reason(self.kernel_, y, X)
        if reason is not None:       # before any work, and before the RandomState is touched: the base class does all of it
            return self._fit_on_host(X, y, reason)
        self._host_mode = False
        self._rng = check_random_state(self.random_state)

        # sklearn's own input validation (_gpr.py:254-262): same ValueErrors for NaN/inf, wrong rank, length
        # mismatch; sets n_features_in_
        X, y = validate_data(self, X, y, multi_output=True, y_numeric=True, ensure_2d=True, dtype="numeric")
        X = np.asarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        self._y_2d = y.ndim == 2
        if self._y_2d:
            y = y[:, 0]

        if self.normalize_y:  # _gpr.py:272-277 + preprocessing/_data.py:107-110
            self._y_train_mean = np.mean(y, axis=0)
            std = np.std(y, axis=0)
            self._y_train_std = 1.0 if std < 10 * np.finfo(np.float64).eps else std
            y = (y - self._y_train_mean) / self._y_train_std
        else:
            self._y_train_mean = np.zeros(())
            self._y_train_std = np.ones(())
        self.X_train_ = np.copy(X) if self.copy_X_train else X
        self.y_train_ = np.copy(y) if self.copy_X_train else y
        self.__dict__.pop("_L_cache", None)
        self.__dict__.pop("_alpha_cache", None)

        self._in_fit = True
        bare = _bare_length_scale_kernel(self.kernel_)
        theta0, bounds = _length_scale_theta(self.kernel_) if bare else (None, None)
        if self.optimizer is not None and (theta0.shape[0] if bare else self.kernel_.n_dims) > 0:  # _gpr.py:296-338 (L-BFGS-B on the host;
            # each objective evaluation runs on the device when _device_lml_ok)
            def obj_func(theta, eval_gradient=True):
                if eval_gradient:
                    lml, grad = self.log_marginal_likelihood(theta, eval_gradient=True, clone_kernel=False)
                    return -lml, -grad
                return -self.log_marginal_likelihood(theta, clone_kernel=False)

            # sklearn draws each restart's start right before running it (_gpr.py:325-334); the runs never touch the
            # RandomState, so drawing all starts first consumes the stream identically
            if not bare:
                theta0, bounds = self.kernel_.theta, self.kernel_.bounds
            starts = [theta0]
            if self.n_restarts_optimizer > 0:
                if not np.isfinite(bounds).all():
                    raise ValueError("Multiple optimizer restarts (n_restarts_optimizer>0) requires that all bounds are finite.")
                for _ in range(self.n_restarts_optimizer):
                    starts.append(self._rng.uniform(bounds[:, 0], bounds[:, 1]))
            # (each lockstep lane holds its own K, L, W: ~40 N^2 bytes per lane -> sequential beyond N = 16384)
            if (self.theta_lockstep and len(starts) > 1 and self.optimizer == "fmin_l_bfgs_b"
                    and self.X_train_.shape[0] <= 16384 and self._device_lml_ok(self.kernel_)):
                optima = self._theta_search_lockstep(starts, bounds)
            else:
                optima = [self._constrained_optimization(obj_func, start, bounds) for start in starts]
            lml_values = list(map(itemgetter(1), optima))
            theta_opt = optima[np.argmin(lml_values)][0]
            if bare:
                _set_length_scale_theta(self.kernel_, theta_opt)
                # (_check_bounds_params warns for the entries np.isclose finds on a bound: ask it only when there is one)
                if np.isclose(bounds, np.atleast_2d(theta_opt).T).any():
                    self.kernel_._check_bounds_params()
            else:
                self.kernel_.theta = theta_opt
                self.kernel_._check_bounds_params()
            self.log_marginal_likelihood_value_ = -np.min(lml_values)
        else:
            self.__dict__.pop("log_marginal_likelihood_value_", None)   # evaluated lazily (property below): _gpr.py:339-342
            self._lml_lazy = True

        self._in_fit = False
        kind, ls = describe_kernel(self.kernel_)
        if ls.shape[0] not in (1, self.n_features_in_):
            raise ValueError("Anisotropic kernel must have the same number of dimensions as data")
        self._kind, self._ls = kind, ls
        # _gpr.py:346-364 on the device (LinAlgError with sklearn's hint when K is not PD)
        self._device_fit_tail()
        return self

    def _theta_search_lockstep(self, starts, bounds):
        """[(theta_opt, -lml_opt)] of sklearn's `_constrained_optimization` from every start, the runs advanced together:
        per round ONE gpbo_lml_batch call evaluates the theta each live run is asking for (up to 8 side by side; the
        factorisations are latency-bound, so they overlap on the device).  Each run sees exactly the values it would see
        alone (every lane of the batch is bitwise gpbo_lml)."""
        from .lockstep import Lockstep

        eng = self._engine()
        X, y, noise = self._tx(self.X_train_), self.y_train_, float(self.alpha)
        n_dims = len(starts[0])

        # theta = log(length scale(s)) for the kernels _device_lml_ok admits (one free length-scale hyper-parameter,
        # kernels.py:Hyperparameter/theta); checked once here — starts[0] IS kernel_.theta — instead of cloning the kernel
        # for every evaluation (a clone costs ~0.1 ms of sklearn's get_params / signature machinery)
        kind, ls0 = describe_kernel(self.kernel_)
        if not np.all(np.abs(ls0 - np.exp(starts[0])) <= 1e-12 * np.abs(np.exp(starts[0]))):
            raise RuntimeError("theta does not map to the length scale as expected")    # pragma: no cover
        uploaded = [False]
        # how much work the search did (bench.py quotes the calls whose search ran >= 10 rounds separately)
        self.theta_search_rounds_ = 0      # lockstep rounds = gpbo_lml_batch calls on the critical path
        self.theta_search_evals_ = 0       # LML + gradient evaluations over all restarts

        batch_arrays = getattr(eng, "lml_batch_arrays", None)      # (GpEngine; engines without it: the list form)
        # a single device (not a group, whose lanes go to other devices through lml_batch_arrays): one call frame for the whole search
        frame = eng.lml_search_rounds(X, y, kind, n_dims, noise) if type(eng).__name__ == "GpEngine" else None

        def evaluate(thetas):
            self.theta_search_rounds_ += 1
            self.theta_search_evals_ += len(thetas)
            rows = np.empty((len(thetas), 1 + n_dims))
            scales = np.exp(np.asarray(thetas, dtype=np.float64))
            for lo in range(0, len(thetas), 8):
                if frame is not None:
                    vals, grads = frame(scales[lo:lo + 8])
                    rows[lo:lo + len(vals), 0] = vals
                    rows[lo:lo + len(vals), 1:] = grads
                elif batch_arrays is not None:
                    vals, grads = batch_arrays(X, y, kind, scales[lo:lo + 8], noise, True, uploaded[0])
                    rows[lo:lo + len(vals), 0] = vals
                    rows[lo:lo + len(vals), 1:] = grads
                else:
                    part = eng.lml_batch(X, y, kind, scales[lo:lo + 8], noise, eval_gradient=True, reuse_inputs=uploaded[0])
                    for j, (val, grad) in enumerate(part):
                        rows[lo + j, 0] = val
                        rows[lo + j, 1:] = grad
                uploaded[0] = True
            return rows

        from . import lbfgsb_lockstep

        if lbfgsb_lockstep.driver_available() and not np.any(bounds[:, 0] == bounds[:, 1]):
            # SciPy's setulb driven for all runs on this thread; what sklearn's _constrained_optimization does around
            # `minimize(obj_func, theta0, method="L-BFGS-B", jac=True, bounds=bounds)` (_gpr.py:656-668) follows it
            from sklearn.utils.optimize import _check_optimize_result

            def value_and_grad(thetas):
                rows = evaluate(thetas)
                return -rows[:, 0], -rows[:, 1:]

            optima = []
            # (the objective runs on the device: L-BFGS-B's own tiny BLAS calls stay off the host's thread pool)
            on_device = type(eng).__module__.startswith("bayesianoptimization_amd.")
            for res in lbfgsb_lockstep.minimize_many_with_grad(value_and_grad, starts, bounds, single_thread_blas=on_device):
                _check_optimize_result("lbfgs", res)
                optima.append((res.x, res.fun))
            return optima

        hub = Lockstep(evaluate, len(starts))
        results, errors = [None] * len(starts), [None] * len(starts)

        def run(idx, start):
            def objective(theta, eval_gradient=True):
                row = hub.ask(idx, np.asarray(theta, dtype=np.float64).reshape(1, -1))[0]
                return (-row[0], -row[1:]) if eval_gradient else -row[0]

            try:
                results[idx] = self._constrained_optimization(objective, start, bounds)
            except Lockstep.Abort:
                pass
            except BaseException as exc:
                errors[idx] = exc
            finally:
                hub.retire(idx)

        import threading

        threads = [threading.Thread(target=run, args=(i, np.array(s, dtype=np.float64)), daemon=True)
                   for i, s in enumerate(starts)]
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

    def _device_fit_tail(self):
        """gpbo_fit, or gpbo_fit_append when this fit extends the one the slot still holds (same theta, noise,
        precision; the previous inputs are a prefix of the new ones) — SURVEY.md §8 f4.  The reference refits from
        scratch on every maximize() iteration (bayesian_optimization.py:377-388); with optimizer=None that is the
        same factorisation plus one row, which is what the append computes."""
        eng = self._engine()
        X, y = self.X_train_, self.y_train_
        key = (self._kind, self._ls.tobytes(), float(self.alpha), self._precision_code())
        held = self.__dict__.get("_held")
        self._held = None
        # (copy_X_train=False keeps the caller's array: an in-place edit would go unnoticed, so no appends then)
        if (self.incremental and self.copy_X_train and held is not None and held["key"] == key and held["engine"] is eng
                and eng.fit_serial(self.slot) == held["serial"]):
            X0 = held["X"]
            n0 = X0.shape[0]
            if n0 <= X.shape[0] and X0.shape[1] == X.shape[1] and np.array_equal(X0, X[:n0]):
                serial = eng.fit_append(self._tx(X[n0:]) if X.shape[0] > n0 else np.empty((0, X.shape[1])), y,
                                        slot=self.slot)
                self._held = {"key": key, "engine": eng, "serial": serial, "X": X}
                return
        serial = eng.fit(self._tx(X), y, self._kind, self._ls, float(self.alpha), slot=self.slot,
                         precision=self._precision_code())
        self._held = {"key": key, "engine": eng, "serial": serial, "X": X}

    # -- predict -----------------------------------------------------------------------------------
    def predict(self, X, return_std=False, return_cov=False):
        if return_std and return_cov:
            raise RuntimeError("At most one of return_std or return_cov can be requested.")
        if not hasattr(self, "X_train_") or self._host_mode:
            # prior (unfitted) predictions are not on the hot path: sklearn's own code handles them; so it does for a model
            # outside the device path (_fit_on_host)
            return super().predict(X, return_std=return_std, return_cov=return_cov)
        X = np.asarray(validate_data(self, X, ensure_2d=True, dtype="numeric", reset=False), dtype=np.float64)  # _gpr.py:412
        self._ensure_resident()
        if return_cov:   # _gpr.py:458-469 on the device: V = W K*^T and V^T V as MFMA GEMMs; only M x M comes back
            if X.shape[0] > 16384:
                return super().predict(X, return_cov=True)      # (host path over the fetched L_: beyond the device buffers)
            mean, cov = self._engine().predict_cov(self._tx(X), slot=self.slot, y_mean=float(self._y_train_mean),
                                                   y_std=float(self._y_train_std))
            return mean, cov
        eng = self._engine()
        if return_std:
            eng.take_negative_variance_flag()          # clear: only this call's clips count
        mean, std = eng.predict(self._tx(X), slot=self.slot, y_mean=float(self._y_train_mean),
                                y_std=float(self._y_train_std))
        if return_std:
            # _gpr.py:479-485: sklearn warns when it clips NEGATIVE variances (a variance of exactly 0 is silent).  The
            # device clips inside its finalize kernel, which records that it did
            if eng.take_negative_variance_flag():
                warnings.warn("Predicted variances smaller than 0. Setting those variances to 0.", stacklevel=2)
            return mean, std
        return mean

    def _posterior_trusted(self, X):
        """(mean, std) for points this package generated itself (finite, right shape): predict(return_std=True)
        without sklearn's input validation and without the clipped-variance warning — the objective of the host
        optimisers calls this hundreds of times per suggest()."""
        if self._host_mode:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return GaussianProcessRegressor.predict(self, X, return_std=True)
        self._ensure_resident()
        return self._engine().predict(self._tx(X), slot=self.slot, y_mean=float(self._y_train_mean),
                                      y_std=float(self._y_train_std))

    def _posterior_grad_trusted(self, X):
        """(mean, std, d mean / d x, d std / d x) for a small batch of points this package generated itself
        (gpbo_predict_grad; identity input transform only — the chain rule through a host transform is not formed)."""
        if self.transform is not None or self._host_mode:
            raise NotImplementedError("input gradients need the identity input transform and a model on the device path")
        self._ensure_resident()
        return self._engine().predict_grad(np.ascontiguousarray(X, dtype=np.float64), slot=self.slot,
                                           y_mean=float(self._y_train_mean), y_std=float(self._y_train_std))

    # engine-resident posterior for the fused acquisition path ----------------------------------
    def posterior_resident(self):
        """Run the posterior kernel over the engine's resident candidates, keeping mu/sd on the device."""
        self._ensure_resident()
        self._engine().posterior(self.slot, float(self._y_train_mean), float(self._y_train_std), fetch=False)
