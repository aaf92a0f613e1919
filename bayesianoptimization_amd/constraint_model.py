"""HipConstraintModel — the protocol of bayes_opt's ConstraintModel with engine-backed GPs.

What the rest of bayes_opt expects from `space.constraint` (bayes_opt/constraint.py:23-263): `fit(X, Y)`
(:132-151), `predict(X)` = probability that every constraint holds (:153-221), `approx(X)` = the GPs' means
(:223-243), `allowed(values)` (:245-263), the bounds `lb`/`ub` and the per-constraint estimators in `_model`.
Constraint j lives in engine slot j + 1.  `predict` on a host batch goes through `HipGPR.predict`; the fused
acquisition path (fused_acquisition.py) reads `_model`, `_lb`, `_ub` directly and keeps every posterior on the
device.  Written against that protocol, not copied from the reference.
"""
from __future__ import annotations

import numpy as np
from scipy.special import ndtr
from sklearn.gaussian_process.kernels import Matern

from .gpr import HipGPR


def _interval_probability(lo, hi, mean, std):
    """P(lo <= N(mean, std^2) <= hi) with SciPy's conventions: an infinite bound contributes exactly 0 / 1
    (constraint.py:202-207 short-circuits it), and a non-positive std gives NaN (frozen-distribution check)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        upper = 1.0 if hi == np.inf else np.where(std > 0, ndtr((hi - mean) / std), np.nan)
        lower = 0.0 if lo == -np.inf else np.where(std > 0, ndtr((lo - mean) / std), np.nan)
    return upper - lower


class HipConstraintModel:
    def __init__(self, fun, lb, ub, transform=None, random_state=None, engine=None, first_slot=1):
        self.fun = fun
        self._lb = np.atleast_1d(lb).astype(np.float64)
        self._ub = np.atleast_1d(ub).astype(np.float64)
        if np.any(self._lb >= self._ub):
            raise ValueError("Lower bounds must be less than upper bounds.")
        # same estimator configuration as the reference gives its constraint GPs (constraint.py:72-81)
        gp_config = dict(alpha=1e-6, normalize_y=True, n_restarts_optimizer=5, random_state=random_state)
        self._model = [HipGPR(kernel=Matern(nu=2.5), transform=transform, engine=engine, slot=first_slot + j,
                              **gp_config) for j in range(self._lb.size)]

    lb = property(lambda self: self._lb)
    ub = property(lambda self: self._ub)
    model = property(lambda self: self._model)

    def _columns(self, Y):
        Y = np.asarray(Y)
        return [Y] if len(self._model) == 1 else [Y[:, j] for j in range(len(self._model))]

    def fit(self, X, Y):
        for gp, column in zip(self._model, self._columns(Y)):
            gp.fit(X, column)

    def predict(self, X):
        lead_shape = X.shape[:-1]
        pts = X.reshape((-1, self._model[0].n_features_in_))
        prob = None
        for gp, lo, hi in zip(self._model, self._lb, self._ub):
            mean, std = gp.predict(pts, return_std=True)
            p = np.broadcast_to(_interval_probability(lo, hi, mean, std), mean.shape)
            prob = p if prob is None else prob * p
        return np.asarray(prob).reshape(lead_shape)

    def _predict_trusted(self, pts):
        """predict() for an (M, d) batch of points generated inside this package (see HipGPR._posterior_trusted)."""
        prob = None
        for gp, lo, hi in zip(self._model, self._lb, self._ub):
            mean, std = gp._posterior_trusted(pts)
            p = np.broadcast_to(_interval_probability(lo, hi, mean, std), mean.shape)
            prob = p if prob is None else prob * p
        return np.asarray(prob)

    def approx(self, X):
        lead_shape = X.shape[:-1]
        pts = X.reshape((-1, self._model[0].n_features_in_))
        means = [gp.predict(pts) for gp in self._model]
        if len(means) == 1:
            return means[0].reshape(lead_shape)
        return np.column_stack(means).reshape(lead_shape + (len(means),))

    def allowed(self, constraint_values):
        values = np.asarray(constraint_values)
        inside = (self._lb <= values) & (values <= self._ub)
        return inside if self._lb.size == 1 else np.all(inside, axis=-1)
