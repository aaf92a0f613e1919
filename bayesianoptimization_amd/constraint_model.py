"""HipConstraintModel — bayes_opt's ConstraintModel protocol with engine-backed GPs.

Mirrors bayes_opt/constraint.py:23-263 (`fit` :132-151, `predict` :153-221, `approx` :223-243,
`allowed` :245-263).  Constraint j lives in engine slot j+1; `predict` on a host batch goes through
HipGPR.predict, while the fused acquisition path (fused_acquisition.py) reads `_model`, `_lb`, `_ub`
and keeps every posterior on the device.
"""
from __future__ import annotations

import numpy as np
from scipy.special import ndtr
from sklearn.gaussian_process.kernels import Matern

from .gpr import HipGPR


def _cdf(bound, mean, std):
    """scipy.stats.norm(loc=mean, scale=std).cdf(bound): NaN where std <= 0."""
    with np.errstate(divide="ignore", invalid="ignore"):
        out = ndtr((bound - mean) / std)
    return np.where(std > 0, out, np.nan)


class HipConstraintModel:
    def __init__(self, fun, lb, ub, transform=None, random_state=None, engine=None, first_slot=1):
        self.fun = fun
        self._lb = np.atleast_1d(lb).astype(np.float64)
        self._ub = np.atleast_1d(ub).astype(np.float64)
        if np.any(self._lb >= self._ub):
            raise ValueError("Lower bounds must be less than upper bounds.")
        self._model = [
            HipGPR(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5,
                   random_state=random_state, transform=transform, engine=engine, slot=first_slot + j)
            for j in range(len(self._lb))
        ]

    @property
    def lb(self):
        return self._lb

    @property
    def ub(self):
        return self._ub

    @property
    def model(self):
        return self._model

    def fit(self, X, Y):
        if len(self._model) == 1:
            self._model[0].fit(X, Y)
        else:
            for i, gp in enumerate(self._model):
                gp.fit(X, Y[:, i])

    def predict(self, X):
        X_shape = X.shape
        X = X.reshape((-1, self._model[0].n_features_in_))
        result = None
        for j, gp in enumerate(self._model):
            y_mean, y_std = gp.predict(X, return_std=True)
            p_lower = _cdf(self._lb[j], y_mean, y_std) if self._lb[j] != -np.inf else np.array([0])
            p_upper = _cdf(self._ub[j], y_mean, y_std) if self._ub[j] != np.inf else np.array([1])
            p = p_upper - p_lower
            result = p if result is None else result * p
        return result.reshape(X_shape[:-1])

    def approx(self, X):
        X_shape = X.shape
        X = X.reshape((-1, self._model[0].n_features_in_))
        if len(self._model) == 1:
            return self._model[0].predict(X).reshape(X_shape[:-1])
        result = np.column_stack([gp.predict(X) for gp in self._model])
        return result.reshape(X_shape[:-1] + (len(self._lb),))

    def allowed(self, constraint_values):
        if self._lb.size == 1:
            return np.less_equal(self._lb, constraint_values) & np.less_equal(constraint_values, self._ub)
        return np.all(constraint_values <= self._ub, axis=-1) & np.all(constraint_values >= self._lb, axis=-1)
