"""FloatSpace — the slice of bayes_opt's TargetSpace protocol that the suggest() hot path reads.

The reference's own `TargetSpace` (bayes_opt/target_space.py) is used unchanged when bayes_opt is
installed; this stand-in exists so that benches and GPU tests can drive the same acquisition code on
a box without the reference.  It covers all-float parameter spaces only (every BASELINE.json config):
observation store (`params`, `target`, `register`), `bounds`, `random_sample` (same RandomState
stream as target_space.py:565-603 / parameter.py:86-87), `_target_max`/`mask` (target_space.py:387-410,
605-622), `constraint` and `_constraint_values`.
"""
from __future__ import annotations

import numpy as np

from .workloads import make_candidates


def ensure_rng(random_state=None) -> np.random.RandomState:
    """bayes_opt/util.py:8-30."""
    if random_state is None:
        return np.random.RandomState()
    if isinstance(random_state, int):
        return np.random.RandomState(random_state)
    if isinstance(random_state, np.random.RandomState):
        return random_state
    raise TypeError("random_state should be an instance of np.random.RandomState, an int, or None.")


class FloatSpace:
    def __init__(self, pbounds: dict, constraint=None):
        self._keys = list(pbounds.keys())
        self._bounds = np.array([[float(lo), float(hi)] for lo, hi in pbounds.values()], dtype=np.float64)
        self._dim = len(self._keys)
        self._params = np.empty((0, self._dim))
        self._target = np.empty((0,))
        self._constraint = constraint
        self._constraint_values = np.empty((0,)) if constraint is not None else None

    # -- protocol -----------------------------------------------------------------------------
    def __len__(self):
        return len(self._target)

    @property
    def empty(self):
        return len(self) == 0

    @property
    def params(self):
        return self._params

    @property
    def target(self):
        return self._target

    @property
    def dim(self):
        return self._dim

    @property
    def keys(self):
        return self._keys

    @property
    def bounds(self):
        return self._bounds

    @property
    def constraint(self):
        return self._constraint

    @property
    def continuous_dimensions(self):
        return np.ones(self._dim, dtype=bool)

    def kernel_transform(self, value):
        return np.atleast_2d(value)  # FloatParameter.kernel_transform is the identity (parameter.py:222-234)

    def register_bulk(self, X, y, constraint_values=None):
        self._params = np.ascontiguousarray(X, dtype=np.float64)
        self._target = np.ascontiguousarray(y, dtype=np.float64)
        if self._constraint is not None:
            self._constraint_values = np.ascontiguousarray(constraint_values, dtype=np.float64)

    def register(self, params, target, constraint_value=None):
        x = np.asarray(params, dtype=np.float64).ravel()
        self._params = np.concatenate([self._params, x.reshape(1, -1)])
        self._target = np.concatenate([self._target, [target]])
        if self._constraint is not None:
            self._constraint_values = np.concatenate([self._constraint_values, [constraint_value]])

    def random_sample(self, n_samples: int = 0, random_state=None):
        rng = ensure_rng(random_state)
        data = make_candidates(self._bounds, max(1, n_samples), rng)
        return data.ravel() if n_samples == 0 else data

    @property
    def mask(self):
        mask = np.ones_like(self._target, dtype=bool)
        if self._constraint is not None:
            mask &= self._constraint.allowed(self._constraint_values)
        within = np.all((self._bounds[:, 0] <= self._params) & (self._params <= self._bounds[:, 1]), axis=1)
        return mask & within

    def _target_max(self):
        if len(self._target) == 0:
            return None
        sel = self._target[self.mask]
        return None if len(sel) == 0 else sel.max()

    def array_to_params(self, x):
        return dict(zip(self._keys, np.asarray(x).ravel()))
