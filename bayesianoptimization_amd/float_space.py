"""FloatSpace / MixedSpace — the slice of bayes_opt's TargetSpace protocol that the suggest() hot path reads.

The reference's own `TargetSpace` (bayes_opt/target_space.py) is used unchanged when bayes_opt is
installed; this stand-in exists so that benches and GPU tests can drive the same acquisition code on
a box without the reference.  It covers all-float parameter spaces only (every BASELINE.json config):
observation store (`params`, `target`, `register`), `bounds`, `random_sample` (same RandomState
stream as target_space.py:565-603 / parameter.py:86-87), `_target_max`/`mask` (target_space.py:387-410,
605-622), `constraint` and `_constraint_values`.  `MixedSpace` adds the reference's integer and categorical parameters
(bayes_opt/parameter.py:237-449: `randint` sampling, `np.round` / one-hot `kernel_transform`, the per-parameter column
masks of target_space.py:281-301) with the same names and attributes the fused acquisition inspects — the GPU box has no
bayes_opt, and tests/test_host_logic.py checks this stand-in against the real TargetSpace bit for bit.
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


# ---- parameters of a mixed space: the reference's semantics restated (bayes_opt/parameter.py) -------------------------------
class FloatParameter:
    """parameter.py:173-234: uniform(lo, hi); identity kernel transform; one column."""

    is_continuous = True
    dim = 1

    def __init__(self, name, bounds):
        self.name = name
        self.bounds = np.array([float(bounds[0]), float(bounds[1])])

    def random_sample(self, n_samples, random_state):
        return ensure_rng(random_state).uniform(self.bounds[0], self.bounds[1], n_samples)       # parameter.py:86-87

    def kernel_transform(self, value):
        return value


class IntParameter:
    """parameter.py:237-325: randint(lo, hi + 1).astype(float); kernel transform np.round; one column."""

    is_continuous = False
    dim = 1

    def __init__(self, name, bounds):
        self.name = name
        self.bounds = np.array([int(bounds[0]), int(bounds[1])])

    def random_sample(self, n_samples, random_state):
        return ensure_rng(random_state).randint(self.bounds[0], self.bounds[1] + 1, n_samples).astype(float)   # :280-284

    def kernel_transform(self, value):
        return np.round(value)                                                                       # :308-320


class CategoricalParameter:
    """parameter.py:328-454: randint(0, n_categories) -> one-hot rows; bounds [0, 1] per category column; the kernel
    transform is the reference's own, batch behaviour included (`res[:, argmax(value, axis=1)] = 1`, :434-449)."""

    is_continuous = False

    def __init__(self, name, categories):
        if len(categories) != len(set(categories)):
            raise ValueError("Categories must be unique.")
        if len(categories) < 2:
            raise ValueError("At least two categories are required.")
        self.name = name
        self.categories = list(categories)
        self.dim = len(self.categories)
        self.bounds = np.vstack((np.zeros(self.dim), np.ones(self.dim))).T

    def random_sample(self, n_samples, random_state):
        res = ensure_rng(random_state).randint(0, len(self.categories), n_samples)                   # :372-377
        one_hot = np.zeros((n_samples, len(self.categories)))
        one_hot[np.arange(n_samples), res] = 1
        return one_hot.astype(float)

    def kernel_transform(self, value):
        value = np.atleast_2d(value)
        res = np.zeros(value.shape)
        res[:, np.argmax(value, axis=1)] = 1
        return res


class MixedSpace(FloatSpace):
    """FloatSpace with the reference's three parameter kinds.  `pbounds` values as TargetSpace.make_params reads them
    (target_space.py:237-279): (lo, hi) -> float, (lo, hi, int) -> integer, any other sequence -> categories."""

    def __init__(self, pbounds: dict, constraint=None):
        self._keys = list(pbounds.keys())
        self._params_config = {}
        for key, pb in pbounds.items():
            if len(pb) == 2 and all(isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool) for v in pb):
                self._params_config[key] = FloatParameter(key, pb)
            elif len(pb) == 3 and pb[-1] is float:
                self._params_config[key] = FloatParameter(key, pb[:2])
            elif len(pb) == 3 and pb[-1] is int:
                self._params_config[key] = IntParameter(key, pb[:2])
            else:
                self._params_config[key] = CategoricalParameter(key, pb)
        self._dim = sum(p.dim for p in self._params_config.values())
        self._masks, pos = {}, 0
        for key in self._keys:                                                                      # target_space.py:281-301
            m = np.zeros(self._dim, dtype=bool)
            m[pos:pos + self._params_config[key].dim] = True
            self._masks[key] = m
            pos += self._params_config[key].dim
        self._bounds = np.empty((self._dim, 2))
        for key in self._keys:
            self._bounds[self._masks[key]] = np.atleast_2d(self._params_config[key].bounds).astype(float)
        self._params = np.empty((0, self._dim))
        self._target = np.empty((0,))
        self._constraint = constraint
        self._constraint_values = np.empty((0,)) if constraint is not None else None

    @property
    def masks(self):
        return self._masks

    @property
    def continuous_dimensions(self):
        out = np.zeros(self._dim, dtype=bool)
        for key in self._keys:
            out[self._masks[key]] = self._params_config[key].is_continuous
        return out

    def kernel_transform(self, value):                                                             # target_space.py:340-347
        value = np.atleast_2d(value)
        return np.hstack([self._params_config[p].kernel_transform(value[:, self._masks[p]]) for p in self._keys])

    def random_sample(self, n_samples: int = 0, random_state=None):                                # target_space.py:565-603
        rng = ensure_rng(random_state)
        n = max(1, n_samples)
        data = np.empty((n, self._dim))
        for key in self._keys:
            smpl = self._params_config[key].random_sample(n, rng)
            data[:, self._masks[key]] = smpl.reshape(n, self._params_config[key].dim)
        return data.ravel() if n_samples == 0 else data
