"""Deterministic synthetic inputs for the suggest() hot path.

The same generators feed bench.py, the parity tests and oracle/gen_golden.py, so a fixture generated
from the reference in the build container and a run on the GPU box see bit-identical inputs.
NumPy only: nothing here touches the device or the oracle.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

RBF = 0
MATERN25 = 1
UCB = 0
EI = 1
POI = 2

KERNEL_NAMES = {RBF: "rbf", MATERN25: "matern25"}
ACQ_NAMES = {UCB: "ucb", EI: "ei", POI: "poi"}


@dataclass(frozen=True)
class Workload:
    """One row of BASELINE.json `configs` (or a reduced parity case)."""

    name: str
    d: int
    N: int
    M: int
    kernel: int = MATERN25
    acq: int = UCB
    acq_param: float = 2.576  # kappa (UCB) or xi (EI/POI)
    length_scale: float | None = None  # None -> take theta from the fitted reference GP (golden file)
    noise: float = 1e-6  # GaussianProcessRegressor(alpha=1e-6), bayesian_optimization.py:124-130
    constrained: bool = False
    constraint_ub: float = 0.5
    constraint_length_scale: float | None = None
    dtype: str = "f64"
    readme_function: bool = False
    y_noise: float = 0.1
    bounds: tuple = field(default=())

    def pbounds(self) -> dict:
        if self.bounds:
            return {k: (lo, hi) for k, lo, hi in self.bounds}
        return {f"x{i}": (0.0, 1.0) for i in range(self.d)}

    def bounds_array(self) -> np.ndarray:
        return np.array([[lo, hi] for lo, hi in self.pbounds().values()], dtype=np.float64)


# BASELINE.json configs.  Length scales are FIXED: sklearn's theta search on this noisy generator
# (0.1*randn against alpha=1e-6) runs to the lower bound 1e-5 (K = I, a degenerate posterior; probed
# at d=8/16/32), and at N >= 4096 a search costs minutes on the CPU reference.  The
# values give cond(K) ~ 1e4..3e5 and non-trivial mu/sigma.  F1 is the fitted-theta case: its length
# scale is read from the golden file written by the reference's own L-BFGS-B search.
C1 = Workload("C1", d=2, N=25, M=1024, kernel=RBF, acq=UCB, acq_param=2.576, length_scale=None, y_noise=0.0,
              readme_function=True, bounds=(("x", 2.0, 4.0), ("y", -3.0, 3.0)))
C2 = Workload("C2", d=8, N=512, M=65536, kernel=MATERN25, acq=EI, acq_param=0.01, length_scale=1.0)
C3 = Workload("C3", d=16, N=4096, M=1 << 20, kernel=MATERN25, acq=UCB, acq_param=2.576, length_scale=1.5)
C4 = Workload("C4", d=16, N=4096, M=1 << 23, kernel=MATERN25, acq=EI, acq_param=0.01, length_scale=1.5)
C5 = Workload("C5", d=32, N=8192, M=1 << 21, kernel=MATERN25, acq=EI, acq_param=0.01, length_scale=2.0,
              constrained=True, constraint_length_scale=2.0, dtype="f32")
# Reduced parity cases (oracle finishes in seconds).
C5S = Workload("C5S", d=4, N=256, M=8192, kernel=MATERN25, acq=EI, acq_param=0.01, length_scale=0.5,
               constrained=True, constraint_length_scale=0.7)
P1 = Workload("P1", d=3, N=200, M=4096, kernel=MATERN25, acq=POI, acq_param=0.01, length_scale=0.4)
P2 = Workload("P2", d=5, N=300, M=4096, kernel=RBF, acq=EI, acq_param=0.01, length_scale=0.6)
F1 = Workload("F1", d=3, N=60, M=4096, kernel=MATERN25, acq=UCB, acq_param=2.576, length_scale=None,
              y_noise=0.0)

# Plateau cases (tie order under exact ties, which np.argsort leaves unspecified): T1 — POI with xi = 100 underflows to
# exactly 0 for every candidate; T2 — EI with xi = 5.36 underflows to exactly 0 for all but 9 of the 512 candidates
# (z <= -39.5 for the rest; the 9 have z >= -34.8, i.e. normal numbers), so argsort[:16] crosses into the plateau.
T1 = Workload("T1", d=2, N=60, M=4096, kernel=MATERN25, acq=POI, acq_param=100.0, length_scale=0.4)
T2 = Workload("T2", d=2, N=60, M=512, kernel=MATERN25, acq=EI, acq_param=5.36, length_scale=0.4)

ALL = {w.name: w for w in (C1, C2, C3, C4, C5, C5S, P1, P2, F1, T1, T2)}


def readme_black_box(x, y):
    """The README example target (reference README.md:66-85)."""
    return -(x**2) - (y - 1) ** 2 + 1


def make_observations(w: Workload):
    """(X (N,d) f64 C-order, y (N,), c (N,) or None) — SURVEY.md §8d 'Synthetic inputs'."""
    if w.readme_function:
        rng = np.random.RandomState(1)
        b = w.bounds_array()
        X = np.empty((w.N, w.d))
        for j in range(w.d):
            X[:, j] = rng.uniform(b[j, 0], b[j, 1], w.N)
        y = readme_black_box(X[:, 0], X[:, 1])
        return X, y, None
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(w.N, w.d))
    y = np.sin(3.0 * X.sum(1)) + w.y_noise * rng.randn(w.N)
    c = np.cos(2.0 * X.sum(1)) if w.constrained else None
    return X, y, c


def make_candidates(bounds: np.ndarray, M: int, random_state) -> np.ndarray:
    """Candidate matrix (M,d) f64 C-order drawn exactly as the reference does for an all-float space.

    Follows TargetSpace.random_sample (reference bayes_opt/target_space.py:593-600) with
    FloatParameter.random_sample (bayes_opt/parameter.py:86-87): one `uniform(lo, hi, M)` draw per
    parameter, in key order, written into that parameter's column.
    """
    if not isinstance(random_state, np.random.RandomState):
        random_state = np.random.RandomState(random_state)
    bounds = np.asarray(bounds, dtype=np.float64)
    n = max(1, int(M))
    data = np.empty((n, bounds.shape[0]))
    for j in range(bounds.shape[0]):
        data[:, j] = random_state.uniform(bounds[j, 0], bounds[j, 1], n)
    return data


def normalize_targets(y):
    """(y_norm, mean, std) as GaussianProcessRegressor.fit(normalize_y=True) forms them (_gpr.py:272-277; a zero std
    becomes 1, preprocessing/_data.py:107-110)."""
    y = np.asarray(y, dtype=np.float64)
    mean, std = float(np.mean(y)), float(np.std(y))
    if std < 10 * np.finfo(np.float64).eps:
        std = 1.0
    return (y - mean) / std, mean, std


def flops_per_candidate(N: int, d: int, n_gp: int = 1) -> float:
    """F_cand = N^2 + (3d + 12) N per GP."""
    return float(n_gp) * (float(N) * N + (3 * d + 12) * float(N))


def feasible_y_max(w: Workload, y, c):
    """y_max as TargetSpace._target_max (target_space.py:605-622): max over feasible, in-bounds points."""
    if c is None:
        return float(np.max(y))
    ok = c <= w.constraint_ub
    return float(np.max(y[ok])) if ok.any() else None
