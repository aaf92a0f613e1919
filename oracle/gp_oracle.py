"""CPU ORACLE — TEST INFRASTRUCTURE ONLY, never the product path.

A plain NumPy/SciPy restatement of the arithmetic the reference delegates to scikit-learn/SciPy on
its suggest() hot path, at a FIXED kernel hyper-parameter theta.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the product
package `bayesianoptimization_amd` never does (it fails loudly when the HIP library is missing).

Parity status: PINNED.  The reference has no golden vectors for this path, so the
oracle is pinned against outputs of the reference itself run in the build container:
`oracle/gen_golden.py` drives bayes_opt 3.3.0 -> scikit-learn 1.7.2 / SciPy 1.15.3 / NumPy 2.2.6 and
commits the results under `tests/golden/`; `tests/test_oracle_golden.py` checks every function here
against them (and directly against sklearn when it is importable).

What each function follows (SK = site-packages/sklearn, SP = site-packages/scipy, paths in the
reference are relative to /root/reference):
  kernel_matrix      SK/gaussian_process/kernels.py:1711-1738 (Matern nu=2.5: X/length_scale, cdist
                     euclidean, K = sqrt(5)*d, (1 + K + K**2/3) * exp(-K)), :1556-1565 (RBF: sqeuclidean,
                     exp(-0.5 d2)); called from bayes_opt/parameter.py:484-487 (WrappedKernel).
  normalize_targets  SK/gaussian_process/_gpr.py:272-277, SK/preprocessing/_data.py:107-110.
  fit_fixed_theta    SK/gaussian_process/_gpr.py:346-364 (K[diag] += alpha; cholesky lower; cho_solve).
  predict            SK/gaussian_process/_gpr.py:443-447, 454-456, 474-494.
  base_acq_*         bayes_opt/acquisition.py:485 (UCB), :660-661 (POI), :847-849 (EI).
  constraint_prob    bayes_opt/constraint.py:199-221.
  neg_acquisition    bayes_opt/acquisition.py:198-217 (the _get_acq closures).
  arg_best           bayes_opt/acquisition.py:312-317 (argmin / min / argsort[:k]).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.linalg import cho_solve, cholesky, solve_triangular
from scipy.spatial.distance import cdist
from scipy.special import ndtr

RBF = 0
MATERN25 = 1
UCB = 0
EI = 1
POI = 2

_SQRT5 = np.sqrt(5.0)
_SQRT_2PI = np.sqrt(2.0 * np.pi)


def kernel_matrix(kind: int, Xa: np.ndarray, Xb: np.ndarray | None, length_scale) -> np.ndarray:
    """k(Xa, Xb) (or k(Xa, Xa) with an exact unit diagonal when Xb is None)."""
    ls = np.asarray(length_scale, dtype=np.float64)
    A = np.asarray(Xa, dtype=np.float64) / ls
    B = A if Xb is None else np.asarray(Xb, dtype=np.float64) / ls
    if kind == MATERN25:
        d = cdist(A, B, metric="euclidean")
        k = d * _SQRT5
        out = (1.0 + k + k**2 / 3.0) * np.exp(-k)
    elif kind == RBF:
        d2 = cdist(A, B, metric="sqeuclidean")
        out = np.exp(-0.5 * d2)
    else:
        raise ValueError(f"unsupported kernel kind {kind}")
    if Xb is None:
        np.fill_diagonal(out, 1.0)  # kernels.py:1735-1738 (pdist+squareform+fill_diagonal)
    return out


def normalize_targets(y: np.ndarray, normalize_y: bool = True):
    y = np.asarray(y, dtype=np.float64)
    if not normalize_y:
        return y.copy(), 0.0, 1.0
    mean = float(np.mean(y, axis=0))
    std = float(np.std(y, axis=0))
    if std < 10 * np.finfo(np.float64).eps:  # _handle_zeros_in_scale, preprocessing/_data.py:107-110
        std = 1.0
    return (y - mean) / std, mean, std


@dataclass
class GPState:
    kind: int
    length_scale: np.ndarray
    noise: float
    X: np.ndarray
    L: np.ndarray
    alpha: np.ndarray
    y_mean: float
    y_std: float


def fit_fixed_theta(kind, X, y, length_scale, noise=1e-6, normalize_y=True) -> GPState:
    X = np.ascontiguousarray(X, dtype=np.float64)
    yn, mean, std = normalize_targets(y, normalize_y)
    K = kernel_matrix(kind, X, None, length_scale)
    K[np.diag_indices_from(K)] += noise
    L = cholesky(K, lower=True, check_finite=False)
    alpha = cho_solve((L, True), yn, check_finite=False)
    return GPState(kind, np.atleast_1d(np.asarray(length_scale, dtype=np.float64)), float(noise), X, L, alpha,
                   mean, std)


def predict(gp: GPState, Xc: np.ndarray):
    """Posterior mean and standard deviation, as GaussianProcessRegressor.predict(return_std=True)."""
    Xc = np.asarray(Xc, dtype=np.float64).reshape(-1, gp.X.shape[1])
    Kt = kernel_matrix(gp.kind, Xc, gp.X, gp.length_scale)
    mean = Kt @ gp.alpha
    mean = gp.y_std * mean + gp.y_mean
    V = solve_triangular(gp.L, Kt.T, lower=True, check_finite=False)
    var = np.ones(Xc.shape[0]) - np.einsum("ij,ji->i", V.T, V)
    var[var < 0] = 0.0
    var = var * gp.y_std**2
    return mean, np.sqrt(var)


def negative_variances(gp: GPState, Xc: np.ndarray) -> int:
    """How many predicted variances sklearn would clip (and warn about): y_var < 0 before the clip, _gpr.py:479-485."""
    Xc = np.asarray(Xc, dtype=np.float64).reshape(-1, gp.X.shape[1])
    Kt = kernel_matrix(gp.kind, Xc, gp.X, gp.length_scale)
    V = solve_triangular(gp.L, Kt.T, lower=True, check_finite=False)
    return int(np.count_nonzero(np.ones(Xc.shape[0]) - np.einsum("ij,ji->i", V.T, V) < 0))


def predict_cov(gp: GPState, Xc: np.ndarray):
    """(mean, covariance) as GaussianProcessRegressor.predict(return_cov=True) (_gpr.py:443-469)."""
    Xc = np.asarray(Xc, dtype=np.float64).reshape(-1, gp.X.shape[1])
    Kt = kernel_matrix(gp.kind, Xc, gp.X, gp.length_scale)
    mean = gp.y_std * (Kt @ gp.alpha) + gp.y_mean
    V = solve_triangular(gp.L, Kt.T, lower=True, check_finite=False)
    cov = kernel_matrix(gp.kind, Xc, None, gp.length_scale) - V.T @ V
    return mean, cov * gp.y_std**2


def predict_grad(gp: GPState, Xc: np.ndarray):
    """(mean, std, d mean / d x, d std / d x): the analytic input gradient of `predict` — the checker of
    gpbo_predict_grad.  sklearn has no such routine; the formulas differentiate _gpr.py:443-494 with the kernels of
    kernels.py:1722-1724 / 1559-1560:  dk/dx_t = f(r) (x_t - X_kt) / l_t^2 with f = -(5/3)(1 + sqrt5 r) exp(-sqrt5 r)
    (Matern-2.5) or -k (RBF);  d var_n / d x = -2 (K^-1 k*)^T dk*/dx."""
    Xc = np.asarray(Xc, dtype=np.float64).reshape(-1, gp.X.shape[1])
    ls = np.broadcast_to(gp.length_scale, (gp.X.shape[1],))
    mean, std = predict(gp, Xc)
    Kt = kernel_matrix(gp.kind, Xc, gp.X, gp.length_scale)                    # (M, N)
    diff = (Xc[:, None, :] - gp.X[None, :, :]) / ls**2                        # (M, N, d)
    if gp.kind == MATERN25:
        r = cdist(Xc / ls, gp.X / ls, metric="euclidean")
        f = -(5.0 / 3.0) * (1.0 + _SQRT5 * r) * np.exp(-_SQRT5 * r)
    else:
        f = -Kt
    dK = f[:, :, None] * diff                                                   # (M, N, d)
    dmean = gp.y_std * np.einsum("mnd,n->md", dK, gp.alpha)
    u = cho_solve((gp.L, True), Kt.T, check_finite=False).T                     # (M, N) = K^-1 k*
    dvar_n = -2.0 * np.einsum("mnd,mn->md", dK, u)
    sd_n = std / gp.y_std
    with np.errstate(divide="ignore", invalid="ignore"):
        dstd = np.where(sd_n[:, None] > 0, gp.y_std * dvar_n / (2.0 * sd_n[:, None]), 0.0)
    return mean, std, dmean, dstd


def norm_cdf(x):
    return ndtr(x)


def norm_pdf(x):
    return np.exp(-(x**2) / 2.0) / _SQRT_2PI  # scipy/stats/_continuous_distns.py:360-362


def base_acq_ucb(mean, std, kappa):
    return mean + kappa * std


def base_acq_ei(mean, std, y_max, xi):
    with np.errstate(divide="ignore", invalid="ignore"):
        a = mean - y_max - xi
        z = a / std
        return a * norm_cdf(z) + std * norm_pdf(z)


def base_acq_poi(mean, std, y_max, xi):
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (mean - y_max - xi) / std
        return norm_cdf(z)


def base_acq(kind, mean, std, param, y_max=0.0):
    if kind == UCB:
        return base_acq_ucb(mean, std, param)
    if kind == EI:
        return base_acq_ei(mean, std, y_max, param)
    if kind == POI:
        return base_acq_poi(mean, std, y_max, param)
    raise ValueError(kind)


def _cdf_loc_scale(bound, mean, std):
    """scipy.stats.norm(loc, scale).cdf(bound): NaN where scale <= 0 (frozen-distribution arg check)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        out = ndtr((bound - mean) / std)
    return np.where(std > 0, out, np.nan)


def constraint_prob(gps, lb, ub, Xc):
    """Probability that every constraint j satisfies lb[j] <= c_j(x) <= ub[j] (independent GPs)."""
    lb = np.atleast_1d(np.asarray(lb, dtype=np.float64))
    ub = np.atleast_1d(np.asarray(ub, dtype=np.float64))
    result = None
    for j, gp in enumerate(gps):
        mean, std = predict(gp, Xc)
        p_lower = _cdf_loc_scale(lb[j], mean, std) if lb[j] != -np.inf else np.array([0.0])
        p_upper = _cdf_loc_scale(ub[j], mean, std) if ub[j] != np.inf else np.array([1.0])
        p = p_upper - p_lower
        result = p if result is None else result * p
    return np.broadcast_to(result, (np.asarray(Xc).reshape(-1, gps[0].X.shape[1]).shape[0],)).copy()


def neg_acquisition(gp, Xc, kind, param, y_max=0.0, constraint=None):
    """-1 * base_acq(mean, std) [* p_constraints]: the function the reference minimises."""
    mean, std = predict(gp, Xc)
    vals = -1 * base_acq(kind, mean, std, param, y_max)
    if constraint is not None:
        gps, lb, ub = constraint
        vals = vals * constraint_prob(gps, lb, ub, Xc)
    return vals


def arg_best(ys: np.ndarray, k: int = 0):
    """(argmin, min, argsort[:k]) with NumPy semantics: first NaN wins argmin; NaNs sort last."""
    ys = np.asarray(ys)
    idx = int(ys.argmin())
    seeds = np.argsort(ys)[:k] if k else np.empty(0, dtype=np.int64)
    return idx, float(ys.min()), seeds


def log_marginal_likelihood(kind, X, y_norm, length_scale, noise=1e-6, eval_gradient=True):
    """LML and d LML / d log(length_scale) at fixed theta, following sklearn _gpr.py:575-652 and the kernel
    gradients kernels.py:1764-1766 (Matern nu=2.5) / :1567-1582 (RBF).  Non-PD K -> (-inf, zeros)."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y_norm, dtype=np.float64)
    ls = np.atleast_1d(np.asarray(length_scale, dtype=np.float64))
    K = kernel_matrix(kind, X, None, ls)
    K[np.diag_indices_from(K)] += noise
    try:
        L = cholesky(K, lower=True, check_finite=False)
    except np.linalg.LinAlgError:
        return (-np.inf, np.zeros(ls.shape[0])) if eval_gradient else -np.inf
    alpha = cho_solve((L, True), y, check_finite=False)
    lml = -0.5 * float(y @ alpha) - np.log(np.diag(L)).sum() - K.shape[0] / 2 * np.log(2 * np.pi)
    if not eval_gradient:
        return lml
    Xs = X / ls
    D = (Xs[:, None, :] - Xs[None, :, :]) ** 2           # (N, N, d) squared scaled differences
    d2 = D.sum(-1)
    if kind == MATERN25:
        tmp = np.sqrt(5 * d2)
        g = 5.0 / 3.0 * (tmp + 1) * np.exp(-tmp)
    else:
        g = np.exp(-0.5 * d2)
    K_inv = cho_solve((L, True), np.eye(K.shape[0]), check_finite=False)
    inner = np.outer(alpha, alpha) - K_inv
    if ls.shape[0] == 1:
        grad = np.array([0.5 * np.sum(inner * g * d2)])
    else:
        grad = 0.5 * np.einsum("ij,ijt->t", inner * g, D)
    return lml, grad


def flops_per_candidate(N: int, d: int, n_gp: int = 1) -> float:
    """Algorithmic flops per candidate: one triangular solve + k* build + mu/sigma."""
    return n_gp * (float(N) * N + (3.0 * d + 12.0) * N)
