"""A/B of one debug switch of the large fit path (libgpbo_dbg.so reads it per call): the Cholesky alone, one LML value + gradient
at N = 2048 / 4096 / 8192, six LML lanes at N = 4096, a fixed-theta fit at 4096 — and whether the two settings give the same bits.

    python scripts/r06_fit_ab.py GPBO_CHOL_STEP_TPW 2 3   > profiles/r06_chol_step_tpw_ab.json
    python scripts/r06_fit_ab.py GPBO_GEMM_FAT 0 1        > profiles/r06_gemm_fat_ab.json

What is measured replaces: cholesky(K, lower=True) (sklearn _gpr.py:349) and log_marginal_likelihood(theta, eval_gradient=True)
(_gpr.py:575-652), d = 16, Matern-2.5, the smooth target of scripts/theta_search_timing.py.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

MATERN25 = 1


def matern25_K(X, ls, noise):
    """A positive definite test matrix for the Cholesky-alone entry: Matern-2.5 over X / ls + noise I (kernels.py:1711-1738)."""
    Z = X / ls
    d2 = np.maximum((Z * Z).sum(1)[:, None] + (Z * Z).sum(1)[None, :] - 2.0 * Z @ Z.T, 0.0)
    r = np.sqrt(5.0 * d2)
    K = (1.0 + r + r * r / 3.0) * np.exp(-r)
    K[np.diag_indices_from(K)] = 1.0 + noise
    return K


def data(N, d=16):
    rng = np.random.RandomState(N)
    X = rng.uniform(size=(N, d))
    y = np.exp(-((X - 0.5) ** 2).sum(1)) + 0.01 * rng.standard_normal(N)
    return X, (y - y.mean()) / y.std()


def med(f, n=9):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


def main():
    var, settings = sys.argv[1], sys.argv[2:]
    sizes = (2048, 4096) if "--no-8192" in settings else (2048, 4096, 8192)
    settings = [s for s in settings if not s.startswith("--")]
    eng = GpEngine(0, debug=True)
    out = {"switch": var, "rows": {}}
    vals = {}
    for setting in settings:
        os.environ[var] = setting
        r = {}
        for N in sizes:
            X, yn = data(N)
            if N <= 4096:
                K = matern25_K(X, 0.9, 1e-6)
                L, dinv, stamps, ms, info = eng.debug_cholesky(K, variant=3, iters=10)
                vals[(setting, N, "L")] = L
                r[f"cholesky_N{N}_ms"] = round(ms, 4)
                if setting == settings[0]:
                    Lr = np.linalg.cholesky(K)
                    r[f"cholesky_N{N}_rel_err_vs_lapack"] = float(np.max(np.abs(L - Lr)) / np.max(np.abs(Lr)))
            vals[(setting, N)] = eng.lml(X, yn, MATERN25, [0.9], 1e-6, eval_gradient=True)
            m_, lo = med(lambda: eng.lml(X, yn, MATERN25, [0.9], 1e-6, eval_gradient=True))
            r[f"lml_grad_N{N}_ms"] = round(m_, 4)
            r[f"lml_grad_N{N}_min_ms"] = round(lo, 4)
        X, yn = data(4096)
        th = np.array([[0.5], [0.7], [0.9], [1.2], [1.6], [2.2]])
        vals[(setting, "lanes")] = eng.lml_batch(X, yn, MATERN25, th, 1e-6)
        r["lml_6_lanes_N4096_ms"] = round(med(lambda: eng.lml_batch(X, yn, MATERN25, th, 1e-6, reuse_inputs=True), 5)[0], 4)
        r["lml_2_lanes_N4096_ms"] = round(med(lambda: eng.lml_batch(X, yn, MATERN25, th[:2], 1e-6, reuse_inputs=True), 5)[0], 4)
        r["fit_fixed_theta_N4096_ms"] = round(med(lambda: eng.fit(X, yn, MATERN25, [0.9], 1e-6))[0], 4)
        vals[(setting, "alpha")] = eng.get_alpha(4096)
        out["rows"][f"{var}={setting}"] = r
        print(setting, r, file=sys.stderr, flush=True)
    os.environ.pop(var, None)
    a, b = settings[0], settings[-1]
    out["bitwise_equal_L"] = bool(all(np.array_equal(vals[(a, N, "L")], vals[(b, N, "L")]) for N in sizes if N <= 4096))
    out["bitwise_equal_lml"] = bool(all(vals[(a, N)][0] == vals[(b, N)][0] and np.array_equal(vals[(a, N)][1], vals[(b, N)][1]) for N in sizes)
                                    and all(x[0] == y[0] and np.array_equal(x[1], y[1]) for x, y in zip(vals[(a, "lanes")], vals[(b, "lanes")])))
    out["lml_rel_diff"] = {str(N): [abs(vals[(a, N)][0] - vals[(b, N)][0]) / abs(vals[(a, N)][0]),
                                    float(np.max(np.abs(vals[(a, N)][1] - vals[(b, N)][1]) / np.abs(vals[(a, N)][1])))] for N in sizes}
    out["alpha_rel_diff"] = float(np.max(np.abs(vals[(a, "alpha")] - vals[(b, "alpha")])) / np.max(np.abs(vals[(a, "alpha")])))
    # a lane of the batch must stay bitwise gpbo_lml under every setting (lane 2 has theta 0.9 = the single evaluation's)
    out["lane_equals_single"] = bool(all(vals[(s, "lanes")][2][0] == vals[(s, 4096)][0] and np.array_equal(vals[(s, "lanes")][2][1], vals[(s, 4096)][1])
                                         for s in settings))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
