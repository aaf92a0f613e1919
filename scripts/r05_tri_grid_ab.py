"""Lower-only fit GEMMs (W^T W, the SYRK-shaped trailing updates) on a 1-D grid of their LIVE tiles in lower-triangle order against
the square grid whose upper half exits at once (launch_gemm, csrc/fit_kernels.hip; debug build: GPBO_TRI_GRID=0 is the square grid).

    python scripts/r05_tri_grid_ab.py [pmc]  > profiles/r05_tri_grid_ab.json

Per setting: the SYRK-shaped product of a rank-1024 trailing update (3072 x 3072 x 1024, lower) by itself (HIP events, 20 launches),
one LML value + gradient at N = 2048 / 4096 / 8192 (d = 16, wall, median of 7) and six lanes at N = 4096; the results must be bitwise
equal (same tiles, same arithmetic).  With the argument `pmc` the script runs ONE LML + gradient at N = 4096 per setting and nothing
else (for rocprofv3 --pmc / --kernel-trace passes around it: GPBO_TRI_GRID is then taken from the environment).
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

MATERN25 = 1


def data(N, d=16):
    rng = np.random.RandomState(N)
    X = rng.uniform(size=(N, d))
    y = np.exp(-((X - 0.5) ** 2).sum(1)) + 0.01 * rng.standard_normal(N)
    return X, (y - y.mean()) / y.std()


def med(f, n=7):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def main():
    eng = GpEngine(0, debug=True)
    if len(sys.argv) > 1 and sys.argv[1] == "pmc":
        X, yn = data(4096)
        for _ in range(3):
            v = eng.lml(X, yn, MATERN25, [0.9], 1e-6, eval_gradient=True)
        print(json.dumps({"GPBO_TRI_GRID": os.environ.get("GPBO_TRI_GRID", "1"), "lml": v[0], "grad": list(v[1])}))
        return
    out = {"what": __doc__.strip().split("\n")[0], "rows": {}}
    vals = {}
    for setting in ("0", "1"):
        os.environ["GPBO_TRI_GRID"] = setting
        r = {"syrk_3072x3072x1024_lower": eng.gemm_bench(3072, 3072, 1024, b_trans=True, lower_only=True, iters=20)}
        for N in (2048, 4096, 8192):
            X, yn = data(N)
            vals[(setting, N)] = eng.lml(X, yn, MATERN25, [0.9], 1e-6, eval_gradient=True)
            r[f"lml_grad_N{N}_ms"] = round(med(lambda: eng.lml(X, yn, MATERN25, [0.9], 1e-6, eval_gradient=True)), 4)
        X, yn = data(4096)
        th = np.array([[0.5], [0.7], [0.9], [1.2], [1.6], [2.2]])
        vals[(setting, "lanes")] = eng.lml_batch(X, yn, MATERN25, th, 1e-6)
        r["lml_6_lanes_N4096_ms"] = round(med(lambda: eng.lml_batch(X, yn, MATERN25, th, 1e-6, reuse_inputs=True), 5), 4)
        out["rows"]["square_grid" if setting == "0" else "live_tiles_1d"] = r
        print(setting, r, file=sys.stderr, flush=True)
    os.environ.pop("GPBO_TRI_GRID", None)
    same = all(vals[("0", N)][0] == vals[("1", N)][0] and np.array_equal(vals[("0", N)][1], vals[("1", N)][1]) for N in (2048, 4096, 8192))
    same = same and all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(vals[("0", "lanes")], vals[("1", "lanes")]))
    out["bitwise_equal"] = bool(same)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
