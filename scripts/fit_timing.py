"""gpbo_fit stage timings (kernel matrix, Cholesky, W = L^-1) per N, with the one-level (GPBO_CHOL_OUTER=64) and the
two-level (default 256) Cholesky; K/L parity of the two checked bitwise-or-rounding."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd.engine import MATERN25, GpEngine  # noqa: E402

eng = GpEngine(0, debug=True)
out = {}
outers = sys.argv[1:] or ["64", "128", "256", "512"]
for N, d in ((1024, 16), (2048, 16), (4096, 16), (8192, 32)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(N, d))
    y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(N)
    yn = (y - y.mean()) / y.std()
    r, Lref = {}, None
    for outer in outers:
        os.environ["GPBO_CHOL_OUTER"] = outer
        for _ in range(2):
            eng.fit(X, yn, MATERN25, 1.5 if d == 16 else 2.0, 1e-6)
        ts = []
        for _ in range(3):
            eng.fit(X, yn, MATERN25, 1.5 if d == 16 else 2.0, 1e-6)
            t = eng.last_timings()
            ts.append((t["fit"], t["cholesky"], t["trtri"]))
        best = min(ts)
        L = eng.get_L(N)
        if Lref is None:
            Lref = L
        r[outer] = {"fit_ms": round(best[0], 3), "cholesky_ms": round(best[1], 3), "trtri_ms": round(best[2], 3),
                    "L_maxdiff_vs_first": float(np.max(np.abs(L - Lref)))}
    os.environ.pop("GPBO_CHOL_OUTER", None)
    out[str(N)] = r
    print(N, r, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fit_timing.json"), "w"), indent=1)
