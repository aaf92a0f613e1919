"""W^T W by split-k on 128x128 tiles (product) against the 64x64-tile launch of rounds 2-3 (GPBO_SPLITK=0, debug build): one
gpbo_lml value + gradient at N = 2048 / 3072 / 4096, d = 16 — median ms of 10 calls and the results (the two sum K^-1 in a different
order: equal to rounding, not bitwise).  Usage: [GPBO_SPLITK=0] python scripts/r04_splitk_ab.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bayesianoptimization_amd.engine import MATERN25, RBF, GpEngine  # noqa: E402

eng = GpEngine(0, debug=True)
out = {"splitk": os.environ.get("GPBO_SPLITK", "1"), "cases": {}}
for N in (2048, 3072, 4096):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(N, 16))
    y = np.exp(-((X - 0.5) ** 2).sum(1)) + 0.01 * rng.standard_normal(N)
    yn = (y - y.mean()) / y.std()
    for kind, name in ((MATERN25, "matern"), (RBF, "rbf")):
        ls = np.linspace(0.9, 1.6, 16) if kind == RBF else 1.3
        for _ in range(3):
            v, g = eng.lml(X, yn, kind, ls, 1e-6)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            v, g = eng.lml(X, yn, kind, ls, 1e-6)
            ts.append((time.perf_counter() - t0) * 1e3)
        out["cases"][f"{N}/{name}"] = {"ms": round(float(np.median(ts)), 3), "lml": float(v), "grad": [float(x) for x in np.atleast_1d(g)[:3]]}
print(json.dumps(out))
