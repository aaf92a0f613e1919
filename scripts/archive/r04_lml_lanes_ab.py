"""gpbo_lml_batch at N = 2048 / 4096 for 1..6 live lanes: one stream (+ graph) per lane — the product's rule from NP = 2048 — against
groups of 2 / 3 / all lanes sharing the launches (debug build: GPBO_LML_PER_GROUP).  One process per setting (the switch is read
per call, but the cached graphs are keyed by the group layout).  Usage: GPBO_LML_PER_GROUP=k python scripts/archive/r04_lml_lanes_ab.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bayesianoptimization_amd.engine import MATERN25, GpEngine  # noqa: E402

eng = GpEngine(0, debug=True)
out = {"per_group": os.environ.get("GPBO_LML_PER_GROUP", "product rule (1 from NP = 2048)"), "ms": {}}
for N in (2048, 4096):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(N, 16))
    y = np.exp(-((X - 0.5) ** 2).sum(1)) + 0.01 * rng.standard_normal(N)
    yn = (y - y.mean()) / y.std()
    for n in (1, 2, 3, 5, 6):
        sc = np.linspace(0.8, 2.5, n)[:, None]
        for _ in range(3):
            eng.lml_batch(X, yn, MATERN25, sc, 1e-6)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter()
            eng.lml_batch(X, yn, MATERN25, sc, 1e-6, reuse_inputs=True)
            ts.append((time.perf_counter() - t0) * 1e3)
        out["ms"][f"{N}/{n}"] = round(float(np.median(ts)), 3)
print(json.dumps(out))
