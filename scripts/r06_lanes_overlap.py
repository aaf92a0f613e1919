"""How much of each other do the lanes of one gpbo_lml_batch call hide at N = 4096 (one stream per lane there)?  Wall time of
1 / 2 / 3 / 4 / 6 lanes (median of 7 after 3), value + gradient, d = 16 — and, with `trace` as the first argument, a workload for
rocprofv3 --kernel-trace --stats: `trace1` = 24 single evaluations, `trace6` = 4 six-lane calls (the per-kernel totals of
docs/LAB_NOTEBOOK.md §10.8), `trace` = 4 two-lane calls, then 4 six-lane calls.

    python scripts/r06_lanes_overlap.py > profiles/r06_lanes_overlap.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd.engine import MATERN25, GpEngine  # noqa: E402

N = int(os.environ.get("LANES_N", 4096))
eng = GpEngine(0, debug=os.environ.get("LANES_DEBUG") == "1")
rng = np.random.RandomState(0)
X = rng.uniform(size=(N, 16))
y = np.exp(-((X - 0.5) ** 2).sum(1)) + 0.01 * rng.standard_normal(N)
yn = (y - y.mean()) / y.std()
scales = np.array([[0.8], [1.0], [1.3], [1.6], [2.0], [2.5]])


def call(n):
    t0 = time.perf_counter()
    eng.lml_batch(X, yn, MATERN25, scales[:n], 1e-6)
    return (time.perf_counter() - t0) * 1e3


if len(sys.argv) > 1 and sys.argv[1] in ("trace1", "trace6"):       # kernel-stats workloads: 24 evaluations either way
    n = int(sys.argv[1][-1])
    for _ in range(24 // n):
        print(n, call(n))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "trace":
    for n in (2, 6):
        for _ in range(4):
            print(n, call(n))
    sys.exit(0)
out = {"N": N, "d": 16, "env": {k: os.environ[k] for k in os.environ if k.startswith(("GPU_MAX", "GPBO_", "HIP_", "HSA_"))}, "lanes_ms": {}}
for n in (1, 2, 3, 4, 6):
    ts = [call(n) for _ in range(10)]
    out["lanes_ms"][str(n)] = round(float(np.median(ts[3:])), 4)
one = out["lanes_ms"]["1"]
out["lanes_over_sequential"] = {k: round(v / (int(k) * one), 3) for k, v in out["lanes_ms"].items()}
print(json.dumps(out))
