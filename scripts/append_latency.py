"""gpbo_fit (from scratch) against gpbo_fit_append (one new observation) at fixed theta.  Writes
gpurun_out/append_latency.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd.engine import MATERN25, GpEngine  # noqa: E402

eng = GpEngine(0)
out = {}
for N, d in ((512, 8), (1024, 16), (2048, 16), (4096, 16), (8192, 32)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(N + 40, d))
    y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(N + 40)
    norm = lambda v: (v - v.mean()) / v.std()  # noqa: E731
    n0 = N - 20                                     # appends stay inside the 64-row padding
    for _ in range(2):
        eng.fit(X[:n0], norm(y[:n0]), MATERN25, 1.5, 1e-6)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.fit(X[:n0], norm(y[:n0]), MATERN25, 1.5, 1e-6)
    t_fit = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for n in range(n0 + 1, n0 + 17):
        eng.fit_append(X[n - 1:n], norm(y[:n]))
    t_app = (time.perf_counter() - t0) / 16
    out[str(N)] = {"fit_ms": t_fit * 1e3, "append_one_row_ms": t_app * 1e3}
    print(N, out[str(N)], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "append_latency.json"), "w"), indent=1)
