"""gpbo_lml_batch latency for 1 / 6 / 8 lanes per N; run once as is (hipGraph replay) and once with GPBO_LML_GRAPH=0."""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bayesianoptimization_amd.engine import MATERN25, GpEngine
eng = GpEngine(0, debug=True)
for N, d in ((256, 4), (512, 8), (1024, 16), (2048, 16)):
    rng = np.random.RandomState(0); X = rng.uniform(size=(N, d)); y = np.sin(X.sum(1)); yn = (y - y.mean()) / y.std()
    for n in (1, 6, 8):
        sc = np.linspace(0.5, 3.0, n)[:, None]
        for _ in range(3): eng.lml_batch(X, yn, MATERN25, sc, 1e-6)
        t0 = time.perf_counter()
        for _ in range(10): eng.lml_batch(X, yn, MATERN25, sc, 1e-6)
        print(N, n, "lanes:", round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms", flush=True)
