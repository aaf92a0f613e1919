"""predict() latency against the batch size on the two device paths (batched GEMV vs MFMA tiles), the
measurement behind small_batch_limit() in posterior_small.hip.  Writes gpurun_out/small_batch_latency.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd.engine import MATERN25, GpEngine  # noqa: E402

eng = GpEngine(0, debug=True)
out = {}
for N, d in ((512, 8), (1024, 16), (2048, 16), (4096, 16), (8192, 32)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(N, d))
    y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(N)
    yn = (y - y.mean()) / y.std()
    eng.fit(X, yn, MATERN25, 1.5, 1e-6)
    row = {}
    for M in (1, 17, 72, 170, 340, 680, 1024, 2048, 4096):
        Xc = rng.uniform(size=(M, d))
        rec = {}
        for label, env in (("gemv", "1024"), ("mfma", "0")):
            if label == "gemv" and M > 1024:
                continue
            os.environ["GPBO_SMALL_MAX"] = env
            for _ in range(3):
                eng.predict(Xc, y_mean=0.0, y_std=1.0)
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.predict(Xc, y_mean=0.0, y_std=1.0)
            rec[label + "_us"] = (time.perf_counter() - t0) / reps * 1e6
        row[str(M)] = rec
    os.environ.pop("GPBO_SMALL_MAX", None)
    out[str(N)] = row
    print(N, row, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "small_batch_latency.json"), "w"), indent=1)
