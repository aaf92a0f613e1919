"""The posterior pass over bayes_opt's DEFAULT candidate count (n_random = 10 000) by padded size: the fused 8-wave kernel (v2:
256-row chunks, k* generated once per chunk), the slab + GEMM pair (v3) and the fused 16-wave kernel (v4: 512-row chunks), forced in
turn through GPBO_POST_KERNEL (debug build); wall ms of gpbo_posterior (median of 30), d = 4 and 8.

    python scripts/r06_post_10k_ab.py > profiles/r06_post_10k_ab.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd import workloads as W  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

eng = GpEngine(0, debug=True)
eng.set_timing(False)
rows = []
for d in (4, 8):
    for N in (200, 300, 384, 450, 512, 600, 768, 1024, 1500):
        for M in (10_000, 20_000):
            rng = np.random.RandomState(N + d)
            X = rng.uniform(size=(N, d))
            y = np.exp(-np.sum((X - 0.5) ** 2, axis=1)) + 0.01 * rng.standard_normal(N)
            ym, ys = float(y.mean()), float(y.std())
            eng.fit(X, (y - ym) / ys, W.MATERN25, 0.3 * np.sqrt(d), 1e-6, slot=0)
            eng.set_candidates(rng.uniform(size=(M, d)))
            row = {"N": N, "d": d, "M": M}
            for k in ("default", "2", "3", "4"):
                if k == "4" and N > 1024:
                    continue
                if k == "default":
                    os.environ.pop("GPBO_POST_KERNEL", None)
                else:
                    os.environ["GPBO_POST_KERNEL"] = k
                ts = []
                for _ in range(35):
                    eng.synchronize()
                    t0 = time.perf_counter()
                    eng.posterior(0, ym, ys, fetch=False)
                    eng.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                row["v" + k if k != "default" else "default"] = round(float(np.median(ts[5:])), 4)
            os.environ.pop("GPBO_POST_KERNEL", None)
            rows.append(row)
            print(row, file=sys.stderr)
print(json.dumps({"what": __doc__.split("\n\n")[0], "rows": rows}))
