"""Round 4: the posterior for 384 <= NP <= 1024 at chip-filling batches — fused 256-row-chunk kernel (GPBO_POST_KERNEL=2), k*
slab + GEMM (3), fused 16-wave kernel with 512-row chunks (4, the new default there).  HIP-event time of the posterior's main
launch(es), best of 8, same inputs, and the max difference between the paths.  Debug build (the switch is a debug switch).
-> gpurun_out/r04_post_small_np_ab.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd.engine import MATERN25, GpEngine  # noqa: E402

eng = GpEngine(0, debug=True)
out = {}
for N, d, M, ls in ((512, 8, 65536, 1.0), (448, 8, 65536, 1.0), (1024, 16, 65536, 1.5), (768, 16, 1 << 18, 1.5), (512, 8, 8192, 1.0)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(N, d))
    y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(N)
    yn = (y - y.mean()) / y.std()
    eng.fit(X, yn, MATERN25, ls, 1e-6)
    eng.set_candidates(rng.uniform(size=(M, d)))
    r, ref = {}, None
    for path in ("2", "3", "4"):
        os.environ["GPBO_POST_KERNEL"] = path
        ts = []
        for _ in range(10):
            mu, sd = eng.posterior(0, 0.0, 1.0)
            ts.append(eng.last_timings()["posterior_main"])
        os.environ.pop("GPBO_POST_KERNEL")
        fl = (float(N) * N + (3 * d + 12) * N) * M
        r[f"v{path}_ms"] = float(np.min(ts[2:]))
        r[f"v{path}_frac_of_78.6"] = fl / (r[f"v{path}_ms"] * 1e-3) / 78.6e12
        if ref is None:
            ref = (mu, sd)
        else:
            r[f"v{path}_max_abs_diff_vs_v2"] = [float(np.max(np.abs(mu - ref[0]))), float(np.max(np.abs(sd - ref[1])))]
    out[f"N{N}_d{d}_M{M}"] = r
    print(N, d, M, r, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_post_small_np_ab.json"), "w"), indent=1)
