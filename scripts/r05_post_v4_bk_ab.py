"""Round 5: the fused 16-wave posterior kernel of BASELINE config 2 with 64-point stages (half the workgroup
barriers of its 1024-thread workgroup; debug build: GPBO_POST_V4_BK=64) against the product's 32-point stages.  HIP-event time of the
posterior's main launch, best of 10, same inputs; results must be bitwise equal (same arithmetic, same order).
-> stdout (JSON)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd.engine import MATERN25, GpEngine  # noqa: E402

eng = GpEngine(0, debug=True)
out = {}
for N, d, M, ls in ((512, 8, 65536, 1.0), (448, 8, 65536, 1.0), (512, 16, 1 << 17, 1.5), (400, 4, 65536, 0.8)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(N, d))
    y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(N)
    yn = (y - y.mean()) / y.std()
    eng.fit(X, yn, MATERN25, ls, 1e-6)
    eng.set_candidates(rng.uniform(size=(M, d)))
    r, ref = {}, None
    os.environ["GPBO_POST_KERNEL"] = "4"
    for bk in ("32", "64", "32", "64"):
        os.environ["GPBO_POST_V4_BK"] = bk
        ts = []
        for _ in range(12):
            mu, sd = eng.posterior(0, 0.0, 1.0)
            ts.append(eng.last_timings()["posterior_main"])
        fl = (float(N) * N + (3 * d + 12) * N) * M
        r.setdefault(f"bk{bk}_ms", []).append(float(np.min(ts[2:])))
        r[f"bk{bk}_frac_of_78.6"] = fl / (min(r[f"bk{bk}_ms"]) * 1e-3) / 78.6e12
        if ref is None:
            ref = (mu, sd)
        else:
            r[f"bk{bk}_bitwise_equal_bk32"] = bool(np.array_equal(mu, ref[0]) and np.array_equal(sd, ref[1]))
    os.environ.pop("GPBO_POST_KERNEL"); os.environ.pop("GPBO_POST_V4_BK")
    out[f"N{N}_d{d}_M{M}"] = r
    print(N, d, M, r, file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
