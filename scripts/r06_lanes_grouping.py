"""Lanes per group for gpbo_lml_batch at NP >= 2048 (debug build, GPBO_LML_PER_GROUP): wall time of n = 2 ... 6 lanes when the
launch sequence runs once per group of p lanes (lane = a grid dimension inside a group, one stream per group), for every p that
divides the work differently.  p = 1 was the rule from NP = 2048 on until round 6.

    python scripts/r06_lanes_grouping.py > profiles/r06_lanes_grouping.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time

    import numpy as np

    sys.path.insert(0, ROOT)
    from bayesianoptimization_amd.engine import MATERN25, GpEngine

    N = int(sys.argv[2])
    eng = GpEngine(0, debug=True)
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(N, 16))
    y = np.exp(-((X - 0.5) ** 2).sum(1)) + 0.01 * rng.standard_normal(N)
    yn = (y - y.mean()) / y.std()
    scales = np.array([[0.8], [1.0], [1.3], [1.6], [2.0], [2.5]])
    res = {}
    for n in range(1, 7):
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            eng.lml_batch(X, yn, MATERN25, scales[:n], 1e-6)
            ts.append((time.perf_counter() - t0) * 1e3)
        res[str(n)] = round(float(np.median(ts[3:])), 4)
    print(json.dumps(res))
    sys.exit(0)

out = {"what": __doc__.split("\n\n")[0], "rows": {}}
for N in (2048, 3072, 4096, 6144):
    out["rows"][str(N)] = {}
    for p in (1, 2, 3, 6):
        env = dict(os.environ, GPBO_LML_PER_GROUP=str(p))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(N)], env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        out["rows"][str(N)][f"per_group_{p}"] = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
        print(N, p, out["rows"][str(N)][f"per_group_{p}"], file=sys.stderr)
print(json.dumps(out))
