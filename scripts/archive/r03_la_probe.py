"""Round-3 look-ahead probe (MI355X): the outer-panel look-ahead of the Cholesky (chol_kernels.hip) off / on a plain
bulk stream / on a CU-masked bulk stream; fit stage timings and a hash of L (must not depend on the schedule)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    from bayesianoptimization_amd.engine import MATERN25, GpEngine
    eng = GpEngine(0, debug=True)
    out = {}
    for N, d in ((2048, 16), (4096, 16), (8192, 32)):
        rng = np.random.RandomState(0)
        X = rng.uniform(size=(N, d))
        y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(N)
        yn = (y - y.mean()) / y.std()
        ls = 1.5 if d == 16 else 2.0
        for _ in range(2):
            eng.fit(X, yn, MATERN25, ls, 1e-6)
        ts = []
        for _ in range(6):
            eng.fit(X, yn, MATERN25, ls, 1e-6)
            t = eng.last_timings()
            ts.append((t["fit"], t["kmat"], t["cholesky"], t["trtri"]))
        b = np.min(np.array(ts), axis=0)
        L = eng.get_L(N)
        out[str(N)] = {"fit_ms": float(b[0]), "cholesky_ms": float(b[2]), "trtri_ms": float(b[3]),
                       "L_sha": hashlib.sha256(np.ascontiguousarray(L).tobytes()).hexdigest()[:16]}
    print("JSON" + json.dumps(out))


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
        sys.exit(0)
    res = {}
    settings = [("off", {"GPBO_CHOL_LA": "0"}), ("plain", {"GPBO_CHOL_LA_CUS": "0"})]
    for cus in sys.argv[1:] or ("128", "192", "224"):
        settings.append((f"cus{cus}", {"GPBO_CHOL_LA_CUS": cus}))
    for name, extra in settings:
        env = dict(os.environ, **extra)
        p = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True, timeout=200)
        sys.stderr.write(p.stderr[-2000:])
        for line in p.stdout.splitlines():
            if line.startswith("JSON"):
                res[name] = json.loads(line[4:])
                print(name, line[4:], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r03_la_probe.json"), "w"), indent=1)
