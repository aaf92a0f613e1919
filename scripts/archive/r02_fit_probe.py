"""Round-2 fit-path probe (MI355X): fp64 GEMM throughput of the fit kernels (128x128 double-buffered vs the 64x64
kernel, GPBO_GEMM128=0 in a child process), gpbo_fit stage timings per N, kernel-matrix assembly GB/s.
Writes gpurun_out/r02_fit_probe.json."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def probe():
    from bayesianoptimization_amd.engine import MATERN25, GpEngine
    eng = GpEngine(0, debug=True)
    out = {"gemm128": os.environ.get("GPBO_GEMM128", "1"), "gemm": {}, "fit": {}}
    for (m, n, k, bt, at, lo, tag) in [(4096, 4096, 4096, 1, 0, 0, "NT 4096^3"), (4096, 4096, 4096, 0, 0, 0, "NN 4096^3"),
                                       (4096, 4096, 4096, 0, 1, 0, "TN 4096^3"), (3584, 3584, 512, 1, 0, 1, "SYRK 3584 k512 lower"),
                                       (7680, 7680, 512, 1, 0, 1, "SYRK 7680 k512 lower"), (2048, 448, 64, 1, 0, 0, "panel update 2048x448 k64"),
                                       (1024, 1024, 1024, 0, 0, 0, "NN 1024^3"), (512, 512, 512, 1, 0, 0, "NT 512^3")]:
        r = eng.gemm_bench(m, n, k, bt, at, lo, iters=10)
        out["gemm"][tag] = r
        print(tag, r, flush=True)
    for N, d in ((512, 8), (1024, 16), (2048, 16), (4096, 16), (8192, 32)):
        rng = np.random.RandomState(0)
        X = rng.uniform(size=(N, d))
        y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(N)
        yn = (y - y.mean()) / y.std()
        ls = 1.0 if d == 8 else 1.5 if d == 16 else 2.0
        for _ in range(2):
            eng.fit(X, yn, MATERN25, ls, 1e-6)
        ts = []
        for _ in range(5):
            eng.fit(X, yn, MATERN25, ls, 1e-6)
            t = eng.last_timings()
            ts.append((t["fit"], t["kmat"], t["cholesky"], t["trtri"]))
        b = np.min(np.array(ts), axis=0)
        NP = (N + 63) // 64 * 64
        r = {"fit_ms": float(b[0]), "kmat_ms": float(b[1]), "cholesky_ms": float(b[2]), "trtri_ms": float(b[3]),
             "kmat_GBps": NP * (NP + 64) / 2 * 8 / (b[1] * 1e-3) / 1e9, "chol_TFLOPs": N**3 / 3 / (b[2] * 1e-3) / 1e12}
        out["fit"][str(N)] = r
        print(N, r, flush=True)
    return out


if __name__ == "__main__":
    if "--child" in sys.argv:
        print("JSON" + json.dumps(probe()))
        sys.exit(0)
    res = {}
    for flag in ("1", "0"):
        env = dict(os.environ, GPBO_GEMM128=flag)
        p = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        sys.stderr.write(p.stderr[-2000:])
        for line in p.stdout.splitlines():
            if line.startswith("JSON"):
                res["gemm128=" + flag] = json.loads(line[4:])
            else:
                print(f"[gemm128={flag}]", line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r02_fit_probe.json"), "w"), indent=1)
