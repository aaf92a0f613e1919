"""Round-3 Cholesky probe (MI355X): the 128-column-step schedule (chol_kernels.hip) against the round-2 schedule and
against numpy, on the Cholesky alone (gpbo_debug_cholesky) and inside gpbo_fit.  Writes gpurun_out/r03_chol_probe.json."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

PHASES = ["load", "factor 0..63 (+L10)", "syrk", "sub+barrier, factor 64..127", "inv(L00) tail", "inv(L11)"]


def spd(n, seed=0, kind="kernel"):
    rng = np.random.RandomState(seed)
    if kind == "kernel":     # the matrices the fit factorises: Matern-2.5 kernel matrix + 1e-6 I
        d = 16
        X = rng.uniform(size=(n, d)) / 1.5
        G = X @ X.T
        sq = np.maximum(np.diag(G)[:, None] + np.diag(G)[None, :] - 2 * G, 0.0)
        k = np.sqrt(5.0 * sq)
        K = (1 + k + k * k / 3) * np.exp(-k)
        np.fill_diagonal(K, 1.0 + 1e-6)
        return K
    B = rng.standard_normal((n, n))
    return B @ B.T / n + np.eye(n)


def probe():
    from bayesianoptimization_amd.engine import MATERN25, GpEngine
    eng = GpEngine(0, debug=True)
    out = {"chol": {}, "fit": {}, "variant_env": os.environ.get("GPBO_CHOL", "3")}
    for n in (64, 128, 192, 512, 576, 1024, 2048, 4096, 8192):
        for kind in ("random", "kernel"):
            if kind == "random" and n not in (128, 576):
                continue
            A = spd(n, 1, kind)
            Lref = np.linalg.cholesky(A)
            rec = {}
            for variant in (2, 3):
                L, dinv, stamps, ms, info = eng.debug_cholesky(A, variant=variant, iters=5)
                err = np.linalg.norm(L - Lref) / np.linalg.norm(Lref)
                derr = 0.0
                for b in range(n // 64):
                    blk = Lref[64 * b:64 * b + 64, 64 * b:64 * b + 64]
                    derr = max(derr, np.linalg.norm(dinv[b] @ blk - np.eye(64)))
                rec[f"v{variant}"] = {"ms": ms, "rel_err_L": float(err), "dinv_resid": float(derr), "info": info,
                                      "us_per_col": ms * 1e3 / n}
                if variant == 3:
                    d = np.diff(stamps[:7])
                    rec["stamps_cycles"] = {PHASES[i]: int(d[i]) for i in range(6)}
                    rec["stamps_cycles"]["wave 0: its 16 columns of the first factorisation"] = int(stamps[7] - stamps[1])
                    L2 = eng.debug_cholesky(A, variant=3, iters=1)[0]
                    rec["bitwise_repeat"] = bool(np.array_equal(L, L2))
            out["chol"][f"{n}/{kind}"] = rec
            print(n, kind, json.dumps(rec), flush=True)
    # not positive definite: LAPACK's info = order of the first non-positive leading minor
    A = spd(512, 2, "random")
    A[300, 300] = -1.0
    out["info_bad_pivot_301"] = [eng.debug_cholesky(A, variant=v)[4] for v in (2, 3)]
    print("info", out["info_bad_pivot_301"], flush=True)
    # determinism under repetition
    A = spd(1024, 3, "kernel")
    L0 = eng.debug_cholesky(A, variant=3)[0]
    out["bitwise_20_repeats_1024"] = all(np.array_equal(L0, eng.debug_cholesky(A, variant=3)[0]) for _ in range(20))
    print("repeat", out["bitwise_20_repeats_1024"], flush=True)
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
        r = {"fit_ms": float(b[0]), "kmat_ms": float(b[1]), "cholesky_ms": float(b[2]), "trtri_ms": float(b[3]),
             "us_per_col": float(b[2]) * 1e3 / N, "chol_TFLOPs": N**3 / 3 / (b[2] * 1e-3) / 1e12}
        out["fit"][str(N)] = r
        print(N, r, flush=True)
    return out


if __name__ == "__main__":
    if "--child" in sys.argv:
        print("JSON" + json.dumps(probe()))
        sys.exit(0)
    res = {}
    for flag in ("3", "2"):
        env = dict(os.environ, GPBO_CHOL=flag)
        p = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True, timeout=110)
        sys.stderr.write(p.stderr[-3000:])
        for line in p.stdout.splitlines():
            if line.startswith("JSON"):
                res["GPBO_CHOL=" + flag] = json.loads(line[4:])
            else:
                print(f"[chol={flag}]", line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r03_chol_probe.json"), "w"), indent=1)
