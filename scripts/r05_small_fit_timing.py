"""Fit / LML evaluation per problem size on the three fit paths (MI355X): the multi-launch sequence, the one-workgroup kernel
(csrc/fused_small.hip, NP <= 128) and the strip path (csrc/mid_fit.hip, NP <= 1024).

    python scripts/r05_small_fit_timing.py > profiles/r05_small_fit_timing.json

Debug build (GPBO_FUSED_MAX_NP / GPBO_MID_MAX_NP are read per call).  Per N and path: the fit's device time between its HIP events,
the fit's wall time as the caller sees it (gpbo_fit returns after the stream has drained), one LML value + gradient (gpbo_lml, wall),
six lanes and one lane of gpbo_lml_batch (wall; inputs resident).  Medians of 40 calls after 5 warm-ups.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

MATERN25 = 1


def med(f, n=40, warm=5):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def main():
    eng = GpEngine(0, debug=True)
    out = {"what": __doc__.strip().split("\n")[0], "d": 8, "kernel": "matern25", "rows": []}
    rng = np.random.RandomState(0)
    for N in (16, 64, 128, 160, 192, 256, 320, 384, 448, 512, 640, 768, 1024):
        d = 8
        X = rng.uniform(0, 1, size=(N, d))
        y = np.sin(3 * X.sum(1)) + 0.05 * rng.standard_normal(N)
        yn = (y - y.mean()) / y.std()
        ls = np.array([0.7])
        th6 = np.array([[0.3], [0.5], [0.7], [0.9], [1.3], [2.0]])
        row = {"N": N, "d": d}
        paths = [("multi_launch", 0, 0), ("strip", 0, 1024)]
        if N <= 128:
            paths.append(("one_workgroup", 128, 0))
        for name, fused, mid in paths:
            os.environ["GPBO_FUSED_MAX_NP"] = str(fused)
            os.environ["GPBO_MID_MAX_NP"] = str(mid)
            dev = []

            def fit():
                eng.fit(X, yn, MATERN25, ls, 1e-6)
                dev.append(eng.last_timings()["fit"])

            wall = med(fit)
            r = {"fit_device_ms": float(np.median(dev[5:])), "fit_wall_ms": wall}
            r["lml_grad_wall_ms"] = med(lambda: eng.lml(X, yn, MATERN25, ls, 1e-6, eval_gradient=True))
            eng.lml_batch(X, yn, MATERN25, th6, 1e-6)
            r["lml_6_lanes_wall_ms"] = med(lambda: eng.lml_batch(X, yn, MATERN25, th6, 1e-6, reuse_inputs=True))
            r["lml_1_lane_resident_wall_ms"] = med(lambda: eng.lml_batch(X, yn, MATERN25, th6[:1], 1e-6, reuse_inputs=True))
            row[name] = {k: round(v, 4) for k, v in r.items()}
        out["rows"].append(row)
        print(N, row, file=sys.stderr, flush=True)
    os.environ.pop("GPBO_FUSED_MAX_NP", None)
    os.environ.pop("GPBO_MID_MAX_NP", None)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
