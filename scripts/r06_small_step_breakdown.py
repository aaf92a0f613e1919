"""Where one bench step of the two small configs goes (C1: N = 25, M = 1024; C2: N = 512, M = 65 536): wall time of each of the
three engine calls of a step (fit at fixed theta | posterior | acquisition + arg-best), each ended by its own stream
synchronisation, beside the whole step clocked as bench.py clocks it (one synchronisation, inside the arg-best read-back), the
device's own event times, and a cProfile of 2000 steps by own time — once with the calls recording their HIP event pairs (a new
engine's state) and once without (gpbo_set_timing(ctx, 0): what accelerate() sets).

    python scripts/r06_small_step_breakdown.py > profiles/r06_small_step_breakdown.txt
"""
import cProfile
import dataclasses
import io
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd import workloads as W  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402


def resolved(w):
    if w.length_scale is not None:
        return w
    g = np.load(os.path.join(ROOT, "tests", "golden", f"{w.name}.npz"))
    return dataclasses.replace(w, length_scale=float(np.atleast_1d(g["length_scale"])[0]))


def med(fn, n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e6)


def one(eng, name, n):
    w = resolved(W.ALL[name])
    X, y, _ = W.make_observations(w)
    y_mean, y_std = float(np.mean(y)), float(np.std(y))
    yn = (y - y_mean) / y_std
    y_max = W.feasible_y_max(w, y, None)
    eng.set_candidates(W.make_candidates(w.bounds_array(), w.M, 7))
    ym = 0.0 if y_max is None else y_max

    def fit():
        eng.fit(X, yn, w.kernel, w.length_scale, w.noise, slot=0)

    def post():
        eng.posterior(0, y_mean, y_std, fetch=False)

    def acq():
        return eng.acq_argbest(w.acq, w.acq_param, ym, None, None, k_seeds=10)

    def step():
        fit()
        post()
        return acq()

    for _ in range(50):
        step()
    eng.synchronize()
    print(f"== {name}: N = {w.N}, d = {w.d}, M = {w.M}   (microseconds, medians of {n})")
    print(f"whole step, one synchronisation (bench.py's step)        {med(step, n):8.1f}")

    def fit_s():
        fit()
        eng.synchronize()

    def post_s():
        post()
        eng.synchronize()

    print(f"fit + synchronise                                        {med(fit_s, n):8.1f}")
    print(f"posterior + synchronise                                  {med(post_s, n):8.1f}")
    print(f"acquisition + arg-best (reads back: synchronises)        {med(acq, n):8.1f}")
    print(f"fit, enqueue only (returns when the pivot word is known) {med(fit, n):8.1f}")
    eng.synchronize()
    print(f"posterior, enqueue only                                  {med(post, n):8.1f}")
    eng.synchronize()
    step()
    t = eng.last_timings()
    print("device events of the last step (ms):", {k: round(v, 4) for k, v in t.items()})
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        step()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
    print("\n".join(line for line in s.getvalue().splitlines() if line.strip())[:3500])


def main():
    eng = GpEngine(0)
    for timing in (True, False):
        eng.set_timing(timing)
        print(f"######## event pairs recorded: {timing} (gpbo_set_timing)")
        one(eng, "C1", 400)
        one(eng, "C2", 200)


if __name__ == "__main__":
    main()
