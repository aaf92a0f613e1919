"""Where the host spends a default suggest() at small N: cProfile over the maximize()-shaped loop (scripts/r06_maximize_loop.py's
configuration) for N = 16 ... 144 on the device, printed by own time and by cumulative time.

    python scripts/r06_suggest_host_profile.py > profiles/r06_suggest_host_profile.txt
"""
import cProfile
import io
import os
import pstats
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sklearn.gaussian_process.kernels import Matern  # noqa: E402

from bayesianoptimization_amd import fused_acquisition as A  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402
from bayesianoptimization_amd.float_space import FloatSpace  # noqa: E402
from bayesianoptimization_amd.gpr import HipGPR  # noqa: E402

D = 4


def black_box(x):
    x = np.asarray(x, dtype=np.float64)
    return float(-np.sum((x - 0.3) ** 2) + 0.5 * np.sin(5.0 * x[0]) * np.cos(3.0 * x[1]))


def loop(n0, n1, profile):
    eng = GpEngine(0)
    if os.environ.get("GPBO_SCRIPT_TIMING") == "0":
        eng.set_timing(False)
    sp = FloatSpace({f"x{j}": (0.0, 1.0) for j in range(D)})
    rng = np.random.RandomState(1)
    X0 = rng.uniform(size=(n0, D))
    sp.register_bulk(X0, np.array([black_box(x) for x in X0]))
    gp = HipGPR(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5, random_state=np.random.RandomState(1),
                engine=eng)
    fn = A.UpperConfidenceBound(kappa=2.576)
    rs = np.random.RandomState(7)
    x = fn.suggest(gp, sp, n_random=10_000, n_smart=10, fit_gp=True, random_state=rs)     # first call: contexts, graphs, pools
    sp.register(x, black_box(x))
    ms, rounds = [], []
    if profile:
        profile.enable()
    while len(sp) < n1:
        t0 = time.perf_counter()
        x = fn.suggest(gp, sp, n_random=10_000, n_smart=10, fit_gp=True, random_state=rs)
        ms.append((time.perf_counter() - t0) * 1e3)
        rounds.append(int(gp.theta_search_rounds_))
        sp.register(x, black_box(x))
    if profile:
        profile.disable()
    return np.array(ms), np.array(rounds)


def main():
    warnings.simplefilter("ignore")
    n0, n1 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 144)
    ms, rounds = loop(n0, n1, None)
    print(f"unprofiled: {len(ms)} suggest() calls, N = {n0 + 1} ... {n1 - 1}: median {np.median(ms):.3f} ms, mean {np.mean(ms):.3f} ms, "
          f"theta-search rounds median {np.median(rounds):.0f}, ms per round (median of ratios) {np.median(ms / np.maximum(rounds, 1)):.4f}")
    pr = cProfile.Profile()
    ms, rounds = loop(n0, n1, pr)
    print(f"profiled:   median {np.median(ms):.3f} ms, mean {np.mean(ms):.3f} ms")
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
        print(s.getvalue())


if __name__ == "__main__":
    main()
