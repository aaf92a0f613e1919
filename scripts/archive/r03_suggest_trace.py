"""Whole suggest() calls at one BASELINE config for a kernel trace and a host profile (where does ms/suggest go beyond the
resident step?):  python scripts/archive/r03_suggest_trace.py C2 [n_smart] [device|reference] [reps]
Prints the median wall time and the cProfile top of the timed calls.  Run on the GPU box, e.g. under
rocprofv3 --kernel-trace --stats."""
import cProfile
import io
import os
import pstats
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sklearn.gaussian_process.kernels import Matern  # noqa: E402

from bayesianoptimization_amd import fused_acquisition as A  # noqa: E402
from bayesianoptimization_amd import workloads as W  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402
from bayesianoptimization_amd.float_space import FloatSpace  # noqa: E402
from bayesianoptimization_amd.gpr import HipGPR  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n_smart = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mode = sys.argv[3] if len(sys.argv) > 3 else "reference"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
w = W.ALL[name]
X, y, _ = W.make_observations(w)
eng = GpEngine(0)
sp = FloatSpace(w.pbounds())
sp.register_bulk(X, y)
gp = HipGPR(kernel=Matern(nu=2.5, length_scale=w.length_scale), alpha=w.noise, normalize_y=True, optimizer=None, engine=eng,
            incremental=False)
fn = {W.UCB: lambda: A.UpperConfidenceBound(kappa=w.acq_param), W.EI: lambda: A.ExpectedImprovement(xi=w.acq_param),
      W.POI: lambda: A.ProbabilityOfImprovement(xi=w.acq_param)}[w.acq]()
fn.device_polish = (mode == "device")
M = w.M // 8 if name in ("C4", "C5") else w.M
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(3):
        fn.suggest(gp, sp, n_random=M, n_smart=n_smart, fit_gp=True, random_state=np.random.RandomState(7 + rep))
    ts = []
    pr = cProfile.Profile()
    for rep in range(reps):
        rs = np.random.RandomState(100 + rep)
        t0 = time.perf_counter()
        pr.enable()
        fn.suggest(gp, sp, n_random=M, n_smart=n_smart, fit_gp=True, random_state=rs)
        pr.disable()
        ts.append((time.perf_counter() - t0) * 1e3)
print(f"{name} n_smart={n_smart} {mode}: median {np.median(ts):.3f} ms, min {np.min(ts):.3f} ms over {reps} calls (profiler on)")
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(22)
print("\n".join(line[:150] for line in out.getvalue().splitlines()[:48]))
eng.close()
