"""cProfile of one C3 suggest(n_smart=0) through the seams: where the host time goes."""
import cProfile
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
from bayesianoptimization_amd import workloads as W  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402
from bayesianoptimization_amd.float_space import FloatSpace  # noqa: E402
from bayesianoptimization_amd.gpr import HipGPR  # noqa: E402

warnings.simplefilter("ignore")
eng = GpEngine(0)
w = W.ALL[sys.argv[1] if len(sys.argv) > 1 else "C3"]
X, y, c = W.make_observations(w)
sp = FloatSpace(w.pbounds())
sp.register_bulk(X, y)
fn = A.UpperConfidenceBound(kappa=2.576)
gp = HipGPR(kernel=Matern(nu=2.5, length_scale=w.length_scale), alpha=w.noise, normalize_y=True, optimizer=None, engine=eng)
for _ in range(2):
    fn.suggest(gp, sp, n_random=w.M, n_smart=int(os.environ.get("NSMART", "0")), fit_gp=True, random_state=np.random.RandomState(7))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
fn.suggest(gp, sp, n_random=w.M, n_smart=int(os.environ.get("NSMART", "0")), fit_gp=True, random_state=np.random.RandomState(7))
pr.disable()
print("wall ms", (time.perf_counter() - t0) * 1e3)
print("device timings", eng.last_timings())
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
