import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd import workloads as W
from bayesianoptimization_amd.engine import GpEngine
eng = GpEngine(0, debug=True)
w = W.C3
X, y, c = W.make_observations(w)
ym, ys = float(np.mean(y)), float(np.std(y)); yn = (y - ym) / ys
eng.fit(X, yn, w.kernel, w.length_scale, w.noise)
eng.set_candidates(W.make_candidates(w.bounds_array(), 1 << 18, 7))
res = {"normal": [], "no_gen": []}
for rnd in range(3):
    for name, v in (("normal", "0"), ("no_gen", "1")):
        os.environ["GPBO_POST_ABLATE_GEN"] = v
        eng.posterior(0, ym, ys, fetch=False)
        res[name].append(eng.last_timings()["posterior_main"])
print(res)
