import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd import workloads as W
from bayesianoptimization_amd.engine import GpEngine, F32
eng = GpEngine(0, debug=True)
for name, M in (("C5", 1 << 18), ("C3", 1 << 20)):
    w = W.ALL[name]
    X, y, c = W.make_observations(w)
    yn, ym, ys = W.normalize_targets(y)
    eng.set_candidates(W.make_candidates(w.bounds_array(), M, 7))
    eng.fit(X, yn, w.kernel, w.length_scale, w.noise, precision=F32)
    res = {"2": [], "4": []}; sds = {}
    for rnd in range(3):
        for rt in ("2", "4"):
            os.environ["GPBO_F32_RT"] = rt
            mu, sd = eng.posterior(0, ym, ys)
            res[rt].append(round(eng.last_timings()["posterior_main"], 2)); sds[rt] = sd
    os.environ.pop("GPBO_F32_RT")
    fl = W.flops_per_candidate(w.N, w.d) * M
    print(name, res, "TF", {k: round(fl / (min(v) * 1e-3) / 1e12, 1) for k, v in res.items()},
          "max|dsd| rt2 vs rt4", float(np.max(np.abs(sds["2"] - sds["4"]))), flush=True)
