import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd import workloads as W
from bayesianoptimization_amd.engine import GpEngine, F32, F64
eng = GpEngine(0)
for name, M in (("C3", 1 << 20), ("C5", 1 << 18)):
    w = W.ALL[name]
    X, y, c = W.make_observations(w)
    yn, ym, ys = W.normalize_targets(y)
    eng.set_candidates(W.make_candidates(w.bounds_array(), M, 7))
    res = {}
    for pname, prec in (("f64", F64), ("f32", F32)):
        eng.fit(X, yn, w.kernel, w.length_scale, w.noise, precision=prec)
        ts = []
        for _ in range(3):
            mu, sd = eng.posterior(0, ym, ys)
            ts.append(eng.last_timings()["posterior_main"])
        res[pname] = {"ms": ts, "tflops": W.flops_per_candidate(w.N, w.d) * M / (min(ts) * 1e-3) / 1e12}
        res[pname + "_sd"] = sd
    d = np.abs(res["f32_sd"] ** 2 - res["f64_sd"] ** 2)
    print(name, {k: v for k, v in res.items() if not k.endswith("_sd")}, "max |dvar|/ys^2", float(d.max() / ys**2),
          "max rel sd err", float(np.max(np.abs(res["f32_sd"] - res["f64_sd"]) / res["f64_sd"])), flush=True)
