"""Within-process interleaved A/B of the posterior paths (GPBO_POST_KERNEL=2 fused | 3 slab+GEMM)
plus fit / LML timings, on C3 (and C2 for the latency regime).  Development aid; writes gpurun_out/ab.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd import workloads as W  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

eng = GpEngine(0, debug=True)
out = {}
print(out, flush=True)
for name, M in (("C3", 1 << 20), ("C2", 1 << 16)):
    w = W.ALL[name]
    X, y, c = W.make_observations(w)
    ym, ys = float(np.mean(y)), float(np.std(y))
    yn = (y - ym) / ys
    fits = []
    for _ in range(3):
        eng.fit(X, yn, w.kernel, w.length_scale, w.noise)
        fits.append(eng.last_timings())
    out[name + "_fit_ms"] = fits[-1]
    import time
    lm = []
    for _ in range(3):
        t0 = time.perf_counter()
        val, grad = eng.lml(X, yn, w.kernel, w.length_scale, w.noise)
        lm.append((time.perf_counter() - t0) * 1e3)
    out[name + "_lml_wall_ms"] = lm
    out[name + "_lml"] = [val, grad.tolist()]
    if name == "C3":
        from sklearn.gaussian_process import GaussianProcessRegressor
        from sklearn.gaussian_process.kernels import Matern
        sk = GaussianProcessRegressor(kernel=Matern(nu=2.5, length_scale=w.length_scale), alpha=w.noise,
                                      normalize_y=True, optimizer=None).fit(X, y)
        t0 = time.perf_counter()
        v_s, g_s = sk.log_marginal_likelihood(sk.kernel_.theta, eval_gradient=True)
        out["C3_lml_sklearn_s"] = time.perf_counter() - t0
        out["C3_lml_sklearn"] = [float(v_s), g_s.tolist()]
    eng.fit(X, yn, w.kernel, w.length_scale, w.noise)
    eng.set_candidates(W.make_candidates(w.bounds_array(), M, 7))
    res = {"v2": [], "v3": []}
    ref = None
    for rnd in range(4):
        for v, env in (("v2", {"GPBO_POST_KERNEL": "2"}), ("v3", {"GPBO_POST_KERNEL": "3"})):
            os.environ.update(env)
            mu, sd = eng.posterior(0, ym, ys)
            res[v].append(eng.last_timings()["posterior_main"])
            if ref is None:
                ref = (mu.copy(), sd.copy())
            res[v + "_maxdiff"] = float(max(np.max(np.abs(mu - ref[0])), np.max(np.abs(sd - ref[1]))))
    out[name + "_post_ms"] = res
    print(name, out[name + "_fit_ms"], res, flush=True)
os.environ.pop("GPBO_POST_SCHED", None)
os.environ.pop("GPBO_POST_KERNEL", None)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ab.json"), "w"), indent=1)
