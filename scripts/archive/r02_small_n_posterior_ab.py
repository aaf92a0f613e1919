import os
import sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bayesianoptimization_amd import workloads as W
from bayesianoptimization_amd.engine import GpEngine
eng = GpEngine(0, debug=True)
for name in ("C2",):
    w = W.ALL[name]
    X, y, c = W.make_observations(w)
    Xc = W.make_candidates(w.bounds_array(), w.M, 7)
    yn, ym, ys = W.normalize_targets(y)
    eng.fit(X, yn, w.kernel, w.length_scale, w.noise, slot=0)
    eng.set_candidates(Xc)
    ref = None
    for kv in ("2", "3", "2", "3"):
        os.environ["GPBO_POST_KERNEL"] = kv
        ts = []
        for _ in range(20):
            eng.posterior(0, ym, ys, fetch=False)
            ts.append(eng.last_timings()["posterior_main"])
        mu, sd = eng.posterior(0, ym, ys)
        if ref is None: ref = (mu, sd)
        print(name, "kernel", kv, "posterior_main ms min/median", min(ts), float(np.median(ts)), "max|dmu|", float(np.max(np.abs(mu-ref[0]))), float(np.max(np.abs(sd-ref[1]))))
for N, d, M in ((1024, 8, 65536), (512, 8, 262144), (256, 4, 65536)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(N, d)); y = np.sin(3*X.sum(1)); yn = (y-y.mean())/y.std()
    eng.fit(X, yn, W.MATERN25, 1.0, 1e-6, slot=0)
    eng.set_candidates(rng.uniform(size=(M, d)))
    for kv in ("2", "3"):
        os.environ["GPBO_POST_KERNEL"] = kv
        ts = []
        for _ in range(10):
            eng.posterior(0, 0.0, 1.0, fetch=False); ts.append(eng.last_timings()["posterior_main"])
        print("N", N, "M", M, "kernel", kv, min(ts))
