"""theta search (default bayes_opt GP config: L-BFGS-B + 5 restarts) with the LML on the device: the six runs one
after another (gpbo_lml) against advanced in lockstep (gpbo_lml_batch), plus the raw batch-of-6 evaluation time."""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sklearn.gaussian_process.kernels import Matern  # noqa: E402

from bayesianoptimization_amd.engine import MATERN25, GpEngine  # noqa: E402
from bayesianoptimization_amd.gpr import HipGPR  # noqa: E402

warnings.simplefilter("ignore")
eng = GpEngine(0)
out = {}
for N, d in ((512, 8), (1024, 16), (2048, 16), (4096, 16)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(N, d))
    y = np.exp(-((X - 0.5) ** 2).sum(1)) + 0.01 * rng.standard_normal(N)   # smooth target: an interior optimum
    yn = (y - y.mean()) / y.std()
    r = {}
    scales = np.array([[0.5], [0.8], [1.0], [1.5], [2.0], [3.0]])
    for _ in range(2):
        eng.lml_batch(X, yn, MATERN25, scales, 1e-6)
        [eng.lml(X, yn, MATERN25, s, 1e-6) for s in scales]
    t0 = time.perf_counter(); eng.lml_batch(X, yn, MATERN25, scales, 1e-6); r["lml_batch6_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); [eng.lml(X, yn, MATERN25, s, 1e-6) for s in scales]; r["lml_x6_ms"] = (time.perf_counter() - t0) * 1e3
    res = {}
    rounds = []
    orig_frame = eng.lml_search_rounds        # (what a theta search calls: one frame per search, one call per round)

    def timed_frame(*a, **k):
        one_round = orig_frame(*a, **k)

        def timed_round(scales_):
            t1 = time.perf_counter()
            out_ = one_round(scales_)
            rounds.append((len(out_[0]), round((time.perf_counter() - t1) * 1e3, 3)))
            return out_

        return timed_round

    eng.lml_search_rounds = timed_frame
    lock_s = []
    for lockstep in (True, True, True, False):          # the first search at a shape also captures its lane groups' graphs
        gp = HipGPR(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5,
                    random_state=np.random.RandomState(3), engine=eng, lml_on_device=True, theta_lockstep=lockstep)
        if lockstep:
            rounds.clear()
        t0 = time.perf_counter()
        n_eval = [0]
        if not lockstep:
            orig = gp.log_marginal_likelihood

            def counted(*a, _o=orig, **k):
                n_eval[0] += 1
                return _o(*a, **k)

            gp.log_marginal_likelihood = counted
        gp.fit(X, y)
        res[lockstep] = (time.perf_counter() - t0, gp.kernel_.theta.copy())
        if lockstep:
            lock_s.append(res[True][0])
        if not lockstep:
            r["lml_evaluations"] = n_eval[0]
    del eng.lml_search_rounds
    r["fit_theta_search_lockstep_first_s"] = lock_s[0]
    r["lockstep_rounds_lanes_ms"] = list(rounds)        # of the last lockstep search: (live runs, ms) per round
    r["fit_theta_search_lockstep_s"] = res[True][0]
    r["fit_theta_search_sequential_s"] = res[False][0]
    r["same_theta"] = bool(np.array_equal(res[True][1], res[False][1]))
    r["length_scale"] = float(np.exp(res[True][1][0]))
    out[str(N)] = r
    print(N, r, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "theta_search_timing.json"), "w"), indent=1)
