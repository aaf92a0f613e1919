"""End-to-end ms/suggest through the drop-in seams (FloatSpace + HipGPR + fused acquisition), C2 and C3.
Stages timed separately: host candidate sampling, fused random stage (fit + H2D + posterior + acq + arg-best),
the host L-BFGS-B "smart" stage over single-point device predicts, and the theta search with the LML on
the device.  Development/measurement aid; writes gpurun_out/suggest_latency.json."""
import json
import os
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
from bayesianoptimization_amd.gpr import HipGPR  # noqa: E402
from bayesianoptimization_amd.float_space import FloatSpace  # noqa: E402

warnings.simplefilter("ignore")
eng = GpEngine(0, debug=True)
out = {}
for name in ("C2", "C3"):
    w = W.ALL[name]
    X, y, c = W.make_observations(w)
    sp = FloatSpace(w.pbounds())
    sp.register_bulk(X, y)
    fn = A.UpperConfidenceBound(kappa=2.576) if w.acq == W.UCB else A.ExpectedImprovement(xi=w.acq_param)
    r = {}
    t0 = time.perf_counter(); sp.random_sample(w.M, np.random.RandomState(7)); r["host_random_sample_ms"] = (time.perf_counter() - t0) * 1e3
    gp = HipGPR(kernel=Matern(nu=2.5, length_scale=w.length_scale), alpha=w.noise, normalize_y=True, optimizer=None, engine=eng)
    for n_smart in (0, 10):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            x = fn.suggest(gp, sp, n_random=w.M, n_smart=n_smart, fit_gp=True, random_state=np.random.RandomState(7))
            ts.append((time.perf_counter() - t0) * 1e3)
        r[f"suggest_fixed_theta_nsmart{n_smart}_ms"] = ts
    fn.device_sampling = False   # host random_sample + upload (the path before the MT19937 device generator)
    for n_smart in (0, 10):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            x_host = fn.suggest(gp, sp, n_random=w.M, n_smart=n_smart, fit_gp=True, random_state=np.random.RandomState(7))
            ts.append((time.perf_counter() - t0) * 1e3)
        r[f"suggest_host_sampling_nsmart{n_smart}_ms"] = ts
    r["device_stream_equals_host_stream"] = bool(np.array_equal(x, x_host))
    fn.device_sampling = "auto"
    t0 = time.perf_counter()
    eng.generate_candidates_like(w.M, sp.bounds[:, 0], sp.bounds[:, 1], np.random.RandomState(7))
    r["device_mt19937_generation_ms"] = (time.perf_counter() - t0) * 1e3
    fn.lockstep = False          # one L-BFGS-B run after another (round-1 behaviour before the lockstep hub)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        x_seq = fn.suggest(gp, sp, n_random=w.M, n_smart=10, fit_gp=True, random_state=np.random.RandomState(7))
        ts.append((time.perf_counter() - t0) * 1e3)
    r["suggest_fixed_theta_nsmart10_sequential_runs_ms"] = ts
    fn.lockstep = True
    # bitwise only when one small-batch path is pinned (GPBO_SMALL_MAX); across the GEMV/MFMA switch: to rounding
    r["lockstep_vs_sequential_max_abs_diff"] = float(np.max(np.abs(x - x_seq)))
    fn.device_sampling = True   # throughput mode: Philox candidates generated on the device (non-parity)
    for n_smart in (0, 10):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            x = fn.suggest(gp, sp, n_random=w.M, n_smart=n_smart, fit_gp=True, random_state=np.random.RandomState(7))
            ts.append((time.perf_counter() - t0) * 1e3)
        r[f"suggest_device_sampling_nsmart{n_smart}_ms"] = ts
    fn.device_sampling = "auto"
    # single-point predict latency (what L-BFGS-B's finite differences call)
    xs = sp.random_sample(64, np.random.RandomState(1))
    t0 = time.perf_counter()
    for i in range(64):
        gp.predict(xs[i:i + 1], return_std=True)
    r["single_point_predict_ms"] = (time.perf_counter() - t0) * 1e3 / 64
    # theta search (default bayes_opt GP config: 5 restarts) with the LML on the device
    gp2 = HipGPR(kernel=Matern(nu=2.5), alpha=w.noise, normalize_y=True, n_restarts_optimizer=5,
                 random_state=np.random.RandomState(3), engine=eng, lml_on_device=True)
    n_eval = [0]
    orig = gp2.log_marginal_likelihood

    def counted(*a, **k):
        n_eval[0] += 1
        return orig(*a, **k)

    gp2.log_marginal_likelihood = counted
    t0 = time.perf_counter(); gp2.fit(X, y); r["fit_with_theta_search_device_lml_s"] = time.perf_counter() - t0
    r["lml_evaluations"] = n_eval[0]
    r["fitted_length_scale"] = float(gp2.kernel_.length_scale)
    out[name] = r
    print(name, r, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "suggest_latency.json"), "w"), indent=1)
