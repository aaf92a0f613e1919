"""Round 4 (VERDICT r3 #1c): where does the device LML overtake scikit-learn's on the host?

`HipGPR(lml_on_device="auto")` sent the theta search's log-marginal-likelihood evaluations to the device from N >= 512
only, a threshold no measurement backed.  Per N in 16 .. 1024 (d = 8, Matern-2.5, default bayes_opt GP configuration):
  * one value + gradient: sklearn's `log_marginal_likelihood(theta, eval_gradient=True)` on the host (_gpr.py:575-652; the
    box's BLAS threading as it comes, and pinned to 1 / 8 threads) vs `gpbo_lml` vs one lane of a 6-lane `gpbo_lml_batch`;
  * the whole `fit()` with 5 restarts (what `BayesianOptimization.suggest()` pays per call): host LML vs device LML in lockstep.
-> gpurun_out/r04_lml_crossover.json"""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sklearn.gaussian_process import GaussianProcessRegressor  # noqa: E402
from sklearn.gaussian_process.kernels import Matern  # noqa: E402

from bayesianoptimization_amd.engine import MATERN25, GpEngine  # noqa: E402
from bayesianoptimization_amd.gpr import HipGPR  # noqa: E402

warnings.simplefilter("ignore")
try:
    from threadpoolctl import threadpool_limits
except Exception:  # noqa: BLE001
    threadpool_limits = None

eng = GpEngine(0)
out = {"host": {"cpu_count": os.cpu_count()}, "N": {}}


def med(f, reps=7):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


for N in (16, 32, 64, 96, 128, 192, 256, 384, 512, 768, 1024):
    d = 8
    rng = np.random.RandomState(N)
    X = rng.uniform(size=(N, d))
    y = np.exp(-((X - 0.5) ** 2).sum(1)) + 0.01 * rng.standard_normal(N)
    yn = (y - y.mean()) / y.std()
    sk = GaussianProcessRegressor(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, optimizer=None).fit(X, y)
    theta = np.log([0.7])
    r = {}
    r["host_lml_ms_default_threads"] = med(lambda: sk.log_marginal_likelihood(theta, eval_gradient=True))
    if threadpool_limits is not None:
        for nt in (1, 8):
            with threadpool_limits(limits=nt):
                r[f"host_lml_ms_{nt}_threads"] = med(lambda: sk.log_marginal_likelihood(theta, eval_gradient=True))
    r["device_lml_ms"] = med(lambda: eng.lml(X, yn, MATERN25, 0.7, 1e-6))
    scales = np.array([[0.5], [0.8], [1.0], [1.5], [2.0], [3.0]])
    eng.lml_batch(X, yn, MATERN25, scales, 1e-6)
    r["device_lml_batch6_ms"] = med(lambda: eng.lml_batch(X, yn, MATERN25, scales, 1e-6, reuse_inputs=True))
    r["device_lml_batch2_ms"] = med(lambda: eng.lml_batch(X, yn, MATERN25, scales[:2], 1e-6, reuse_inputs=True))
    r["device_lml_batch1_ms"] = med(lambda: eng.lml_batch(X, yn, MATERN25, scales[:1], 1e-6, reuse_inputs=True))
    for name, on_dev in (("fit_host_lml_ms", False), ("fit_device_lml_ms", True)):
        def fit(on_dev=on_dev):
            gp = HipGPR(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5,
                        random_state=np.random.RandomState(3), engine=eng, lml_on_device=on_dev, incremental=False)
            gp.fit(X, y)
            return gp
        r[name] = med(fit, reps=5)
    g_h = HipGPR(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5, random_state=np.random.RandomState(3),
                 engine=eng, lml_on_device=False).fit(X, y)
    g_d = HipGPR(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5, random_state=np.random.RandomState(3),
                 engine=eng, lml_on_device=True).fit(X, y)
    r["lml_opt_host"], r["lml_opt_device"] = float(g_h.log_marginal_likelihood_value_), float(g_d.log_marginal_likelihood_value_)
    r["theta_host"], r["theta_device"] = float(g_h.kernel_.theta[0]), float(g_d.kernel_.theta[0])
    out["N"][str(N)] = r
    print(N, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_lml_crossover.json"), "w"), indent=1)
