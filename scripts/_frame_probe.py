import os, sys, json, warnings
import numpy as np
sys.path.insert(0, os.getcwd())
from sklearn.gaussian_process.kernels import Matern
from bayesianoptimization_amd.engine import GpEngine
from bayesianoptimization_amd.gpr import HipGPR
warnings.simplefilter("ignore")
d = json.load(open("gpurun_out/r06_maximize_loop.json")) if os.path.exists("gpurun_out/r06_maximize_loop.json") else None
eng = GpEngine(0)
def black_box(x):
    x = np.asarray(x, dtype=np.float64)
    return float(-np.sum((x - 0.3) ** 2) + 0.5 * np.sin(5.0 * x[0]) * np.cos(3.0 * x[1]))
for N in (200, 300, 400):
    rng = np.random.RandomState(N)
    # clustered points like a BO run: half uniform, half near the optimum
    X = np.vstack([rng.uniform(size=(N // 3, 4)), np.clip(0.3 + 0.05 * rng.standard_normal((N - N // 3, 4)), 0, 1)])
    y = np.array([black_box(x) for x in X])
    for use_frame in (True, False):
        gp = HipGPR(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5, random_state=np.random.RandomState(1), engine=eng)
        if not use_frame:
            eng.__class__ = type("GpEngineNoFrame", (GpEngine,), {})
        else:
            eng.__class__ = GpEngine
        gp.fit(X, y)
        print(N, "frame" if use_frame else "arrays", gp.theta_search_rounds_, gp.theta_search_evals_, float(np.exp(gp.kernel_.theta[0])), gp.log_marginal_likelihood_value_)
eng.__class__ = GpEngine
