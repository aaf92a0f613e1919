"""Round-2: ms/suggest through the seams with the local search in its two gradient modes — the reference's finite
differences (d + 1 points per evaluation, batched, lockstep; bit-parity mode) and the analytic gradient
(gpbo_predict_grad, 1 point per evaluation) — and the acquisition value each ends at.
C2 and C3, unconstrained, plus the constrained C5S shape.  Writes gpurun_out/r02_suggest_modes.json."""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sklearn.gaussian_process.kernels import Matern  # noqa: E402

from bayesianoptimization_amd import fused_acquisition as A  # noqa: E402
from bayesianoptimization_amd import workloads as W  # noqa: E402
from bayesianoptimization_amd.constraint_model import HipConstraintModel  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402
from bayesianoptimization_amd.float_space import FloatSpace  # noqa: E402
from bayesianoptimization_amd.gpr import HipGPR  # noqa: E402

warnings.simplefilter("ignore")
eng = GpEngine(0)
out = {}
for name in ("C2", "C3", "C5S"):
    w = W.ALL[name]
    X, y, c = W.make_observations(w)
    cons = None
    if w.constrained:
        cons = HipConstraintModel(None, -np.inf, w.constraint_ub, engine=eng)
        cons._model[0].set_params(kernel=Matern(nu=2.5, length_scale=w.constraint_length_scale), optimizer=None)
    sp = FloatSpace(w.pbounds(), constraint=cons)
    sp.register_bulk(X, y, c)
    gp = HipGPR(kernel=Matern(nu=2.5, length_scale=w.length_scale), alpha=w.noise, normalize_y=True, optimizer=None, engine=eng)
    r = {}
    for mode in ("fd", "analytic"):
        fn = A.UpperConfidenceBound(kappa=2.576) if w.acq == W.UCB else A.ExpectedImprovement(xi=w.acq_param)
        fn.analytic_gradient = (mode == "analytic")
        ts, vals = [], []
        for rep in range(5):
            t0 = time.perf_counter()
            x = fn.suggest(gp, sp, n_random=w.M, n_smart=10, fit_gp=True, random_state=np.random.RandomState(7 + rep))
            ts.append((time.perf_counter() - t0) * 1e3)
            if w.acq != W.UCB:
                fn.y_max = sp._target_max()
            vals.append(float(fn._get_acq(gp, sp.constraint)(x[None])[0]))
        r[mode] = {"ms": ts, "median_ms": float(np.median(ts[1:])), "neg_acq_at_suggestion": vals}
    ts0 = []
    fn = A.UpperConfidenceBound(kappa=2.576) if w.acq == W.UCB else A.ExpectedImprovement(xi=w.acq_param)
    for rep in range(4):
        t0 = time.perf_counter()
        fn.suggest(gp, sp, n_random=w.M, n_smart=0, fit_gp=True, random_state=np.random.RandomState(7 + rep))
        ts0.append((time.perf_counter() - t0) * 1e3)
    r["n_smart_0_median_ms"] = float(np.median(ts0[1:]))
    r["analytic_at_least_as_good"] = [a <= f + 1e-9 * max(1.0, abs(f)) for a, f in
                                      zip(r["analytic"]["neg_acq_at_suggestion"], r["fd"]["neg_acq_at_suggestion"])]
    out[name] = r
    print(name, json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r02_suggest_modes.json"), "w"), indent=1)
