"""TEST INFRASTRUCTURE — generate golden vectors by running THE REFERENCE ITSELF in this container.

    python -m oracle.gen_golden [CASE ...]        (default: every case; C3 full-M takes ~10 min)

Drives bayes_opt 3.3.0 from /root/reference (via oracle/refenv.py) exactly as
BayesianOptimization.suggest() does (bayesian_optimization.py:323-333 -> acquisition.py:116-169):
the reference's own TargetSpace, wrapped kernel, GaussianProcessRegressor, `_fit_gp`, `_get_acq`
closure and `random_sample` produce every number stored.  Outputs: tests/golden/<CASE>.npz plus
tests/golden/MANIFEST.json (library versions, sizes, theta).  The GPU box has no /root/reference, so
these committed files are what the `-m gpu` parity tests compare against.
"""
from __future__ import annotations

import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bayesianoptimization_amd import workloads as W  # noqa: E402
from oracle.refenv import import_reference  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
SAMPLE = 8192  # per-candidate values stored for the first SAMPLE candidates
TOPK = 16
CAND_SEED = 7


def _chunk_for(N: int) -> int:
    return 8192 if N <= 4096 else 2048


def build_reference_objects(w: W.Workload):
    """Real reference objects for workload `w`, fitted. Returns (optimizer, acquisition fn)."""
    import_reference()
    from bayes_opt import BayesianOptimization, acquisition
    from bayes_opt.parameter import wrap_kernel
    from scipy.optimize import NonlinearConstraint
    from sklearn.gaussian_process.kernels import RBF, Matern

    if w.acq == W.UCB:
        fn = acquisition.UpperConfidenceBound(kappa=w.acq_param)
    elif w.acq == W.EI:
        fn = acquisition.ExpectedImprovement(xi=w.acq_param)
    else:
        fn = acquisition.ProbabilityOfImprovement(xi=w.acq_param)
    constraint = NonlinearConstraint(lambda **kw: 0.0, -np.inf, w.constraint_ub) if w.constrained else None
    opt = BayesianOptimization(f=None, pbounds=w.pbounds(), acquisition_function=fn, constraint=constraint,
                               random_state=np.random.RandomState(3), verbose=0, allow_duplicate_points=True)
    X, y, c = W.make_observations(w)
    space = opt._space
    if w.N <= 1024:
        for i in range(w.N):
            if c is None:
                opt.register(params=X[i], target=y[i])
            else:
                opt.register(params=X[i], target=y[i], constraint_value=c[i])
    else:  # bulk load (TargetSpace.register re-allocates per call)
        space._params = X.copy()
        space._target = y.copy()
        if c is not None:
            space._constraint_values = c.copy()
    assert np.array_equal(space.params, X) and np.array_equal(space.target, y)

    def mk(ls):
        return RBF(length_scale=ls) if w.kernel == W.RBF else Matern(nu=2.5, length_scale=ls)

    if w.length_scale is not None:
        opt.set_gp_params(kernel=mk(w.length_scale), optimizer=None)
    elif w.kernel == W.RBF:
        opt.set_gp_params(kernel=mk(1.0))
    if w.constrained and w.constraint_length_scale is not None:
        for m in space.constraint._model:
            m.set_params(kernel=wrap_kernel(Matern(nu=2.5, length_scale=w.constraint_length_scale),
                                            space.kernel_transform), optimizer=None)
    t0 = time.time()
    fn._fit_gp(opt._gp, space)
    fit_s = time.time() - t0
    if w.acq in (W.EI, W.POI):
        fn.y_max = space._target_max()
    return opt, fn, fit_s


def generate(w: W.Workload, full_m: bool = True) -> dict:
    opt, fn, fit_s = build_reference_objects(w)
    space, gp = opt._space, opt._gp
    M = w.M if full_m else min(w.M, SAMPLE)
    Xc = space.random_sample(M, np.random.RandomState(CAND_SEED))
    mine = W.make_candidates(w.bounds_array(), M, CAND_SEED)
    assert np.array_equal(Xc, mine), "workloads.make_candidates diverged from the reference stream"
    acq = fn._get_acq(gp=gp, constraint=space.constraint)
    chunk = _chunk_for(w.N)
    ys = np.empty(M)
    t0 = time.time()
    for s in range(0, M, chunk):
        ys[s:s + chunk] = acq(Xc[s:s + chunk])
    acq_s = time.time() - t0
    S = min(SAMPLE, M)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mu, sd = gp.predict(Xc[:S], return_std=True)
        out = {
            "length_scale": np.atleast_1d(gp.kernel_.length_scale).astype(np.float64),
            "y_mean": np.float64(gp._y_train_mean), "y_std": np.float64(gp._y_train_std),
            "alpha": gp.alpha_.copy(), "L_diag": np.diag(gp.L_).copy(),
            "L_lastrow": gp.L_[-1].copy(), "L_fro": np.float64(np.linalg.norm(gp.L_)),
            "mu": mu, "sd": sd, "ys": ys[:S].copy(),
            "argmin": np.int64(ys.argmin()), "min": np.float64(ys.min()),
            "topk_idx": np.argsort(ys)[:TOPK].astype(np.int64),
            "M_evaluated": np.int64(M),
            "y_max": np.float64(fn.y_max) if getattr(fn, "y_max", None) is not None else np.float64("nan"),
        }
        out["topk_val"] = ys[out["topk_idx"]].copy()
        if w.N <= 256:
            out["L"] = np.ascontiguousarray(gp.L_)
        if w.constrained:
            cm = space.constraint._model[0]
            cmu, csd = cm.predict(Xc[:S], return_std=True)
            out.update({
                "c_length_scale": np.atleast_1d(cm.kernel_.length_scale).astype(np.float64),
                "c_y_mean": np.float64(cm._y_train_mean), "c_y_std": np.float64(cm._y_train_std),
                "c_alpha": cm.alpha_.copy(), "c_mu": cmu, "c_sd": csd,
                "p_c": space.constraint.predict(Xc[:S]),
            })
    # Seam B1 end-to-end with the random stage only (n_smart=0): x = Xc[argmin].
    x_suggest = fn.suggest(gp=gp, target_space=space, n_random=min(M, 1 << 16), n_smart=0, fit_gp=False,
                           random_state=np.random.RandomState(CAND_SEED))
    out["suggest_nsmart0_x"] = np.asarray(x_suggest)
    out["suggest_nsmart0_nrandom"] = np.int64(min(M, 1 << 16))
    meta = {"N": w.N, "d": w.d, "M_evaluated": int(M), "ref_fit_s": round(fit_s, 3),
            "ref_acq_s": round(acq_s, 3), "length_scale": out["length_scale"].tolist(),
            "argmin": int(out["argmin"]), "min": float(out["min"]),
            "top2_gap": float(out["topk_val"][1] - out["topk_val"][0])}
    return out, meta


def main(argv):
    import scipy
    import sklearn

    cases = argv or ["C1", "F1", "P1", "P2", "C5S", "C2", "C3", "C5"]
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    mpath = os.path.join(GOLDEN_DIR, "MANIFEST.json")
    manifest = json.load(open(mpath)) if os.path.exists(mpath) else {}
    manifest["_versions"] = {"bayes_opt": "3.3.0", "sklearn": sklearn.__version__, "scipy": scipy.__version__,
                             "numpy": np.__version__, "candidate_seed": CAND_SEED, "sample": SAMPLE}
    for name in cases:
        w = W.ALL[name]
        full = name != "C5"  # C5 (N=8192, 2 GPs, M=2^21) is sampled: the full pass costs ~40 min on 8 cores
        t0 = time.time()
        out, meta = generate(w, full_m=full)
        np.savez_compressed(os.path.join(GOLDEN_DIR, f"{name}.npz"), **out)
        meta["gen_wall_s"] = round(time.time() - t0, 1)
        manifest[name] = meta
        print(name, meta, flush=True)
        json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
