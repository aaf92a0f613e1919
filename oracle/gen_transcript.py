"""Call transcripts of the REAL driver over the engine seam (test infrastructure; runs in the build container only).

    python oracle/gen_transcript.py            -> tests/golden/transcript_*.npz

What this closes: the GPU box has no /root/reference, so `bayes_opt`'s own `BayesianOptimization.maximize()` never runs over
`libgpbo.so` in one process (VERDICT r5 missing #2).  Here the real `bayes_opt` 3.3.0 (bayesian_optimization.py:124-130, 323-333,
348-391; target_space.py:565-603) drives `accelerate(optimizer)` over a RECORDING engine — tests/helpers.FakeEngine, the CPU oracle
behind the GpEngine surface — and every engine call the driver makes is stored in order: method name, every argument (arrays with
dtype / shape / memory order, RandomState arguments as their MT19937 state before and after the call) and what the oracle returned.
tests/test_gpu_transcript.py replays each file call by call on the real library, on the GPU, and holds every return value to the
bar of its kind.  The replay is open loop — each call gets the RECORDED inputs — so one deviating suggestion cannot hide behind
a trajectory that diverged with it.

Five drivers: all-float UCB with the default theta search in every fit (30 steps), constrained EI, a mixed float / int /
categorical space, GPHedge over UCB / EI / POI, ConstantLiar.  A fixture is data: inputs and expected outputs, no source text.
"""
from __future__ import annotations

import hashlib
import inspect
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from helpers import FakeEngine  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402
from oracle.refenv import import_reference  # noqa: E402

FORMAT = 1
#: engine methods that are recorded (everything HipGPR / the fused acquisition classes / HipConstraintModel call)
RECORDED = ("fit", "fit_append", "lml", "lml_batch", "get_L", "get_alpha", "set_candidates", "generate_candidates_like",
            "get_candidate_rows", "posterior", "predict", "predict_cov", "predict_grad", "polish_seeds", "acq_argbest",
            "take_negative_variance_flag")
#: of a run of consecutive `predict` calls (a host optimiser's objective: differential evolution makes thousands) only the first
#: PREDICT_RUN_CAP are stored; the call is stateless apart from the resident candidates, which the next stage replaces
PREDICT_RUN_CAP = 48
SAMPLE = 256      # posterior(fetch=False): mu / sd of the resident candidates are stored at this many indices


class Pool:
    """Arrays stored once by content (the theta search passes the same X to every round)."""

    def __init__(self):
        self.arrays = {}

    def put(self, a) -> dict:
        a = np.asarray(a)
        order = "C" if a.flags.c_contiguous else ("F" if a.flags.f_contiguous else "strided")
        c = np.ascontiguousarray(a)
        key = "a" + hashlib.sha1(c.tobytes() + str((c.dtype.str, c.shape)).encode()).hexdigest()[:16]
        self.arrays.setdefault(key, c)
        return {"ref": key, "dtype": a.dtype.str, "shape": list(a.shape), "order": order}


def _rng_state(rs):
    st = rs.get_state(legacy=True)
    return {"key": np.asarray(st[1], dtype=np.uint32), "pos": int(st[2]), "has_gauss": int(st[3]), "cached": float(st[4])}


def encode(v, pool: Pool):
    if v is None or isinstance(v, (bool, str)):
        return v
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        return {"f": float(v).hex()}       # exact
    if isinstance(v, np.random.RandomState):
        s = _rng_state(v)
        return {"rng": pool.put(s["key"]), "pos": s["pos"], "has_gauss": s["has_gauss"], "cached": float(s["cached"]).hex()}
    if isinstance(v, np.ndarray):
        return pool.put(v)
    if isinstance(v, (list, tuple)):
        return {"seq": [encode(x, pool) for x in v], "tuple": isinstance(v, tuple)}
    raise TypeError(f"cannot record {type(v)!r}")


class RecordingEngine(FakeEngine):
    def __init__(self):
        super().__init__()
        self.pool = Pool()
        self.log = []
        self._depth = 0
        self._predict_run = 0
        self.skipped_predicts = 0

    def _extra(self, name, bound, ret):
        """What the oracle holds after the call that the call itself does not return."""
        ex = {}
        if name in ("fit", "fit_append"):
            gp = self.models[bound.get("slot", 0)]
            L = np.asarray(gp.L)
            n = L.shape[0]
            ex["N"] = n
            ex["L_diag"] = self.pool.put(np.diag(L).copy())
            ex["L_lastrow"] = self.pool.put(L[n - 1].copy())
            ex["L_fro"] = {"f": float(np.linalg.norm(L)).hex()}
            if n <= 48:
                ex["L"] = self.pool.put(L)
            ex["alpha"] = self.pool.put(np.asarray(gp.alpha).ravel().copy())
            sv = np.linalg.svd(L, compute_uv=False)
            ex["kappa"] = float((sv.max() / sv.min()) ** 2)      # cond_2(K): the replay's bars scale with it above 1e6
        elif name in ("lml", "lml_batch"):
            # conditioning of every theta asked for, cond_2(K): the replay's bars scale with it above 1e6 (at the upper bound of
            # the length scale K is all ones + 1e-6 I: kappa = N / 1e-6, and y^T K^-1 y is held to kappa * eps, not to eps)
            kap = []
            for ls in np.atleast_2d(np.asarray(bound["length_scales" if name == "lml_batch" else "length_scale"], dtype=np.float64)):
                K = O.kernel_matrix(bound["kernel"], np.asarray(bound["X"], dtype=np.float64), None, np.atleast_1d(ls))
                K[np.diag_indices_from(K)] += bound["noise"]
                ev = np.linalg.eigvalsh(K)
                kap.append(float(ev[-1] / ev[0]) if ev[0] > 0 else float("inf"))
            ex["kappa"] = kap
        elif name == "posterior" and not bound.get("fetch", True):
            mu, sd = self.post[bound.get("slot", 0)]
            M = mu.shape[0]
            idx = np.unique(np.concatenate([np.arange(min(64, M)), np.linspace(0, M - 1, min(SAMPLE - 64, M)).astype(np.int64)]))
            ex["sample_idx"] = self.pool.put(idx)
            ex["sample_mu"] = self.pool.put(mu[idx])
            ex["sample_sd"] = self.pool.put(sd[idx])
            ex["mu_absmax"] = {"f": float(np.max(np.abs(mu))).hex()}
            ex["sd_absmax"] = {"f": float(np.max(np.abs(sd))).hex()}
        elif name == "generate_candidates_like":
            ex["M"] = int(self.Xc.shape[0])
            ex["d"] = int(self.Xc.shape[1])
            ex["checksum"] = hashlib.sha1(np.ascontiguousarray(self.Xc).tobytes()).hexdigest()
            ex["first_rows"] = self.pool.put(self.Xc[:4].copy())
        elif name == "acq_argbest":
            # the k + 2 smallest values in the reference's order: the replay asserts exact indices where their gaps allow it
            mu, sd = self.post[0]
            ys = self._last_ys
            nan = np.isnan(ys)
            order = np.lexsort((np.arange(len(ys)), np.where(nan, np.inf, ys) + 0.0, nan))[:bound.get("k_seeds", 0) + 2]
            ex["head_idx"] = self.pool.put(order.astype(np.int64))
            ex["head_val"] = self.pool.put(ys[order])
            ex["n_nan"] = int(nan.sum())
            ex["range"] = {"f": float(np.nanmax(np.abs(ys)) if not nan.all() else 0.0).hex()}
        return ex

    def acq_argbest(self, acq, param, y_max=0.0, lb=None, ub=None, k_seeds=0, index_offset=0, return_values=False):
        out = FakeEngine.acq_argbest(self, acq, param, y_max, lb, ub, k_seeds, index_offset, True)
        self._last_ys = out[4]
        return out[:4] + ((out[4] if return_values else None),)


def _wrap(name):
    base = getattr(FakeEngine, name) if name != "acq_argbest" else RecordingEngine.acq_argbest
    sig = inspect.signature(base)

    def method(self, *a, **k):
        top = self._depth == 0
        rec = None
        if top:
            if name == "predict":
                self._predict_run += 1
            else:
                self._predict_run = 0
            if name != "predict" or self._predict_run <= PREDICT_RUN_CAP:
                b = sig.bind(self, *a, **k)
                b.apply_defaults()
                bound = {kk: vv for kk, vv in b.arguments.items() if kk != "self"}
                rec = {"name": name, "args": {kk: encode(vv, self.pool) for kk, vv in bound.items()}}
            else:
                self.skipped_predicts += 1
        self._depth += 1
        try:
            ret = base(self, *a, **k)
        finally:
            self._depth -= 1
        if rec is not None:
            rec["ret"] = encode(ret, self.pool)
            rec["rng_after"] = {kk: encode(vv, self.pool) for kk, vv in bound.items() if isinstance(vv, np.random.RandomState)}
            rec["extra"] = self._extra(name, bound, ret)
            self.log.append(rec)
        return ret

    method.__name__ = name
    return method


for _n in RECORDED:
    setattr(RecordingEngine, _n, _wrap(_n))


def save(path, eng: RecordingEngine, meta: dict):
    meta = dict(meta, format=FORMAT, n_calls=len(eng.log), skipped_predict_calls=eng.skipped_predicts,
                versions={"numpy": np.__version__, "scipy": __import__("scipy").__version__,
                          "sklearn": __import__("sklearn").__version__, "bayes_opt": "3.3.0"})
    np.savez_compressed(path, __calls__=np.frombuffer(json.dumps(eng.log).encode(), dtype=np.uint8),
                        __meta__=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **eng.pool.arrays)
    kinds = {}
    for c in eng.log:
        kinds[c["name"]] = kinds.get(c["name"], 0) + 1
    print(f"{os.path.basename(path)}: {len(eng.log)} calls {kinds}, {os.path.getsize(path) / 1024:.0f} KiB")


# ---- the five drivers ----------------------------------------------------------------------------------------------------
def _accelerated(f, pbounds, seed, n_random, constraint=None, acq=None):
    import_reference()
    from bayes_opt import BayesianOptimization

    from bayesianoptimization_amd import accelerate

    opt = BayesianOptimization(f=f, pbounds=pbounds, random_state=seed, verbose=0, constraint=constraint, acquisition_function=acq,
                               allow_duplicate_points=True)
    eng = RecordingEngine()
    accelerate(opt, engine=eng, n_random=n_random)      # the default configuration: theta search on the engine, one-call local searches
    return opt, eng


def float_ucb():
    def f(x, y, z):
        return -(x - 0.3) ** 2 - (y + 0.2) ** 2 - 0.5 * (z - 0.6) ** 2 + 0.3 * np.sin(4.0 * x) * np.cos(3.0 * y)

    opt, eng = _accelerated(f, {"x": (-1.0, 1.0), "y": (-1.0, 1.0), "z": (0.0, 1.0)}, seed=11, n_random=4096)
    opt.maximize(init_points=4, n_iter=30)
    return eng, {"driver": "BayesianOptimization.maximize(init_points=4, n_iter=30), default UCB, default GP (theta search, 5 restarts)",
                 "n_random": 4096, "steps": 30}


def constrained_ei():
    from scipy.optimize import NonlinearConstraint

    def f(x, y):
        return np.cos(2 * x) * np.cos(y) + np.sin(x)

    def c(x, y):
        return np.cos(x) * np.cos(y) - np.sin(x) * np.sin(y)

    opt, eng = _accelerated(f, {"x": (0.0, 6.0), "y": (0.0, 6.0)}, seed=5, n_random=2048,
                            constraint=NonlinearConstraint(c, -np.inf, 0.5))
    opt.maximize(init_points=5, n_iter=12)
    return eng, {"driver": "maximize(init_points=5, n_iter=12) with NonlinearConstraint(c, -inf, 0.5): bayes_opt picks EI; two GPs",
                 "n_random": 2048, "steps": 12}


def mixed_space():
    def f(x, k, c):
        return -(x - 0.4) ** 2 - 0.05 * (k - 3) ** 2 + {"a": 0.0, "b": 0.3, "c": -0.2}[c]

    opt, eng = _accelerated(f, {"x": (0.0, 1.0), "k": (0, 6, int), "c": ("a", "b", "c")}, seed=3, n_random=512)
    opt.maximize(init_points=12, n_iter=5)
    return eng, {"driver": "maximize(init_points=12, n_iter=5) over float x int x 3-way categorical (width 5 in kernel space); "
                           "host-sampled candidates, differential evolution on the host over the engine's predict",
                 "n_random": 512, "steps": 5}


def gphedge():
    import_reference()
    from bayes_opt import acquisition as RA

    def f(x, y):
        return -(x - 2.5) ** 2 - (y - 0.5) ** 2 + 1.0

    acq = RA.GPHedge(base_acquisitions=[RA.UpperConfidenceBound(kappa=2.576), RA.ExpectedImprovement(xi=0.01),
                                        RA.ProbabilityOfImprovement(xi=0.01)])
    opt, eng = _accelerated(f, {"x": (2.0, 4.0), "y": (-3.0, 3.0)}, seed=8, n_random=3072, acq=acq)
    opt.maximize(init_points=3, n_iter=8)
    return eng, {"driver": "maximize(init_points=3, n_iter=8) with GPHedge(UCB, EI, POI)", "n_random": 3072, "steps": 8}


def constant_liar():
    import_reference()
    from bayes_opt import acquisition as RA

    def f(x, y):
        return -(x - 2.5) ** 2 - (y - 0.5) ** 2 + 1.0

    acq = RA.ConstantLiar(base_acquisition=RA.UpperConfidenceBound(kappa=2.576), strategy="max")
    opt, eng = _accelerated(f, {"x": (2.0, 4.0), "y": (-3.0, 3.0)}, seed=9, n_random=2048, acq=acq)
    for _ in range(3):
        opt.probe(opt.space.random_sample(random_state=opt._random_state), lazy=False)
    # ConstantLiar's point: several suggestions before any of them is evaluated (acquisition.py:1037-1143)
    for _ in range(3):
        batch = [opt.suggest() for _ in range(3)]
        for x in batch:
            opt.probe(x, lazy=False)
    return eng, {"driver": "3 rounds of [suggest() x 3 without registering, then probe all 3] with ConstantLiar(UCB, 'max')",
                 "n_random": 2048, "steps": 9}


DRIVERS = {"float_ucb": float_ucb, "constrained_ei": constrained_ei, "mixed_space": mixed_space, "gphedge": gphedge,
           "constant_liar": constant_liar}


def generate(outdir=None, only=None):
    outdir = outdir or os.path.join(ROOT, "tests", "golden")
    for name, fn in DRIVERS.items():
        if only and name not in only:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            eng, meta = fn()
        save(os.path.join(outdir, f"transcript_{name}.npz"), eng, dict(meta, name=name))


if __name__ == "__main__":
    generate(only=set(sys.argv[1:]) or None)
