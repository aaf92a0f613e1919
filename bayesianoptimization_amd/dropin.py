"""accelerate(optimizer) — install the HIP engine behind a bayes_opt.BayesianOptimization instance.

Seams B1 + B2: replaces `optimizer._gp` (bayes_opt/bayesian_optimization.py:124-130)
and each constraint GP `optimizer._space._constraint._model[j]` (bayes_opt/constraint.py:72-81) by a
`HipGPR` with the same hyper-parameters and the SAME RandomState object (so the stream is consumed
exactly as before), and replaces a stock UCB/EI/POI acquisition function — also the ones a GPHedge or ConstantLiar
meta-policy delegates to — by its fused counterpart.
Everything else in the optimizer (space, queue, logging, state I/O) is untouched reference code.
"""
from __future__ import annotations

import warnings

import numpy as np

from . import fused_acquisition as A
from ._lib import MAX_DIM
from .gpr import HipGPR, describe_kernel, shared_engine


def _note_unsupported(kernel, what: str) -> str | None:
    """One UserWarning when a model's kernel is outside the device path: the model is swapped all the same (so that a later
    `set_gp_params(kernel=...)` with a supported kernel puts it on the GPU) and runs scikit-learn's code until then."""
    try:
        describe_kernel(kernel)
    except NotImplementedError as exc:
        warnings.warn(f"accelerate(): {what}: {exc}; it keeps running scikit-learn's GaussianProcessRegressor on the host "
                      "(the reference's path) until its kernel is one the HIP engine evaluates", UserWarning, stacklevel=3)
        return str(exc)
    return None


def _identity_transform(space) -> bool:
    cfg = getattr(space, "_params_config", None)
    if cfg is None:
        return True
    return all(type(p).__name__ == "FloatParameter" for p in cfg.values())


def _convert_acquisition(fn):
    name = type(fn).__name__
    if isinstance(fn, A.AcquisitionFunction):
        return fn
    if name == "UpperConfidenceBound":
        new = A.UpperConfidenceBound(kappa=fn.kappa, exploration_decay=fn.exploration_decay,
                                     exploration_decay_delay=fn.exploration_decay_delay)
    elif name == "ExpectedImprovement":
        new = A.ExpectedImprovement(xi=fn.xi, exploration_decay=fn.exploration_decay,
                                    exploration_decay_delay=fn.exploration_decay_delay)
        new.y_max = fn.y_max
    elif name == "ProbabilityOfImprovement":
        new = A.ProbabilityOfImprovement(xi=fn.xi, exploration_decay=fn.exploration_decay,
                                         exploration_decay_delay=fn.exploration_decay_delay)
        new.y_max = fn.y_max
    elif name == "GPHedge" and hasattr(fn, "base_acquisitions"):
        # the meta policy stays reference code (bayes_opt/acquisition.py:1181-1360); the policies it delegates the
        # random + local searches to (`base.suggest(..., fit_gp=False)`, :1306-1316) become the fused ones
        fn.base_acquisitions = [_convert_acquisition(b) for b in fn.base_acquisitions]
        return fn
    elif name == "ConstantLiar" and hasattr(fn, "base_acquisition"):
        fn.base_acquisition = _convert_acquisition(fn.base_acquisition)    # acquisition.py:1135-1143
        return fn
    else:
        return fn  # custom acquisitions keep running their own code over HipGPR.predict
    new.i = fn.i
    return new


#: observation counts of the warm-up problems: one per dispatch band of a small-N suggest() — the one-workgroup fit and the one-launch
#: local searches (NP = 64), the strip path with one-launch local searches (NP = 128, 256), the strip path with lockstep rounds
#: (NP = 320), and the first size of the next padding step
WARM_SIZES = (12, 70, 200, 270, 330)


def warm_up(engine, bounds, acquisition=None, kernel=None, n_restarts_optimizer: int = 5, n_constraints: int = 0,
            n_random: int = 10_000, lml_on_device="auto", sizes=WARM_SIZES) -> float:
    """Pay the first-use costs of a small-N suggest() now instead of inside the user's maximize() loop: every code object is
    loaded, every buffer family allocated for this dimension, every dispatch band's launch sequence (and its captured graph)
    run once.  Five synthetic default-configuration suggest() calls (theta search with restarts, candidates on the device,
    posterior, acquisition, local searches) over a stand-in space with the optimizer's bounds; own RandomStates — the
    optimizer's stream is not touched — and the engine's slots are left unfitted-by-anyone (the next real fit rewrites them).
    Returns the seconds it took.  What it replaces: nothing in the reference — `bayes_opt` has no device to warm; without it
    profiles/r05_maximize_loop.json shows 32 / 79 / 46 ms steps at N = 16 / 28 / 272 against 2-6 ms medians."""
    import copy
    import time

    from sklearn.base import clone
    from sklearn.gaussian_process.kernels import Matern

    from .float_space import FloatSpace

    t0 = time.perf_counter()
    bounds = np.asarray(bounds, dtype=np.float64).reshape(-1, 2)
    d = bounds.shape[0]
    if d < 1 or d > MAX_DIM:
        return 0.0
    span = np.where(bounds[:, 1] > bounds[:, 0], bounds[:, 1] - bounds[:, 0], 1.0)
    rng = np.random.RandomState(20240601)
    fn0 = acquisition if isinstance(acquisition, A.AcquisitionFunction) else A.UpperConfidenceBound(kappa=2.576)
    if kernel is None:
        kernel = Matern(nu=2.5)
    try:
        describe_kernel(kernel)
    except NotImplementedError:
        return 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for N in sizes:
            U = rng.uniform(size=(N, d))
            X = bounds[:, 0] + U * span
            y = np.exp(-((U - 0.4) ** 2).sum(1)) + 0.05 * rng.standard_normal(N)
            cm = None
            if n_constraints:
                from .constraint_model import HipConstraintModel

                cm = HipConstraintModel(None, np.full(n_constraints, -np.inf), np.full(n_constraints, 0.5), engine=engine,
                                        random_state=np.random.RandomState(3))
                for m in cm._model:
                    m.lml_on_device = lml_on_device
            sp = FloatSpace({f"w{j}": (bounds[j, 0], bounds[j, 0] + span[j]) for j in range(d)}, constraint=cm)
            cv = None if cm is None else np.cos(2.0 * U.sum(1))[:, None].repeat(n_constraints, 1).squeeze()
            sp.register_bulk(X, y, cv)
            gp = HipGPR(kernel=clone(kernel), alpha=1e-6, normalize_y=True, n_restarts_optimizer=n_restarts_optimizer,
                        random_state=np.random.RandomState(1), engine=engine)
            gp.lml_on_device = lml_on_device
            fn = copy.deepcopy(fn0)
            try:
                fn.suggest(gp, sp, n_random=n_random, n_smart=10, fit_gp=True, random_state=np.random.RandomState(2))
            except Exception:      # a policy that cannot run on synthetic data (EI without y_max, ...) warms what it reached
                pass
    return time.perf_counter() - t0


def accelerate(optimizer, device: int = 0, n_random: int | None = None, engine=None, precision: str = "f64",
               devices=None, local_search: str = "auto", lml_on_device="auto", warm: bool = True):
    """Swap the GP(s) and the acquisition function of `optimizer` in place; returns `optimizer`.

    `devices=[0, 1, ...]`: shard the random stage of every suggest() over these GPUs from this ONE process (GroupEngine:
    replicated fit, contiguous candidate blocks, one RCCL all-gather of the per-device arg-best records); the suggestion
    is bit for bit the single-GPU one.

    A model whose kernel is outside the HIP path (anything but Matern(nu=2.5) / RBF, see gpr.describe_kernel) — now, or after
    a later `optimizer.set_gp_params(kernel=...)` (bayes_opt/bayesian_optimization.py:403-407) — degrades instead of raising:
    that model runs scikit-learn's own fit / predict (the reference's trajectory, bit for bit) with one UserWarning, the
    fused acquisition classes run over its `predict`.  Supported models have NO CPU fallback: RuntimeError / ImportError
    when no GPU or no built library is available.
    `n_random` overrides the number of random candidates per suggest() (reference default 10_000).
    `precision="f32"` keeps the fp64 factorisation but runs the posterior contraction in fp32 (2x matrix rate).
    `engine` lets several optimizers share (or tests inject) a GpEngine; default: one per device.
    `lml_on_device`: where the theta search of every fit evaluates the log-marginal likelihood and its gradient (sklearn
    _gpr.py:296-338, 537-652).  "auto" (default) / True: on the device, the restarts advanced in lockstep (gpbo_lml_batch) —
    the same optimum to rounding, the shared RandomState consumed identically, 2x .. 300x faster than the host at N = 16 ..
    512 (profiles/r04_lml_crossover.json); False: sklearn's own host arithmetic, theta bit for bit the reference's.
    `local_search`: "auto" (default) runs the local-search stage as one library call (gpbo_polish_seeds: projected L-BFGS,
    analytic gradient, L-BFGS-B's stopping rule — about half the latency of a small-N suggest(), the same or a better
    acquisition value at the returned point, not the same iterates) wherever it applies: all-float spaces (no input
    transform) and stock UCB / EI / POI policies; mixed spaces and custom policies keep the reference-shaped stage.
    "device" asks for the same explicitly; "reference" keeps SciPy's L-BFGS-B over finite differences, iterate for iterate
    the reference's local searches (bayes_opt/acquisition.py:364-374).
    `warm` (default True): run `warm_up` once per engine and dimension — five synthetic suggest() calls (~0.3 s) that load the
    code objects, allocate and launch every small-N path, so that no suggest() of the user's loop carries a first-use spike.
    The optimizer's RandomState is not touched.
    """
    if local_search not in ("auto", "reference", "device"):
        raise ValueError("local_search must be 'auto', 'reference' or 'device'")
    if engine is None:
        engine = shared_engine(tuple(devices)) if devices is not None else shared_engine(device)
    space = optimizer._space
    transform = None if _identity_transform(space) else space.kernel_transform
    noted = _note_unsupported(optimizer._gp.kernel, "the target GP")
    width = int(getattr(space, "bounds", np.zeros((0, 2))).shape[0])      # columns in kernel space (categoricals are one-hot there)
    too_wide = None
    if width > MAX_DIM:
        too_wide = f"HIP path supports up to {MAX_DIM} dimensions in kernel space, this space has {width}"
        if not noted:
            warnings.warn(f"accelerate(): the space is {width} columns in kernel space (parameter.py:434-449: a categorical is "
                          f"one-hot), the HIP engine takes {MAX_DIM}; the models keep running scikit-learn's "
                          "GaussianProcessRegressor on the host (the reference's path)", UserWarning, stacklevel=2)
        noted = noted or too_wide
    optimizer._gp = HipGPR.from_sklearn(optimizer._gp, transform=transform, engine=engine, slot=0, precision=precision)
    optimizer._gp.lml_on_device = lml_on_device
    if noted:
        optimizer._gp._host_warned = noted          # said once, here
    constraint = getattr(space, "_constraint", None)
    if constraint is not None:
        if len(constraint._model) > 7:
            raise NotImplementedError("at most 7 constraint GPs fit the engine's model slots")
        for j, m in enumerate(constraint._model):
            noted = _note_unsupported(m.kernel, f"constraint GP {j}") or too_wide
            constraint._model[j] = HipGPR.from_sklearn(m, transform=transform, engine=engine, slot=j + 1, precision=precision)
            constraint._model[j].lml_on_device = lml_on_device
            if noted:
                constraint._model[j]._host_warned = noted
    optimizer._acquisition_function = _convert_acquisition(optimizer._acquisition_function)
    if n_random is not None and isinstance(optimizer._acquisition_function, A.AcquisitionFunction):
        optimizer._acquisition_function.default_n_random = int(n_random)
    if local_search != "auto":
        fn = optimizer._acquisition_function
        for f in [fn, getattr(fn, "base_acquisition", None), *getattr(fn, "base_acquisitions", [])]:
            if isinstance(f, A.AcquisitionFunction):
                f.device_polish = (local_search == "device")
    from .engine import GpEngine

    if warm and isinstance(engine, GpEngine) and transform is None and not too_wide and not engine.__dict__.get("_warmed", {}).get(width):
        # (a real engine only: test doubles have nothing to warm; mixed spaces run their local searches on the host and warm
        # as they go; once per engine and dimension)
        fn = optimizer._acquisition_function
        warm_up(engine, space.bounds, acquisition=fn if isinstance(fn, A.AcquisitionFunction) else None,
                kernel=optimizer._gp.kernel, n_restarts_optimizer=int(optimizer._gp.n_restarts_optimizer or 0),
                n_constraints=0 if constraint is None else len(constraint._model),
                n_random=int(getattr(fn, "default_n_random", 10_000)), lml_on_device=lml_on_device)
        engine.__dict__.setdefault("_warmed", {})[width] = True
    return optimizer
