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

from . import fused_acquisition as A
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


def accelerate(optimizer, device: int = 0, n_random: int | None = None, engine=None, precision: str = "f64",
               devices=None, local_search: str = "auto", lml_on_device="auto"):
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
    """
    if local_search not in ("auto", "reference", "device"):
        raise ValueError("local_search must be 'auto', 'reference' or 'device'")
    if engine is None:
        engine = shared_engine(tuple(devices)) if devices is not None else shared_engine(device)
    space = optimizer._space
    transform = None if _identity_transform(space) else space.kernel_transform
    noted = _note_unsupported(optimizer._gp.kernel, "the target GP")
    optimizer._gp = HipGPR.from_sklearn(optimizer._gp, transform=transform, engine=engine, slot=0, precision=precision)
    optimizer._gp.lml_on_device = lml_on_device
    if noted:
        optimizer._gp._host_warned = noted          # said once, here
    constraint = getattr(space, "_constraint", None)
    if constraint is not None:
        if len(constraint._model) > 7:
            raise NotImplementedError("at most 7 constraint GPs fit the engine's model slots")
        for j, m in enumerate(constraint._model):
            noted = _note_unsupported(m.kernel, f"constraint GP {j}")
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
    return optimizer
