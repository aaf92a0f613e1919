"""MI355X-native GP-posterior + acquisition engine for bayes_opt's suggest() hot path.

Hand-written HIP (gfx950) behind a C ABI (include/gpbo.h), loaded with ctypes.  No PyTorch, no
Triton, no CPU fallback: importing the host classes works anywhere, creating an engine requires the
built library and an AMD GPU.
"""
__version__ = "0.1.0"

__all__ = ["GpEngine", "HipGPR", "HipConstraintModel", "FloatSpace", "accelerate", "UpperConfidenceBound",
           "ExpectedImprovement", "ProbabilityOfImprovement"]


def __getattr__(name):  # lazy: `import bayesianoptimization_amd.workloads` must not need sklearn/scipy
    if name == "GpEngine":
        from .engine import GpEngine
        return GpEngine
    if name == "HipGPR":
        from .gpr import HipGPR
        return HipGPR
    if name == "HipConstraintModel":
        from .constraint_model import HipConstraintModel
        return HipConstraintModel
    if name == "FloatSpace":
        from .float_space import FloatSpace
        return FloatSpace
    if name == "accelerate":
        from .dropin import accelerate
        return accelerate
    if name in ("UpperConfidenceBound", "ExpectedImprovement", "ProbabilityOfImprovement", "AcquisitionFunction"):
        from . import fused_acquisition as acquisition
        return getattr(acquisition, name)
    raise AttributeError(name)
