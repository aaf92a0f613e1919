"""ctypes binding of libgpbo.so (the C ABI declared in include/gpbo.h).

There is NO CPU fallback: if the HIP library is missing or no AMD GPU is visible, loading or context
creation raises.  Nothing in this package imports the test oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpbo.so")
DEBUG_LIB_PATH = os.path.join(_HERE, "libgpbo_dbg.so")     # the same sources built with -DGPBO_DEBUG (tests / scripts only)

GPBO_OK = 0
ERR_INVALID, ERR_HIP, ERR_NOT_PD, ERR_STATE, ERR_UNSUPPORTED, ERR_COMM, ERR_PEER = -1, -2, -3, -4, -5, -6, -7
MAX_MODELS = 8
MAX_DIM = 64
MAX_SEEDS = 64
ABI_VERSION = 2

_c_double_p = C.POINTER(C.c_double)
_c_int64_p = C.POINTER(C.c_int64)

# name -> (restype, argtypes); must list every symbol include/gpbo.h declares (tests check this)
SIGNATURES = {
    "gpbo_abi_version": (C.c_int, []),
    "gpbo_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gpbo_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "gpbo_destroy": (C.c_int, [C.c_void_p]),
    "gpbo_last_error": (C.c_char_p, [C.c_void_p]),
    "gpbo_synchronize": (C.c_int, [C.c_void_p]),
    "gpbo_device_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "gpbo_fit": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, C.c_int64, C.c_int, C.c_int,
                           _c_double_p, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]),
    "gpbo_fit_begin": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, C.c_int64, C.c_int, C.c_int,
                                 _c_double_p, C.c_int, C.c_double, C.c_int]),
    "gpbo_fit_wait": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "gpbo_fit_append": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, C.c_int64, C.c_int, _c_double_p, C.c_int64,
                                  C.POINTER(C.c_int)]),
    "gpbo_lml": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, C.c_int64, C.c_int, C.c_int, _c_double_p,
                           C.c_int, C.c_double, C.c_int, _c_double_p, _c_double_p, C.POINTER(C.c_int)]),
    "gpbo_lml_batch": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, C.c_int64, C.c_int, C.c_int, _c_double_p,
                                 C.c_int, C.c_double, C.c_int, _c_double_p, _c_double_p, C.POINTER(C.c_int)]),
    "gpbo_get_K": (C.c_int, [C.c_void_p, C.c_int, _c_double_p]),
    "gpbo_get_L": (C.c_int, [C.c_void_p, C.c_int, _c_double_p]),
    "gpbo_get_Linv": (C.c_int, [C.c_void_p, C.c_int, _c_double_p]),
    "gpbo_get_alpha": (C.c_int, [C.c_void_p, C.c_int, _c_double_p]),
    "gpbo_set_candidates": (C.c_int, [C.c_void_p, _c_double_p, C.c_int64, C.c_int]),
    "gpbo_generate_candidates": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, _c_double_p, _c_double_p, C.c_uint64]),
    "gpbo_generate_candidates_mt19937": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, _c_double_p, _c_double_p,
                                                   C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "gpbo_generate_candidate_rows_mt19937": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, _c_double_p,
                                                       _c_double_p, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32),
                                                       C.POINTER(C.c_int)]),
    "gpbo_mt19937_jump_blocks": (C.c_int, [C.POINTER(C.c_uint32), C.c_int64, C.POINTER(C.c_uint32)]),
    "gpbo_generate_candidate_columns_mt19937": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, _c_double_p, _c_double_p,
                                                          C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "gpbo_set_candidate_columns": (C.c_int, [C.c_void_p, _c_double_p, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "gpbo_transform_candidates": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gpbo_get_candidate_rows": (C.c_int, [C.c_void_p, _c_int64_p, C.c_int, _c_double_p]),
    "gpbo_posterior": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, _c_double_p, _c_double_p]),
    "gpbo_predict": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, C.c_int64, C.c_int, C.c_double, C.c_double,
                               _c_double_p, _c_double_p]),
    "gpbo_take_negative_variance_flag": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "gpbo_predict_cov": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, C.c_int64, C.c_int, C.c_double, C.c_double,
                                   _c_double_p, _c_double_p]),
    "gpbo_predict_grad": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, C.c_int64, C.c_int, C.c_double, C.c_double,
                                    _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "gpbo_acq_argbest": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, _c_double_p,
                                   _c_double_p, C.c_int, C.c_int64, _c_int64_p, _c_double_p, _c_int64_p,
                                   _c_double_p, _c_double_p]),
    "gpbo_last_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "gpbo_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "gpbo_comm_unique_id": (C.c_int, [C.c_char_p]),
    "gpbo_comm_init": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "gpbo_comm_allgather_best": (C.c_int, [C.c_void_p, _c_double_p, _c_int64_p, C.c_int, _c_double_p,
                                           _c_int64_p]),
    "gpbo_comm_destroy": (C.c_int, [C.c_void_p]),
    "gpbo_comm_acq_argbest": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, _c_double_p,
                                        _c_double_p, C.c_int, C.c_int64, _c_int64_p, _c_double_p, _c_int64_p,
                                        _c_double_p, _c_double_p]),
    "gpbo_comm_allreduce_max": (C.c_int, [C.c_void_p, _c_double_p]),
    "gpbo_group_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "gpbo_group_destroy": (C.c_int, [C.c_void_p]),
    "gpbo_group_size": (C.c_int, [C.c_void_p]),
    "gpbo_group_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "gpbo_group_collective": (C.c_char_p, [C.c_void_p]),
    "gpbo_group_last_error": (C.c_char_p, [C.c_void_p]),
    "gpbo_group_synchronize": (C.c_int, [C.c_void_p]),
    "gpbo_group_fit": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, C.c_int64, C.c_int, C.c_int,
                                 _c_double_p, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]),
    "gpbo_group_fit_append": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, C.c_int64, C.c_int, _c_double_p, C.c_int64,
                                        C.POINTER(C.c_int)]),
    "gpbo_group_lml_batch": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, C.c_int64, C.c_int, C.c_int, _c_double_p,
                                       C.c_int, C.c_double, C.c_int, _c_double_p, _c_double_p, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int)]),
    "gpbo_group_set_candidates": (C.c_int, [C.c_void_p, _c_double_p, C.c_int64, C.c_int]),
    "gpbo_group_generate_candidates_mt19937": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, _c_double_p, _c_double_p,
                                                         C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "gpbo_group_shard": (C.c_int, [C.c_void_p, C.c_int, _c_int64_p, _c_int64_p]),
    "gpbo_group_posterior": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, _c_double_p, _c_double_p]),
    "gpbo_group_acq_argbest": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, _c_double_p,
                                         _c_double_p, C.c_int, _c_int64_p, _c_double_p, _c_int64_p, _c_double_p,
                                         _c_double_p]),
    "gpbo_group_get_candidate_rows": (C.c_int, [C.c_void_p, _c_int64_p, C.c_int, _c_double_p]),
    "gpbo_polish_seeds": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, _c_double_p, _c_double_p, _c_double_p,
                                    _c_double_p, _c_double_p, C.c_int, C.c_int, _c_double_p, _c_double_p, C.c_int, _c_double_p,
                                    _c_double_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gpbo_mfma_f64_peak": (C.c_int, [C.c_void_p, C.c_int, _c_double_p]),
    "gpbo_mfma_f64_probe": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _c_double_p]),
    "gpbo_hbm_copy_peak": (C.c_int, [C.c_void_p, C.c_int64, _c_double_p]),
}

# entry points only libgpbo_dbg.so exports (include/gpbo.h, "#ifdef GPBO_DEBUG")
DEBUG_SIGNATURES = {
    "gpbo_debug_minimize_box": (C.c_int, [C.c_void_p, C.c_void_p, _c_double_p, C.c_int, C.c_int, _c_double_p, _c_double_p, C.c_int,
                                          _c_double_p, _c_double_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int)]),
    "gpbo_group_debug_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "gpbo_group_debug_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gpbo_debug_cholesky": (C.c_int, [C.c_void_p, _c_double_p, C.c_int64, C.c_int, C.c_int, _c_double_p, _c_double_p,
                                      C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "gpbo_debug_latency_probe": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "gpbo_debug_select": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p,
                                    C.POINTER(C.c_int64), C.POINTER(C.c_float)]),
    "gpbo_debug_gemm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, _c_double_p, _c_double_p,
                                  C.c_int, C.c_double, _c_double_p]),
    "gpbo_debug_gemm_bench": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        _c_double_p]),
    "gpbo_hybrid_probe": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _c_double_p]),
    "gpbo_debug_fail_next_acq": (C.c_int, [C.c_void_p]),
    "gpbo_debug_polish_eval": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _c_double_p, C.c_int, C.c_int,
                                         C.c_int, _c_double_p]),
}

_lib = None
_debug_lib = None


class GpboError(RuntimeError):
    """HIP / state / communicator failure reported by libgpbo."""


def _bind(path: str, signatures: dict):
    lib = C.CDLL(path)
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.gpbo_abi_version() != ABI_VERSION:
        raise ImportError(f"libgpbo ABI {lib.gpbo_abi_version()} != expected {ABI_VERSION}; rebuild")
    return lib


def load_library(path: str | None = None):
    """Load libgpbo.so and attach prototypes. Raises ImportError (loudly) when it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    explicit = path is not None
    path = path or LIB_PATH
    if not os.path.exists(path) and not explicit:
        # the library is git-ignored: on a fresh checkout build it in-tree once (hipcc cross-compiles)
        try:
            from .build import build
            build(verbose=False)
        except Exception as e:  # noqa: BLE001
            raise ImportError(f"{path} not found and building it failed ({e}). There is no CPU fallback.") from e
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension has not been built. Run "
            "`python -m bayesianoptimization_amd.build` (needs hipcc). There is no CPU fallback.")
    lib = _bind(path, SIGNATURES)
    if not explicit:
        _lib = lib
    return lib


def load_debug_library():
    """libgpbo_dbg.so: the product's sources + the debug entry points and A/B switches.  For tests and scripts — nothing on
    the product path (engine defaults, dropin, bench's timed region) loads it."""
    global _debug_lib
    if _debug_lib is None:
        if not os.path.exists(DEBUG_LIB_PATH):
            try:
                from .build import build_debug
                build_debug(verbose=False)
            except Exception as e:  # noqa: BLE001
                raise ImportError(f"{DEBUG_LIB_PATH} not found and building it failed ({e})") from e
        _debug_lib = _bind(DEBUG_LIB_PATH, {**SIGNATURES, **DEBUG_SIGNATURES})
    return _debug_lib


def dptr(a: np.ndarray | None):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_c_double_p)


def iptr(a: np.ndarray | None):
    if a is None:
        return None
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_c_int64_p)


def device_count() -> int:
    lib = load_library()
    n = C.c_int(0)
    rc = lib.gpbo_device_count(C.byref(n))
    return n.value if rc == GPBO_OK else 0


def raise_for_status(lib, handle, rc: int, info: int = 0, group=None):
    if rc == GPBO_OK:
        return
    msg = lib.gpbo_group_last_error(group) if group is not None else lib.gpbo_last_error(handle)
    msg = msg.decode(errors="replace") if msg else f"libgpbo error {rc}"
    if rc == ERR_NOT_PD:
        # same hint as sklearn (gaussian_process/_gpr.py:350-358)
        raise np.linalg.LinAlgError(
            f"The kernel is not returning a positive definite matrix ({info}-th leading minor of the array "
            "is not positive definite). Try gradually increasing the 'alpha' parameter of your "
            f"GaussianProcessRegressor estimator. [{msg}]")
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise GpboError(msg)
