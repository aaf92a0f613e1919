"""TEST INFRASTRUCTURE ONLY — makes the read-only reference importable in this container.

Only tests/, oracle/gen_golden.py and bench.py's cpu_baseline leg may import this module.
The reference (bayes_opt 3.3.0) lives at /root/reference and is NOT installed; two shims are needed
(SURVEY.md §8c):
  1. `colorama` is absent from the image -> oracle/refshim/colorama.py (empty colour codes);
  2. bayes_opt/__init__.py:14 asks importlib.metadata for the distribution version -> patched here.
/root/reference does not exist on the GPU box: `have_reference()` is False there and callers skip.
"""
from __future__ import annotations

import importlib
import importlib.metadata as _md
import os
import sys

REFERENCE_ROOT = os.environ.get("GPBO_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")


def have_reference() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "bayes_opt"))


def import_reference():
    """Return the reference `bayes_opt` module, or raise ImportError when it is not mounted."""
    if not have_reference():
        raise ImportError(f"reference not mounted at {REFERENCE_ROOT}")
    try:
        importlib.import_module("colorama")
    except ImportError:
        if _SHIM not in sys.path:
            sys.path.insert(0, _SHIM)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if not getattr(_md.version, "_gpbo_patched", False):
        orig = _md.version

        def version(name):
            if name == "bayesian-optimization":
                return "3.3.0"
            return orig(name)

        version._gpbo_patched = True
        _md.version = version
    return importlib.import_module("bayes_opt")
