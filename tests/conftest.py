import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    return dict(np.load(path))


@pytest.fixture(scope="session")
def engine():
    """One libgpbo context on cuda:0 — raises (does not skip) when the HIP library or GPU is missing."""
    from bayesianoptimization_amd.engine import GpEngine

    eng = GpEngine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def debug_engine():
    """A context in libgpbo_dbg.so (the product's sources built with -DGPBO_DEBUG): for the tests that need a debug entry
    point (gpbo_debug_*) or one of the kernel A/B environment switches, which the product library does not read."""
    from bayesianoptimization_amd.engine import GpEngine

    eng = GpEngine(0, debug=True)
    yield eng
    eng.close()


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))
