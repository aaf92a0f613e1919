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


def elementwise_err(sd, sd_ref, mu, mu_ref, s_y):
    """north_star's tolerance as written — "within 1e-5 relative" — per candidate: max |sd - sd_ref| / sd_ref over the candidates
    with sd_ref > 0, and max |mu - mu_ref| / max(|mu_ref|, s_y).  (rel_err above is a max-norm: a sigma a thousand times below the
    batch's largest is checked three decades looser by it.)"""
    sd, sd_ref, mu, mu_ref = (np.asarray(v, dtype=np.float64) for v in (sd, sd_ref, mu, mu_ref))
    pos = sd_ref > 0
    e_sd = float(np.max(np.abs(sd - sd_ref)[pos] / sd_ref[pos])) if pos.any() else 0.0
    e_mu = float(np.max(np.abs(mu - mu_ref) / np.maximum(np.abs(mu_ref), s_y)))
    return e_sd, e_mu
