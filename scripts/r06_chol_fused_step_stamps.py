"""In-kernel clocks of the one-launch step's first step (GPBO_CHOL_FUSED_STEP=1, libgpbo_dbg.so): when the diagonal workgroup
published, when the first panel group saw it, when that group had stored and published, when the first next-diagonal tile saw the
panel and when it had stored — microseconds since the diagonal workgroup started, on the 100 MHz wall clock all CUs share.

    python scripts/r06_chol_fused_step_stamps.py >> profiles/r06_chol_fused_step_ab.json (merged by hand)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

def matern25_K(X, ls, noise):
    """A positive definite test matrix for the Cholesky-alone entry: Matern-2.5 over X / ls + noise I (kernels.py:1711-1738)."""
    Z = X / ls
    d2 = np.maximum((Z * Z).sum(1)[:, None] + (Z * Z).sum(1)[None, :] - 2.0 * Z @ Z.T, 0.0)
    r = np.sqrt(5.0 * d2)
    K = (1.0 + r + r * r / 3.0) * np.exp(-r)
    K[np.diag_indices_from(K)] = 1.0 + noise
    return K


eng = GpEngine(0, debug=True)
out = {}
for N in (1024, 4096):
    rng = np.random.RandomState(N)
    X = rng.uniform(size=(N, 16))
    K = matern25_K(X, 0.9, 1e-6)
    for setting in ("0", "1"):
        os.environ["GPBO_CHOL_FUSED_STEP"] = setting
        L, dinv, st, ms, info = eng.debug_cholesky(K, variant=3, iters=10)
        st = np.asarray(st, dtype=np.int64)
        rel = None
        if setting == "1":      # 100 MHz wall clock (10 ns ticks), the same on every CU
            rel = {k: round(float(st[i] - st[14]) * 0.01, 2) for k, i in (("diag_body_end", 14), ("diag_published", 8), ("panel_group0_saw_flag", 9),
                                                                         ("panel_group0_body_done", 13), ("panel_group0_stored_and_published", 10), ("next_diag_tile0_saw_panel", 11),
                                                                         ("next_diag_tile0_stored", 12))}
        out[f"N{N}_fused{setting}"] = {"cholesky_ms": round(ms, 4), "info": int(info), "diag_body_cycles": int(st[6] - st[0]),
                                       "first_step_us_after_the_diagonal_body_ended": rel}
os.environ.pop("GPBO_CHOL_FUSED_STEP", None)
print(json.dumps(out, indent=1))
