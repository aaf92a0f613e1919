"""Round 4: the diagonal workgroup of chol128_step_kernel by phases (in-kernel clocks of a warm run) and the whole factorisation,
checked against numpy (L and the inverted 64x64 diagonal blocks).  Usage: python scripts/archive/r04_chol_chain.py [n ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402
from r03_chol_probe import spd  # noqa: E402

PHASES = ["load", "factor 0..63 (+L10)", "syrk", "exchange + wave 0 of factor 64..127", "rest of factor 64..127", "inverses"]
if os.environ.get("GPBO_CHAIN_LIB"):      # an experiment build of the debug library (scripts/archive/r04_chol_wake_ab.sh)
    from bayesianoptimization_amd import _lib
    _lib._debug_lib = _lib._bind(os.environ["GPBO_CHAIN_LIB"], {**_lib.SIGNATURES, **_lib.DEBUG_SIGNATURES})
eng = GpEngine(0, debug=True)
out = {}
for n in [int(x) for x in (sys.argv[1:] or (128, 512, 2048, 4096))]:
    A = spd(n, 1, "kernel")
    Lref = np.linalg.cholesky(A)
    best = None
    for rep in range(3):
        L, dinv, st, ms, info = eng.debug_cholesky(A, variant=3, iters=6)
        if best is None or ms < best[0]:
            best = (ms, st)
    err = float(np.linalg.norm(L - Lref) / np.linalg.norm(Lref))
    derr = max(float(np.linalg.norm(dinv[b] @ Lref[64 * b:64 * b + 64, 64 * b:64 * b + 64] - np.eye(64))) for b in range(n // 64))
    ph = np.diff(best[1][:7]).tolist()
    out[n] = {"ms": round(best[0], 4), "us_per_128_columns": round(best[0] * 1e3 / (n / 128), 2), "rel_err_L": err, "dinv_resid": derr,
              "info": int(info), "phases_cycles": dict(zip(PHASES, ph)), "diag_workgroup_cycles": int(sum(ph)),
              "waves_done_at": (best[1][7:15] - best[1][1]).tolist(), "diag16_done_after_barrier": int(best[1][15] - best[1][5])}
    print(n, json.dumps(out[n]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_chol_chain.json"), "w"), indent=1)
