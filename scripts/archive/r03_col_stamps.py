"""In-kernel clocks of the diagonal workgroup (warm run): phases, and when each of the eight waves finished its own
8 columns of the first factorisation."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402
from r03_chol_probe import spd  # noqa: E402

eng = GpEngine(0, debug=True)
for n in [int(x) for x in (sys.argv[1:] or (64, 128, 512))]:
    A = spd(n, 1, "kernel")
    Lref = np.linalg.cholesky(A)
    for rep in range(2):
        L, dinv, st, ms, info = eng.debug_cholesky(A, variant=3, iters=4)
        err = np.linalg.norm(L - Lref) / np.linalg.norm(Lref)
        print(n, "ms", round(ms, 4), "err", f"{err:.1e}", "info", info, "phases", np.diff(st[:7]).tolist(),
              "waves done at", (st[7:15] - st[1]).tolist())
