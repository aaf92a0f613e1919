"""Workload for `rocprofv3 --kernel-trace --stats`: the Cholesky alone (gpbo_debug_cholesky) at n = 4096 (argv[1]), variant argv[2]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

from r03_chol_probe import spd  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = GpEngine(0, debug=True)
A = spd(n, 1, "kernel")
L, dinv, stamps, ms, info = eng.debug_cholesky(A, variant=variant, iters=4)
print(n, variant, ms, info)
