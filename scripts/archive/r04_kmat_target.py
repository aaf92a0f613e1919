"""Target of scripts/archive/r04_kmat_pmc.sh: a few fixed-theta fits at the two BASELINE sizes, so that rocprofv3 sees kmat_kernel
(kernel-matrix assembly, csrc/fit_kernels.hip; replaces Matern.__call__(X), sklearn kernels.py:1711-1738) with N = 4096 /
d = 16 and N = 8192 / d = 32."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd.engine import MATERN25, GpEngine  # noqa: E402

eng = GpEngine(0)
for N, d, ls in ((4096, 16, 1.5), (8192, 32, 2.0)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(N, d))
    y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(N)
    yn = (y - y.mean()) / y.std()
    for _ in range(3):
        eng.fit(X, yn, MATERN25, ls, 1e-6)
    print(N, d, "kmat ms", eng.last_timings()["kmat"], flush=True)
