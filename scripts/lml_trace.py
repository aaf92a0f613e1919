"""Workload for rocprofv3 --kernel-trace: three LML value+gradient evaluations at N = argv[1] (default 4096), d = 16."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd.engine import MATERN25, GpEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = GpEngine(0)
rng = np.random.RandomState(0)
X = rng.uniform(size=(N, 16))
y = np.exp(-((X - 0.5) ** 2).sum(1)) + 0.01 * rng.standard_normal(N)
yn = (y - y.mean()) / y.std()
for rep in range(4):
    t0 = time.perf_counter()
    v, g = eng.lml(X, yn, MATERN25, 1.3, 1e-6)
    print(rep, "lml ms", (time.perf_counter() - t0) * 1e3, v, g)
