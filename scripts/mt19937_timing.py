"""Device MT19937 candidate generation (index-parity mode) against the host stream: time and bitwise check."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

eng = GpEngine(0, debug=True)
for M, d in ((65536, 8), (1 << 20, 16), (1 << 20, 32)):
    lo, hi = np.zeros(d), np.ones(d)
    t0 = time.perf_counter()
    ref = np.random.RandomState(7)
    want = np.column_stack([ref.uniform(lo[t], hi[t], M) for t in range(d)])
    t_host = time.perf_counter() - t0
    ts = []
    for _ in range(3):
        dev = np.random.RandomState(7)
        t0 = time.perf_counter()
        eng.generate_candidates_like(M, lo, hi, dev)
        ts.append(time.perf_counter() - t0)
    idx = np.arange(0, M, 257)
    ok = np.array_equal(eng.get_candidate_rows(idx[:4096], d), want[idx[:4096]]) and dev.uniform() == ref.uniform()
    print(f"M={M} d={d}: host {t_host * 1e3:.1f} ms, device {min(ts) * 1e3:.2f} ms, bitwise {ok}", flush=True)

# where the time goes (outputs are invalid in the probe modes): 1 = regeneration only, 2 = emission only
M, d = 1 << 20, 16
for probe in ("1", "2"):
    os.environ["GPBO_MT_PROBE"] = probe
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        eng.generate_candidates_like(M, np.zeros(d), np.ones(d), np.random.RandomState(7))
        ts.append(time.perf_counter() - t0)
    print(f"probe {probe}: {min(ts) * 1e3:.2f} ms", flush=True)
os.environ.pop("GPBO_MT_PROBE")
