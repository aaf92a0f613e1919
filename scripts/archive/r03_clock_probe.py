"""Does the diagonal-block kernel run faster when the rest of the chip is busy (clock / power state)?  The Cholesky's
in-kernel stamps (s_memtime ticks) alone, and while a second context keeps the GPU busy with a long GEMM loop."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402
from r03_chol_probe import spd  # noqa: E402

eng = GpEngine(0, debug=True)
eng2 = GpEngine(0, debug=True)


def fma_ticks():
    o = eng.latency_probe(28)
    return {"fma": float(o[1] - o[0]) / 64, "rsq": float(o[5] - o[0]) / 64, "lds_read": float(o[10] - o[0]) / 64,
            "column_chain": float(o[24] - o[0]) / 16}


def chol_stamp(A):
    L, dinv, st, ms, info = eng.debug_cholesky(A, variant=3, iters=6)
    d = np.diff(st[:7])
    return {"ms": round(ms, 4), "phases": d.tolist(), "wave0_16cols": int(st[7] - st[1]),
            "segments": [int(st[8] - st[1]), int(st[9] - st[8]), int(st[10] - st[9]), int(st[7] - st[10])]}


for n in (128, 512):
    A = spd(n, 1, "kernel")
    print(n, "alone   ", fma_ticks(), chol_stamp(A), flush=True)
    stop = [False]

    def burn():
        while not stop[0]:
            eng2.gemm_bench(4096, 4096, 4096, iters=20)

    th = threading.Thread(target=burn)
    th.start()
    time.sleep(0.3)
    for _ in range(3):
        print(n, "with GEMM", fma_ticks(), chol_stamp(A), flush=True)
    stop[0] = True
    th.join()
    time.sleep(0.2)
    print(n, "alone   ", fma_ticks(), chol_stamp(A), flush=True)
