"""Selection launches alone, both forms (gpbo_debug_select): ms per selection at the candidate counts of C2 / C3 / C4 and
k = 1 / 10 / 64 -> gpurun_out/r03_select_probe.json.  Run on the GPU box."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

eng = GpEngine(0, debug=True)
rng = np.random.RandomState(0)
out = {}
for M in (1 << 16, 1 << 20, 1 << 23):
    ys = rng.standard_normal(M)
    for k in (1, 10, 64):
        row = {}
        for variant in (1, 2):
            idx, _, _, ms = eng.debug_select(ys, k, variant=variant, iters=20)
            row[f"variant{variant}_ms"] = ms
            row[f"variant{variant}_first"] = int(idx[0])
        row["same_picks"] = bool(np.array_equal(eng.debug_select(ys, k, 1)[0], eng.debug_select(ys, k, 2)[0]))
        out[f"M={M},k={k}"] = row
        print(f"M={M} k={k}: passes {row['variant1_ms'] * 1e3:.1f} us, threshold+ranks {row['variant2_ms'] * 1e3:.1f} us, "
              f"same picks {row['same_picks']}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r03_select_probe.json", "w"), indent=1)
eng.close()
