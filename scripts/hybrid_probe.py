import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd.engine import GpEngine
eng = GpEngine(0, debug=True)
names = {0: "16 MFMA", 1: "256 VALU", 2: "16 MFMA + 256 VALU", 3: "16 MFMA + 128 VALU", 4: "8 MFMA + 256 VALU"}
out = {}
for rnd in range(2):
    for cfg in range(5):
        r = eng.hybrid_probe(3000, cfg)
        r["total_tflops"] = r["mfma_tflops"] + r["valu_tflops"]
        out[f"{names[cfg]} #{rnd}"] = r
        print(names[cfg], {k: round(v, 2) for k, v in r.items()}, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "hybrid_probe.json"), "w"), indent=1)
