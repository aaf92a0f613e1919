"""The fit-side fp64 GEMM kernels on dense products (gpbo_debug_gemm_bench, debug build): TFLOP/s and fraction of the 78.6 TFLOP/s
matrix peak for the three operand layouts the fit uses (NN, NT = b_trans, TN = a_trans), square and SYRK-shaped, on the 128 x 128
kernel and (GPBO_GEMM128=0, in a child process) the 64 x 64 one.

    python scripts/r06_gemm_bench.py > profiles/r06_gemm_bench.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(4096, 4096, 4096, 0), (4096, 4096, 1024, 0), (2048, 2048, 2048, 0), (8192, 8192, 1024, 0), (4096, 4096, 4096, 1), (3072, 3072, 1024, 1)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from bayesianoptimization_amd.engine import GpEngine

    eng = GpEngine(0, debug=True)
    res = {}
    for (m, n, k, lower) in SHAPES:
        for name, bt, at in (("NN", 0, 0), ("NT", 1, 0), ("TN", 0, 1)):
            r = eng.gemm_bench(m, n, k, b_trans=bool(bt), a_trans=bool(at), lower_only=bool(lower), iters=10)
            res[f"{m}x{n}x{k}{'_lower' if lower else ''}_{name}"] = {"ms": round(r["ms"], 4), "tflops": round(r["tflops"], 2), "frac": round(r["tflops"] / 78.6, 3)}
    print(json.dumps(res))
    sys.exit(0)

out = {}
for label, env in (("gemm128", {}), ("gemm64", {"GPBO_GEMM128": "0", "GPBO_GEMM_FAT": "0"})):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    out[label] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
    for k, v in out[label].items():
        print(label, k, v, file=sys.stderr)
print(json.dumps(out))
