"""Build libgpbo.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m bayesianoptimization_amd.build [--force]

hipcc cross-compiles without a GPU.  The shared library lands next to this file
(bayesianoptimization_amd/libgpbo.so) so that it travels with the source tree.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libgpbo.so")
OBJDIR = os.path.join(HERE, "build")
ARCH = "gfx950"

# translation unit -> extra flags
SOURCES = {
    "gpbo_api.hip": [],
    "fit_kernels.hip": [],
    "chol_kernels.hip": os.environ.get("GPBO_CHOL_EXTRA_FLAGS", "").split(),   # probe builds (scripts/r03_*): -D switches
    "posterior_kernel.hip": [],
    "posterior_kernel_v2.hip": [],
    "posterior_small.hip": [],
    "polish.hip": [],                          # gpbo_polish_seeds: the local-search stage as one C call (host optimiser, device evaluations)
    "posterior_kernel_f32.hip": [],
    "posterior_cov.hip": [],
    "lml_kernels.hip": [],
    "acq_kernels.hip": ["-ffp-contract=off"],  # elementwise formulas follow NumPy op by op
    "candidates.hip": [],
    "mt19937.hip": ["-ffp-contract=off"],      # lo + (hi - lo) * u as NumPy computes it
    "mt_jump.hip": [],                         # jump-ahead polynomials (host) + sub-stream start states (device)
    "probe.hip": [],
    "latency_probe.hip": [],
    "comm.hip": [],
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
          "-I" + INCLUDE, "-I" + CSRC, "-I/opt/rocm/include"] + os.environ.get("GPBO_EXTRA_FLAGS", "").split()   # probe builds: -D switches


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libgpbo.so")
    return exe


def _fingerprint() -> str:
    h = hashlib.sha256()
    names = sorted(os.listdir(CSRC)) + ["../../include/gpbo.h"]
    for n in names:
        p = os.path.normpath(os.path.join(CSRC, n))
        if os.path.isfile(p):
            h.update(n.encode())
            h.update(open(p, "rb").read())
    h.update(repr(sorted(SOURCES.items())).encode())
    # flags without the absolute include paths: the same sources must give the same fingerprint wherever the tree lies
    # (the GPU box runs from a scratch copy; profiles/ stamps its PMC summaries with this value)
    h.update(" ".join(f for f in COMMON if not f.startswith("-I")).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    stamp = LIB + ".fingerprint"   # next to the library, so the pair travels together
    fp = _fingerprint()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == fp:
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(item):
        src, extra = item
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *COMMON, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES.items()))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", LIB, "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    open(stamp, "w").write(fp)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
