"""Build libgpbo.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m bayesianoptimization_amd.build [--force]

hipcc cross-compiles without a GPU.  Two shared libraries land next to this file so that they travel with the source
tree:
  libgpbo.so      the product: the C ABI of include/gpbo.h, no debug entry points, no A/B environment switches
  libgpbo_dbg.so  the same sources with -DGPBO_DEBUG: + the self-test seams / micro-benchmarks (gpbo_debug_*, probes) and
                  the kernel A/B switches — what tests/ and scripts/ load when they need those (never the product path)
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
LIB_DEBUG = os.path.join(HERE, "libgpbo_dbg.so")
OBJDIR = os.path.join(HERE, "build")
ARCH = "gfx950"

# translation unit -> extra flags
SOURCES = {
    "gpbo_api.hip": [],
    "fit_kernels.hip": [],
    "chol_kernels.hip": os.environ.get("GPBO_CHOL_EXTRA_FLAGS", "").split(),   # probe builds (scripts/archive/r03_*): -D switches
    "posterior_kernel.hip": [],
    "posterior_kernel_v2.hip": [],
    "posterior_small.hip": [],
    "polish.hip": [],                          # gpbo_polish_seeds: the local-search stage as one C call (host optimiser, device evaluations)
    "polish_fused.hip": [],                    # ... and as one launch for NP <= 256: one workgroup per local search (NP <= 128: thread = training point), evaluations + optimiser inside
    "posterior_kernel_f32.hip": [],
    "posterior_cov.hip": [],
    "lml_kernels.hip": [],
    "fused_small.hip": [],                     # fit / LML evaluation of a small problem (NP <= 64) as ONE launch of ONE workgroup
    "mid_fit.hip": [],                         # 64 < NP <= 768: inputs, quarter-tile K, W = L^-1 by column strips, alpha (~15 launches per fit)
    "acq_kernels.hip": ["-ffp-contract=off"],  # elementwise formulas follow NumPy op by op
    "candidates.hip": [],
    "mt19937.hip": ["-ffp-contract=off"],      # lo + (hi - lo) * u as NumPy computes it
    "mt_jump.hip": [],                         # jump-ahead polynomials (host) + sub-stream start states (device)
    "probe.hip": [],                           # calibration: MFMA fp64 peak / sustained clock, HBM copy peak (bench.py's roofline keys)
    "comm.hip": [],
}
DEBUG_ONLY_SOURCES = {
    "latency_probe.hip": [],                   # single-wave instruction latency probe (the one scratch-using kernel): debug build only
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
          "-I" + INCLUDE, "-I" + CSRC, "-I/opt/rocm/include"] + os.environ.get("GPBO_EXTRA_FLAGS", "").split()   # probe builds: -D switches


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libgpbo.so")
    return exe


def _code_only(src: str) -> str:
    """C / C++ source without its comments, trailing blanks and empty lines (string and character literals are left alone).
    The fingerprint below is taken over THIS: it names the code a library was built from, so that a measurement stamped with it
    (profiles/*pmc*.json) stays attached to that code when only its commentary is edited."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == "'" and i > 0 and (src[i - 1].isalnum() or src[i - 1] == "_") and i + 1 < n and src[i + 1].isalnum():
            out.append(c)                                 # a C++14 digit separator (1'000), not the start of a character literal
            i += 1
        elif c == '"' and i > 0 and src[i - 1] == "R":    # a raw string literal R"delim( ... )delim": copied whole, nothing inside is a comment
            k = src.find("(", i)
            delim = src[i + 1:k] if k >= 0 else ""
            end = src.find(")" + delim + '"', k) if k >= 0 else -1
            j = n if end < 0 else end + len(delim) + 2
            out.append(src[i:j])
            i = j
        elif c in "\"'":                                 # a literal: copy it whole (escapes included)
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1])
            i = j + 1
        elif c == "/" and i + 1 < n and src[i + 1] == "/":     # to the end of the line (a backslash-newline continues it)
            j = i
            while j < n and src[j] != "\n":
                j += 2 if (src[j] == "\\" and j + 1 < n) else 1
            i = j
        elif c == "/" and i + 1 < n and src[i + 1] == "*":
            j = src.find("*/", i + 2)
            out.append(" ")
            i = n if j < 0 else j + 2
        else:
            out.append(c)
            i += 1
    lines = [ln.rstrip() for ln in "".join(out).split("\n")]
    return "\n".join(ln for ln in lines if ln)


def _fingerprint() -> str:
    h = hashlib.sha256()
    names = sorted(os.listdir(CSRC)) + ["../../include/gpbo.h"]
    for n in names:
        p = os.path.normpath(os.path.join(CSRC, n))
        if os.path.isfile(p):
            h.update(n.encode())
            h.update(_code_only(open(p, "r", encoding="utf-8").read()).encode())
    h.update(repr(sorted(SOURCES.items())).encode())
    h.update(repr(sorted(DEBUG_ONLY_SOURCES.items())).encode())
    # flags without the absolute include paths: the same sources must give the same fingerprint wherever the tree lies
    # (the GPU box runs from a scratch copy; profiles/ stamps its PMC summaries with this value)
    h.update(" ".join(f for f in COMMON if not f.startswith("-I")).encode())
    return h.hexdigest()


def _build_one(lib: str, debug: bool, force: bool, verbose: bool) -> str:
    stamp = lib + ".fingerprint"   # next to the library, so the pair travels together
    fp = _fingerprint() + ("-debug" if debug else "")
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == fp:
        return lib
    objdir = OBJDIR + ("_dbg" if debug else "")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    sources = dict(SOURCES)
    if debug:
        sources.update(DEBUG_ONLY_SOURCES)
    defines = ["-DGPBO_DEBUG"] if debug else []

    def compile_one(item):
        src, extra = item
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *COMMON, *defines, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(sources))) as ex:
        objs = list(ex.map(compile_one, sources.items()))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", lib, "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    open(stamp, "w").write(fp)
    if verbose:
        print(f"built {lib}")
    return lib


def build(force: bool = False, verbose: bool = True) -> str:
    """Build both libraries (each only when its sources / flags changed); returns the PRODUCT library's path."""
    _build_one(LIB_DEBUG, True, force, verbose)
    return _build_one(LIB, False, force, verbose)


def build_debug(force: bool = False, verbose: bool = True) -> str:
    return _build_one(LIB_DEBUG, True, force, verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
