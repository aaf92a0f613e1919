"""CPU: the C-ABI library builds/loads, exports every symbol include/gpbo.h declares, and fails
loudly (no fallback) when no GPU is present."""
import ctypes
import os
import re

import pytest

from bayesianoptimization_amd import _lib
from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "gpbo.h")


def _declared_symbols(debug=False):
    """Functions include/gpbo.h declares: the product part (outside `#ifdef GPBO_DEBUG`) or the debug-build part."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    head, rest = src.split("#ifdef GPBO_DEBUG", 1)
    dbg, tail = rest.split("#endif", 1)
    part = dbg if debug else head + tail
    return sorted(set(re.findall(r"\b(gpbo_[A-Za-z0-9_]+)\s*\(", part)) - {"gpbo_fg_callback"})


def _exported(path):
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("gpbo_")})


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m bayesianoptimization_amd.build`"
    assert os.path.dirname(_lib.LIB_PATH).endswith("bayesianoptimization_amd")


def test_every_header_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gpbo.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype in _lib.SIGNATURES"
    assert sorted(_lib.SIGNATURES) == declared
    assert _exported(_lib.LIB_PATH) == declared, "the product library exports exactly what the header's product part declares"


def test_debug_entry_points_live_in_the_debug_library_only():
    """self-test seams, single-kernel timers, probes and fault injection are compiled only with -DGPBO_DEBUG
    (libgpbo_dbg.so); the product library exports none of them."""
    from bayesianoptimization_amd import build

    build.build(verbose=False)
    debug_decl = _declared_symbols(debug=True)
    assert len(debug_decl) >= 8 and sorted(_lib.DEBUG_SIGNATURES) == debug_decl
    product = set(_exported(_lib.LIB_PATH))
    assert not product & set(debug_decl)
    assert not [n for n in product if "debug" in n or n in ("gpbo_hybrid_probe",)]
    assert _exported(_lib.DEBUG_LIB_PATH) == sorted(set(_declared_symbols()) | set(debug_decl))
    dbg = _lib.load_debug_library()
    assert dbg.gpbo_abi_version() == _lib.ABI_VERSION


ENV_ALLOW_LIST = {"GPBO_KSTAR_GB", "GPBO_COMM_TIMEOUT_S", "GPBO_GROUP_TIMEOUT_S", "GPBO_GROUP_HOST_MERGE"}


def test_the_product_reads_only_the_documented_environment_variables():
    """No switch that changes a result, or selects a retired kernel, survives in the product: every getenv() in csrc/ is
    either one of the four documented variables or sits behind dbg_env(), which is a constant NULL without -DGPBO_DEBUG."""
    csrc = os.path.join(ROOT, "bayesianoptimization_amd", "csrc")
    direct, debug_only = set(), set()
    for fn in sorted(os.listdir(csrc)):
        src = open(os.path.join(csrc, fn)).read()
        src = re.sub(r"//[^\n]*", "", src)
        for m in re.finditer(r"\b(getenv|dbg_env|env_seconds)\(\s*\"(GPBO_[A-Z0-9_]+)\"", src):
            (debug_only if m.group(1) == "dbg_env" else direct).add(m.group(2))
        # any other getenv must be the two generic helpers (dbg_env's own body, env_seconds' body)
        for m in re.finditer(r"\bgetenv\(\s*([^\")]+)\)", src):
            assert m.group(1).strip() == "name", (fn, m.group(0))
    assert direct == ENV_ALLOW_LIST, direct
    assert not debug_only & ENV_ALLOW_LIST
    hdr = open(HEADER).read()
    for name in ENV_ALLOW_LIST:
        assert name in hdr, f"{name} is not documented in include/gpbo.h"
    # the strings themselves: none of the debug-only names is in the product binary, all of them are in the debug one
    blob = open(_lib.LIB_PATH, "rb").read()
    dblob = open(_lib.DEBUG_LIB_PATH, "rb").read()
    for name in debug_only:
        assert name.encode() + b"\0" not in blob, f"{name} is still read by the product library"
        assert name.encode() in dblob
    for name in ("GPBO_POST_ABLATE_GEN", "GPBO_CHOL_OUTER", "GPBO_SELECT_V2"):
        assert name in debug_only
    for name in ENV_ALLOW_LIST:
        assert name.encode() in blob


def _gfx950_code_objects(so_path):
    """The gfx950 ELF code objects inside a library's .hip_fatbin section (one clang offload bundle per translation unit)."""
    import struct
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so_path, fat], check=True)
        data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], data.find(magic)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", data, pos + len(magic))
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(magic, pos + 1)
    return out


def _kernel_notes(so_path):
    """[(kernel name, private_segment_fixed_size)] of every gfx950 kernel in the library."""
    import shutil
    import subprocess
    import tempfile

    readelf = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf) or shutil.which("objcopy") is None:
        pytest.skip("no object tools")
    found = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, blob in enumerate(_gfx950_code_objects(so_path)):
            co = os.path.join(tmp, f"{i}.co")
            open(co, "wb").write(blob)
            notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
            names = re.findall(r"\.name:\s*(\S+)", notes)
            sizes = re.findall(r"\.private_segment_fixed_size:\s*(\d+)", notes)
            kernels = [n for n in names if n.startswith("_Z") or n.endswith("_kernel")]
            found += list(zip(kernels[:len(sizes)] if len(kernels) >= len(sizes) else ["?"] * len(sizes), map(int, sizes)))
    return found


def test_the_product_library_has_no_scratch_using_kernel():
    """Every gfx950 kernel inside libgpbo.so reports private_segment_fixed_size 0 (no spills, no stack); the latency probe —
    the one kernel that needs scratch — is in the debug library only."""
    prod = _kernel_notes(_lib.LIB_PATH)
    assert len(prod) >= 40
    assert all(sz == 0 for _, sz in prod), [k for k in prod if k[1]]
    dbg = _kernel_notes(_lib.DEBUG_LIB_PATH)
    assert len(dbg) > len(prod) and any(sz > 0 for _, sz in dbg)
    blob, dblob = open(_lib.LIB_PATH, "rb").read(), open(_lib.DEBUG_LIB_PATH, "rb").read()
    assert b"latency_probe_kernel" not in blob and b"latency_probe_kernel" in dblob
    assert b"hybrid_probe_kernel" not in blob


def test_abi_version_and_load():
    lib = _lib.load_library()
    assert lib.gpbo_abi_version() == _lib.ABI_VERSION
    hdr = open(HEADER).read()
    assert f"#define GPBO_ABI_VERSION {_lib.ABI_VERSION}" in hdr


def test_no_gpu_means_loud_failure():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible here")
    from bayesianoptimization_amd.engine import GpEngine

    with pytest.raises(_lib.GpboError):
        GpEngine(0)


def test_missing_library_raises_import_error(tmp_path):
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load_library(str(tmp_path / "libgpbo.so"))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bayesianoptimization_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
            assert "import torch" not in src, fn


def test_oracle_is_only_reachable_from_the_allowed_places():
    """Besides tests/: only bench.py's cpu_baseline leg and __graft_entry__ (build of the checker, smoke) may touch
    oracle/; the measurement scripts and the package may not."""
    import ast
    import glob

    def importers(path):
        tree = ast.parse(open(path).read())
        hits = []
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            if any(n == "oracle" or n.startswith("oracle.") for n in names):
                hits.append(node)
        return tree, hits

    for path in glob.glob(os.path.join(ROOT, "scripts", "*.py")) + glob.glob(os.path.join(ROOT, "bayesianoptimization_amd", "*.py")):
        assert not importers(path)[1], path
    tree, hits = importers(os.path.join(ROOT, "bench.py"))
    baseline = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "cpu_baseline")
    inside = {id(n) for n in ast.walk(baseline)}
    assert hits and all(id(h) in inside for h in hits)


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """include/gpbo.h is the boundary a non-Python host would bind: it must be valid C99 on its own, and a C program
    using only that header must link against libgpbo.so and run its GPU-free entry points (ABI version, device count —
    which reports an error, not a crash, on a box without a GPU — and the NULL-context error string)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    from bayesianoptimization_amd import build

    lib = build.build(verbose=False)
    for flags in ([], ["-DGPBO_DEBUG"]):      # the header is plain C with and without its debug section
        chk = tmp_path / "hdr.c"
        chk.write_text('#include "gpbo.h"\nint main(void) { return 0; }\n')
        subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", *flags,
                        "-I", os.path.join(ROOT, "include"), str(chk)], check=True)
    src = tmp_path / "probe.c"
    src.write_text(
        '#include <stdio.h>\n#include "gpbo.h"\n'
        "int main(void) {\n"
        "  int n = -1;\n"
        "  int rc = gpbo_device_count(&n);\n"
        "  const char* msg = gpbo_last_error(NULL);\n"
        '  printf("%d %d %d %s\\n", gpbo_abi_version() == GPBO_ABI_VERSION, rc, n, msg ? "msg" : "null");\n'
        "  return (rc == GPBO_OK || rc == GPBO_ERR_HIP) ? 0 : 1;\n"
        "}\n")
    exe = tmp_path / "probe"
    inc = os.path.join(ROOT, "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, str(src), "-o", str(exe),
                    lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out[0] == "1" and out[3] == "msg"
