"""CPU: the C-ABI library builds/loads, exports every symbol include/gpbo.h declares, and fails
loudly (no fallback) when no GPU is present."""
import ctypes
import os
import re

import pytest

from bayesianoptimization_amd import _lib
from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "gpbo.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpbo_[A-Za-z0-9_]+)\s*\(", src)))


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
