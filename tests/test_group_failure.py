"""CPU: the device group's failure path.  gpbo_group::run gives every job a deadline and turns a worker
that fails or never comes back into an error for the caller; exercised through the self-test seam (a group of worker
threads without device contexts, include/gpbo.h: gpbo_group_debug_create / gpbo_group_debug_run — entry points of the debug
build, libgpbo_dbg.so: the same comm.hip as the product's)."""
import ctypes as C
import os
import subprocess
import sys

import pytest

from bayesianoptimization_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _group(lib, n):
    g = C.c_void_p()
    assert lib.gpbo_group_debug_create(n, C.byref(g)) == _lib.GPBO_OK
    return g


def test_every_worker_ok_then_one_fails_with_its_code():
    lib = _lib.load_debug_library()
    g = _group(lib, 4)
    assert lib.gpbo_group_size(g) == 4
    assert lib.gpbo_group_debug_run(g, -1, 0, -1, 0) == _lib.GPBO_OK
    assert lib.gpbo_group_debug_run(g, 2, _lib.ERR_HIP, -1, 0) == _lib.ERR_HIP            # rank 2's code, not a hang
    assert b"rank 2" in lib.gpbo_group_last_error(g)
    assert lib.gpbo_group_debug_run(g, -1, 0, 1, 50) == _lib.GPBO_OK                      # a slow rank inside the deadline is fine
    assert lib.gpbo_group_destroy(g) == _lib.GPBO_OK


def test_a_worker_that_never_returns_becomes_an_error_within_the_deadline():
    """Own process: the deadline is read from the environment and a stuck worker thread is left to the process."""
    code = r"""
import ctypes as C, sys, time
sys.path.insert(0, %r)
from bayesianoptimization_amd import _lib
lib = _lib.load_debug_library()
g = C.c_void_p()
assert lib.gpbo_group_debug_create(3, C.byref(g)) == 0
t0 = time.time()
rc = lib.gpbo_group_debug_run(g, -1, 0, 1, 8000)          # rank 1 is stuck for 8 s, the deadline is 1 s (+5 s of grace)
dt = time.time() - t0
msg = lib.gpbo_group_last_error(g).decode()
rc2 = lib.gpbo_group_debug_run(g, -1, 0, -1, 0)           # the group is broken: fails at once
t1 = time.time()
lib.gpbo_group_destroy(g)                                 # must not wait for the stuck worker
print("RESULT", rc, round(dt, 2), rc2, round(time.time() - t1, 2), msg)
""" % ROOT
    env = dict(os.environ, GPBO_GROUP_TIMEOUT_S="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, p.stderr[-2000:]
    _, rc, dt, rc2, t_destroy, msg = line[0].split(" ", 5)
    assert int(rc) == _lib.ERR_COMM and int(rc2) == _lib.ERR_COMM
    assert 1.0 <= float(dt) < 7.5                                      # deadline + grace, not the worker's 8 s
    assert float(t_destroy) < 1.0
    assert "rank(s) 1" in msg and "GPBO_GROUP_TIMEOUT_S" in msg


def test_python_layer_maps_comm_errors_to_runtime_error():
    lib = _lib.load_debug_library()
    g = _group(lib, 2)
    rc = lib.gpbo_group_debug_run(g, 0, _lib.ERR_COMM, -1, 0)
    with pytest.raises(_lib.GpboError):
        _lib.raise_for_status(lib, None, rc, group=g)
    # a communicator error marks the group broken for good
    assert lib.gpbo_group_debug_run(g, -1, 0, -1, 0) == _lib.ERR_COMM
    lib.gpbo_group_destroy(g)


def test_a_relayed_peer_failure_does_not_break_the_group_and_the_root_cause_is_reported():
    """healthy ranks that only relay "a peer's local step failed" (GPBO_ERR_PEER: the exchange completed, the
    communicators are intact) must not poison the group for good, and the caller must see the failing rank's own code."""
    lib = _lib.load_debug_library()
    g = _group(lib, 3)
    # every rank relays: the group stays usable
    assert lib.gpbo_group_debug_run(g, 0, _lib.ERR_PEER, -1, 0) == _lib.ERR_PEER
    assert lib.gpbo_group_debug_run(g, -1, 0, -1, 0) == _lib.GPBO_OK
    assert lib.gpbo_group_destroy(g) == _lib.GPBO_OK
