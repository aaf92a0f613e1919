"""File rendezvous for the one-process-per-GPU mode: ship rank 0's 128-byte RCCL unique id to its peers.

`python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py ...` starts N copies of the script with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment; nothing in this package needs torch for
that — the only thing the ranks must share before `gpbo_comm_init` is the id `ncclGetUniqueId` returned on rank 0.
All ranks of one node see the same filesystem and have the same parent (the launcher), so the id travels through one
small file keyed by (MASTER_PORT, parent pid): written atomically by rank 0, polled by the others, removed by rank 0
at the end.  Single node only (the multi-GPU target of this engine is the 8 GPUs of one node).
"""
from __future__ import annotations

import os
import tempfile
import time


def _parent_start_ticks() -> str:
    """Start time of the launcher process (clock ticks since boot, /proc/<ppid>/stat field 22): every rank of ONE launch
    reads the same value, and no other launch can — a pid may be reused, a (pid, start time) pair cannot.  This is the
    per-launch nonce of the key: a crashed run's leftover file under the same (port, launcher pid) has another name."""
    try:
        with open(f"/proc/{os.getppid()}/stat", "rb") as f:
            stat = f.read().decode(errors="replace")
        return stat[stat.rindex(")") + 2:].split()[19]      # fields after "pid (comm)": state is #3, starttime #22
    except (OSError, ValueError, IndexError):
        return "x"


def _path(key: str | None = None) -> str:
    base = os.environ.get("GPBO_RDZV_DIR", tempfile.gettempdir())
    if key is None:
        # (port, launcher pid, launcher start time) + the launcher's own run id / restart count when it has one: neither an
        # elastic restart under the same launcher nor a new launcher that happens to get the old one's pid meets the
        # previous attempt's file
        key = (f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{_parent_start_ticks()}"
               f"_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}_{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}")
    return os.path.join(base, f"gpbo_rdzv_{key}.id")


def share_unique_id(rank: int, make_id, key: str | None = None, timeout: float = 180.0) -> bytes:
    """Return the communicator id on every rank: rank 0 calls `make_id()` (-> 128 bytes) and publishes it."""
    path = _path(key)
    if rank == 0:
        uid = bytes(make_id())
        if len(uid) != 128:
            raise ValueError("an RCCL unique id is 128 bytes")
        for old in (path, path + ".done"):     # leftovers of a crashed run with the same key
            try:
                os.unlink(old)
            except OSError:
                pass
        tmp = f"{path}.{os.getpid()}.tmp"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)     # nobody else's file, nobody else's to read
        with os.fdopen(fd, "wb") as f:
            f.write(uid)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)          # atomic: a reader sees nothing or all 128 bytes
        return uid
    t_start = time.time()
    deadline = t_start + timeout
    while time.time() < deadline:
        try:
            # a file older than this launch is a leftover of a crashed run with the same key (the ranks of one launch
            # start within seconds of each other; rank 0 also removes any old file before it writes)
            if os.path.getmtime(path) >= t_start - 60.0:
                with open(path, "rb") as f:
                    uid = f.read()
                if len(uid) == 128:
                    return uid
        except OSError:
            pass
        time.sleep(0.02)
    raise TimeoutError(f"rank {rank}: no communicator id at {path} after {timeout:.0f} s (is rank 0 alive?)")


def mark_done(rank: int, key: str | None = None) -> None:
    """Rank 0: tell the peers (host side, no device work) that everything it does alone after the timed region is over."""
    if rank == 0:
        path = _path(key) + ".done"
        try:
            os.unlink(path)            # ours from an earlier call, or a leftover: never written THROUGH
        except OSError:
            pass
        try:
            # a new file of our own: O_EXCL refuses an existing name, O_NOFOLLOW a symlink planted under it (the id file
            # gets the same treatment in share_unique_id)
            fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            os.close(fd)
        except OSError:
            pass


def wait_done(rank: int, key: str | None = None, timeout: float = 150.0) -> bool:
    """Ranks > 0: sleep on the host until rank 0 called mark_done (True) or `timeout` passed (False).  A peer that waited in
    the closing collective instead would keep an RCCL kernel spinning on its GPU while rank 0's child process measures
    ms/suggest on those very devices."""
    if rank == 0:
        return True
    path = _path(key) + ".done"
    deadline = time.time() + timeout
    while time.time() < deadline:
        if os.path.exists(path):
            return True
        time.sleep(0.05)
    return False


def cleanup(rank: int, key: str | None = None) -> None:
    if rank == 0:
        for path in (_path(key), _path(key) + ".done"):
            try:
                os.unlink(path)
            except OSError:
                pass
