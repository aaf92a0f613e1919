"""CPU, world_size = 2 over gloo: the N>1 path (contiguous candidate shards, per-rank arg-best with
global indices, all-gather of (value, index) records, identical merge on every rank) reproduces the
single-rank result.  The per-rank local evaluation is supplied by the CPU oracle (tests only); the
transport is torch.distributed all_gather, standing in for gpbo_comm_allgather_best (RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from bayesianoptimization_amd import workloads as W
    from bayesianoptimization_amd.distributed import ShardedAcquisition
    from helpers import OracleEngine, oracle_case

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = W.ALL[case["name"]]
        oc = oracle_case(w, case["ls"], M=case["M"], c_length_scale=case.get("cls"))
        Xc = oc["Xc"]
        if case.get("poison"):  # NaN candidates: "first NaN wins" must hold across shards
            Xc = Xc.copy()
            Xc[case["poison"]] = np.nan

        def allgather(vals, idxs):
            tv = [torch.zeros(len(vals), dtype=torch.float64) for _ in range(world)]
            ti = [torch.zeros(len(idxs), dtype=torch.int64) for _ in range(world)]
            dist.all_gather(tv, torch.from_numpy(np.ascontiguousarray(vals)))
            dist.all_gather(ti, torch.from_numpy(np.ascontiguousarray(idxs)))
            return torch.cat(tv).numpy(), torch.cat(ti).numpy()

        sh = ShardedAcquisition(OracleEngine(oc["gp"], oc["cons"]), world, rank, allgather)
        sh.set_candidates_global(Xc)
        lb = [-np.inf] if w.constrained else None
        ub = [w.constraint_ub] if w.constrained else None
        out = sh.argbest(w.acq, w.acq_param, oc["y_max"] or 0.0, lb, ub, k_seeds=case["k"])
        q.put((rank, out[0], out[1], np.asarray(out[2]).tolist(), np.asarray(out[3]).tolist()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


CASES = [
    {"name": "P1", "ls": 0.4, "M": 3001, "k": 10},
    {"name": "C5S", "ls": 0.5, "cls": 0.7, "M": 2048, "k": 4},
    {"name": "P2", "ls": 0.6, "M": 1500, "k": 5, "poison": [1203, 40]},
]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_two_rank_merge_equals_single_rank(case):
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bayesianoptimization_amd import workloads as W
    from helpers import OracleEngine, oracle_case

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-rank truth
    w = W.ALL[case["name"]]
    oc = oracle_case(w, case["ls"], M=case["M"], c_length_scale=case.get("cls"))
    Xc = oc["Xc"].copy()
    if case.get("poison"):
        Xc[case["poison"]] = np.nan
    eng = OracleEngine(oc["gp"], oc["cons"])
    eng.set_candidates(Xc)
    bi, bv, si, sv, _ = eng.acq_argbest(w.acq, w.acq_param, oc["y_max"] or 0.0, k_seeds=case["k"])
    for rank, rbi, rbv, rsi, rsv in results:
        assert rbi == bi
        assert (np.isnan(rbv) and np.isnan(bv)) or rbv == bv
        assert rsi == list(si)
        assert np.array_equal(np.asarray(rsv), sv, equal_nan=True)
    if case.get("poison"):
        assert bi == min(case["poison"]) and np.isnan(bv)
