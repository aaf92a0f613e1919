"""Host-side pieces of bench.py and of the one-process-per-GPU rendezvous (no GPU): the reference answer a sharded job is
compared with, the failure line, the PMC-summary guard, and the file rendezvous of the RCCL unique id."""
import importlib.util
import io
import json
import multiprocessing as mp
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_golden_merges_shards_as_one_concatenated_pass():
    """C4 at n GPUs = shards 0..n-1; the reference over the concatenated candidates would return the first global minimum
    and a stable top-k: the merge of the per-shard goldens by (value, global index)."""
    b = _bench()
    M = 1 << 20
    one = b.reference_golden("C4", 1, M)
    g0 = np.load(os.path.join(GOLD, "C4_s0.npz"))
    assert one["argmin"] == int(g0["argmin"]) and one["min"] == float(g0["min"])
    assert np.array_equal(one["top_idx"][:16], g0["topk_idx"][:16])
    for n in (2, 4, 8):
        got = b.reference_golden("C4", n, M)
        vals = np.concatenate([np.load(os.path.join(GOLD, f"C4_s{r}.npz"))["topk_val"] for r in range(n)])
        idx = np.concatenate([np.load(os.path.join(GOLD, f"C4_s{r}.npz"))["topk_idx"].astype(np.int64) + r * M for r in range(n)])
        o = np.lexsort((idx, vals))
        assert got["argmin"] == int(idx[o[0]]) and got["min"] == float(vals[o[0]])
        assert np.array_equal(got["top_idx"][:10], idx[o][:10])
        assert np.all(np.diff(got["top_val"][:64]) >= 0)
        assert f"s{n - 1}" in got["source"]
    # another shard size than the goldens were generated for: no reference, no parity block
    assert b.reference_golden("C4", 2, M // 2) is None
    assert b.reference_golden("C3", 2, M) is None and b.reference_golden("C3", 1, M)["argmin"] == 941430


def test_failed_line_is_loud_and_well_formed():
    b = _bench()
    args = types.SimpleNamespace(steps=5, warmup=2, config=None)
    buf = io.StringIO()
    with redirect_stdout(buf):
        b.emit_failed(args, 8, "RCCL bootstrap failed")
    d = json.loads(buf.getvalue())
    assert d["config"]["collective"] == "FAILED" and d["value"] is None and d["n_gpus"] == 8
    assert d["metric"] == b.METRIC and "RCCL" in d["error"]


def test_pmc_summary_is_used_only_for_the_library_it_was_taken_with(tmp_path, monkeypatch):
    b = _bench()
    from bayesianoptimization_amd import build
    w = types.SimpleNamespace(name="C3")
    pm, note = b.pmc_summary_for(w)
    latest = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_pmc_C3.json"))[-1]
    meta = json.load(open(os.path.join(ROOT, "profiles", latest)))["_meta"]
    if meta["source_fingerprint"] == build._fingerprint():
        assert pm is not None and "same kernel sources" in note
    else:
        assert pm is None and "another state" in note
    monkeypatch.setattr(build, "_fingerprint", lambda: "0" * 64)
    pm, note = b.pmc_summary_for(w)
    assert pm is None and "another state of the kernel sources" in note
    # C4 runs the kernels profiled for C3 (same GP, same candidates per GPU)
    monkeypatch.undo()
    assert "_pmc_C3.json" in b.pmc_summary_for(types.SimpleNamespace(name="C4"))[1]      # the latest round's C3 summary


def _peer(rank, key, rdzv_dir, q):
    os.environ["GPBO_RDZV_DIR"] = rdzv_dir
    sys.path.insert(0, ROOT)
    from bayesianoptimization_amd import rendezvous
    uid = rendezvous.share_unique_id(rank, lambda: bytes(range(128)), key=key, timeout=30.0)
    q.put((rank, uid))


def test_file_rendezvous_ships_the_unique_id_to_every_rank(tmp_path):
    from bayesianoptimization_amd import rendezvous
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = f"test_{os.getpid()}"
    procs = [ctx.Process(target=_peer, args=(r, key, str(tmp_path), q)) for r in (1, 2, 0)]   # rank 0 starts last
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0] == got[1] == got[2] == bytes(range(128))
    os.environ["GPBO_RDZV_DIR"] = str(tmp_path)
    try:
        rendezvous.cleanup(0, key=key)
        assert not [f for f in os.listdir(tmp_path) if f.endswith(".id")]
        with pytest.raises(TimeoutError):
            rendezvous.share_unique_id(1, None, key=key, timeout=0.2)
        with pytest.raises(ValueError):
            rendezvous.share_unique_id(0, lambda: b"short", key=key)
    finally:
        os.environ.pop("GPBO_RDZV_DIR", None)


def test_host_side_done_flag_keeps_the_peers_off_their_gpus_until_rank0_finished(tmp_path, monkeypatch):
    """bench.py (one process per GPU): ranks > 0 sleep on a host flag while rank 0 measures ms/suggest alone; a flag that never
    comes costs a bounded wait, a leftover flag of a crashed run is removed by the next launch's rank 0."""
    import threading
    import time
    from bayesianoptimization_amd import rendezvous
    monkeypatch.setenv("GPBO_RDZV_DIR", str(tmp_path))
    key = f"done_{os.getpid()}"
    assert rendezvous.wait_done(0, key=key, timeout=0.0) is True            # rank 0 never waits for itself
    t0 = time.time()
    assert rendezvous.wait_done(1, key=key, timeout=0.3) is False
    assert 0.25 <= time.time() - t0 < 5.0
    rendezvous.mark_done(1, key=key)                                         # only rank 0 may raise the flag
    assert rendezvous.wait_done(1, key=key, timeout=0.1) is False
    th = threading.Timer(0.2, rendezvous.mark_done, args=(0,), kwargs={"key": key})
    th.start()
    assert rendezvous.wait_done(3, key=key, timeout=30.0) is True
    th.join()
    rendezvous.share_unique_id(0, lambda: bytes(128), key=key)               # the next launch: stale flag gone, id published
    assert rendezvous.wait_done(1, key=key, timeout=0.1) is False
    rendezvous.mark_done(0, key=key)
    rendezvous.cleanup(0, key=key)
    assert os.listdir(tmp_path) == []


def test_extra_config_goldens_exist_for_the_line_the_driver_runs():
    """bench.py's `configs` block compares C2, C4 shard 0 and C5 shard 0 with these committed reference passes."""
    b = _bench()
    assert b.reference_golden("C2", 1, 65536)["source"].endswith("C2.npz")
    assert b.reference_golden("C4", 1, 1 << 20)["argmin"] == int(np.load(os.path.join(GOLD, "C4_s0.npz"))["argmin"])
    assert b.reference_golden("C5", 1, 1 << 18)["argmin"] == int(np.load(os.path.join(GOLD, "C5_s0.npz"))["argmin"])


def test_suggest_child_failure_costs_only_its_key(monkeypatch):
    """Multi-rank runs measure ms/suggest in a child process that owns all devices; without devices the child fails and the
    parent gets an error record instead of an exception (the launcher's rank variables must not reach the child)."""
    b = _bench()
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("GPBO_FORCE_NO_DEVICE", "1")
    res = b.suggest_in_child(2, "C4", timeout_s=120)
    assert isinstance(res, dict) and "error" in res


def test_rendezvous_files_are_never_written_through_a_symlink_and_the_key_is_per_launch(tmp_path, monkeypatch):
    """mark_done created its flag with O_CREAT|O_TRUNC — through a planted symlink that truncates somebody else's
    file; now it is unlink + O_EXCL|O_NOFOLLOW like the id file.  And the default key carries the launcher's START TIME,
    so a crashed run's leftover under the same (port, launcher pid) cannot feed a stale id to a new launch."""
    from bayesianoptimization_amd import rendezvous

    monkeypatch.setenv("GPBO_RDZV_DIR", str(tmp_path))
    victim = tmp_path / "victim.txt"
    victim.write_text("precious")
    key = "symlink_case"
    flag = rendezvous._path(key) + ".done"
    os.symlink(victim, flag)
    rendezvous.mark_done(0, key=key)
    assert victim.read_text() == "precious"                         # not truncated through the link
    assert os.path.exists(flag) and not os.path.islink(flag)        # the link was replaced by a file of our own
    assert oct(os.stat(flag).st_mode & 0o777) == "0o600"
    ticks = rendezvous._parent_start_ticks()
    assert ticks.isdigit() and ticks in os.path.basename(rendezvous._path())
    monkeypatch.setenv("MASTER_PORT", "29511")
    assert f"29511_{os.getppid()}_{ticks}_" in os.path.basename(rendezvous._path())


def test_the_build_fingerprint_names_the_code_not_its_comments(tmp_path, monkeypatch):
    """profiles/*pmc*.json are stamped with build._fingerprint(): it must move with every code change and with nothing else."""
    from bayesianoptimization_amd import build

    src = 'int a = 1; // one\n/* two\n lines */ const char* s = "// kept /* kept */"; char q = \'"\';\n\n#define M(x) \\\n  x // three\n'
    code = build._code_only(src)
    assert code == 'int a = 1;\n  const char* s = "// kept /* kept */"; char q = \'"\';\n#define M(x) \\\n  x'
    assert build._code_only(src.replace("// one", "// another comment").replace("two", "2")) == code
    assert build._code_only(src.replace("a = 1", "a = 2")) != code
    assert build._code_only(src.replace("// kept", "// Kept")) != code           # inside a string literal: code
    # digit separators, raw strings and character literals are code too: nothing behind them is swallowed
    tricky = 'int n = 1\'000\'000; // sep\nconst char* r = R"x(// not a comment */ ")x"; /* gone */ char q = \'"\'; int z = 2; // tail\n'
    got = build._code_only(tricky)
    assert "1'000'000;" in got and "sep" not in got and "gone" not in got and "tail" not in got
    assert 'R"x(// not a comment */ ")x";' in got and "char q = '\"';" in got and "int z = 2;" in got
    assert build._code_only(tricky.replace("// sep", "// another")) == got
    assert build._code_only(tricky.replace("int z = 2", "int z = 3")) != got
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    (csrc / "k.hip").write_text("__global__ void k() {}  // v1\n")
    monkeypatch.setattr(build, "CSRC", str(csrc))
    a = build._fingerprint()
    (csrc / "k.hip").write_text("// a new header comment\n__global__ void k() {}\n")
    assert build._fingerprint() == a
    (csrc / "k.hip").write_text("__global__ void k() { __syncthreads(); }\n")
    assert build._fingerprint() != a


def test_the_summary_is_the_last_key_material_and_stays_small():
    """Round 6: the driver's record keeps the headline keys and a TAIL of the line, so the small configs' numbers ride in a digest
    printed as the last key (bench.summary_of).  From a real line of the GPU box (profiles/r06_bench_default_C3_with_configs.json
    when it exists, else a synthetic one of the same shape): every config's ms / fit / frac / parity, the suggest() latencies of C2
    and C3, the strong-scaling block — in at most 1.5 KB."""
    b = _bench()
    p = os.path.join(ROOT, "profiles", "r06_bench_default_C3_with_configs.json")
    line = None
    if os.path.exists(p):
        rows = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
        line = json.loads(rows[-1]) if rows else None
    if line is None:
        cfg = {"ms_per_step": 0.641, "fit_ms": 0.1895, "roofline": {"frac": 0.6623, "frac_of_whole_step": 0.3649},
               "parity": {"argmin_equals_reference": True, "top10_equals_reference": True},
               "suggest_ms": {"n_smart_0": 1.18, "default_call_interior": 6.77, "smooth_target": {"default_call": 4.7}}}
        line = {"ms_per_step": 255.1, "n_gpus": 1, "roofline": {"frac": 0.899}, "roofline_fit": {"fit_ms_per_gp": 2.32, "chol_ms": 1.55},
                "parity": {"argmin_equals_reference": True, "top10_equals_reference": True}, "config": {"workload": "C3: d=16 ..."},
                "configs": {"C1": dict(cfg, suggest_ms=None), "C2": cfg, "C4_s0": dict(cfg, suggest_ms=None), "C5_f32_s0": {"error": "x" * 200}},
                "suggest_ms": {"n_smart_0": 257.7, "default_call_interior": 313.0, "default_call_interior_minus_n_smart_0": 55.3,
                               "smooth_target": {"default_call": 339.2}},
                "suggest_ms_fixed_total": {"fixed_theta_n_smart_0_ms": 257.1, "default_call_ms": 338.4, "theta_search_ms": 81.5,
                                           "posterior_ms_max_device": 255.5, "serial_fraction": 0.245},
                "cpu_baseline": {"value": 13317.7}}
    sm = b.summary_of(line)
    assert len(json.dumps(sm)) <= 1536
    head = sm[line["config"]["workload"].split(":")[0]]
    assert head["ms"] == round(line["ms_per_step"], 4) and head["frac"] == round(line["roofline"]["frac"], 4)
    assert set(head["fixed_total"]) == {"fixed_theta_n_smart_0_ms", "default_call_ms", "theta_search_ms", "posterior_ms_max_device", "serial_fraction"}
    assert head["suggest"]["default_interior_minus_n_smart_0"] is not None
    for key in ("C1", "C2", "C4_s0"):
        assert {"ms", "fit_ms", "frac", "frac_step", "argmin_ok", "top10_ok"} <= set(sm[key])
    assert "suggest" in sm["C2"] and sm["n_gpus"] == line["n_gpus"]
    # the line itself ends with it
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('out["summary"] = summary_of(out)') < src.index("print(json.dumps(out), flush=True)")
    assert src.count('out["') and src[src.index('out["summary"] = summary_of(out)'):src.index("print(json.dumps(out), flush=True)")].count('out["') <= 2
