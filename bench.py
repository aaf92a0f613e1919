#!/usr/bin/env python
"""bench.py — suggest-step throughput of the HIP GP-posterior + acquisition engine on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--no-cpu-baseline]

A "step" is one pass of the hot path over one batch of synthetic input (BASELINE.json metric,
configs[2] = C3 when it fits one GPU): fit the GP at fixed theta (K, Cholesky, W = L^-1, alpha), then
posterior mu/sigma + acquisition + arg-best/top-10 over M = 2^20 candidates that are already
resident in HBM.  N > 1 (launched by torch.distributed.run, one process per GPU): every rank fits
redundantly and evaluates its own 2^20-candidate shard (weak scaling); the only exchange is an
all-gather of 11 (value, index) records per rank over RCCL.  Rank 0 prints ONE JSON line.

torch is used only as rendezvous plumbing (process group, barrier, max-over-ranks of the wall time)
when N > 1; the product path is ctypes -> libgpbo.so (HIP).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bayesianoptimization_amd import workloads as W  # noqa: E402
from bayesianoptimization_amd.distributed import ShardedAcquisition  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet (matrix FP64); the guide lists no fp64 row
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def flops_per_candidate(N, d, n_gp=1):
    """F_cand = N^2 + (3d + 12) N per GP (one triangular solve + k* build + mu/sigma)."""
    return n_gp * (float(N) * N + (3.0 * d + 12.0) * N)


def cpu_baseline(w, X, y, Xc, y_max, gpu_ys, n_chunks=3, chunk=8192):
    """The reference's CPU arithmetic on this box's host cores, on a bounded sample of the same workload.

    scikit-learn's GaussianProcessRegressor (what bayes_opt delegates to: acquisition.py:84,205,216) at the
    same fixed theta, plus the restated _get_acq closure from oracle/gp_oracle.py; parity is asserted on
    the very chunks that are timed."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, Matern

    from oracle import gp_oracle as O

    k = RBF(length_scale=w.length_scale) if w.kernel == W.RBF else Matern(nu=2.5, length_scale=w.length_scale)
    gp = GaussianProcessRegressor(kernel=k, alpha=w.noise, normalize_y=True, optimizer=None)
    t0 = time.perf_counter()
    gp.fit(X, y)
    fit_s = time.perf_counter() - t0
    times, worst = [], 0.0
    for c in range(n_chunks):
        xs = Xc[c * chunk:(c + 1) * chunk]
        t0 = time.perf_counter()
        mean, std = gp.predict(xs, return_std=True)
        ys = -1 * O.base_acq(w.acq, mean, std, w.acq_param, y_max if y_max is not None else 0.0)
        times.append(time.perf_counter() - t0)
        g = gpu_ys[c * chunk:(c + 1) * chunk]
        worst = max(worst, float(np.max(np.abs(g - ys)) / np.max(np.abs(ys))))
    med = float(np.median(times))
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count()
    step_s = fit_s + med * (w.M / chunk)  # cost is linear in M: extrapolated to the full step
    return {
        "value": w.M / step_s, "unit": "candidates/s", "cores": int(blas_threads), "kind": "port",
        "sample": (f"sklearn {__import__('sklearn').__version__} GaussianProcessRegressor.fit (fixed theta) {fit_s:.2f}s + "
                   f"predict(return_std)+acq on {n_chunks} chunks of {chunk} candidates (median {med:.2f}s/chunk), "
                   f"extrapolated linearly to M={w.M}; host cpu_count={os.cpu_count()}"),
        "acq_pass_value": chunk / med, "fit_s": fit_s,
        "parity_max_rel_vs_gpu_on_timed_chunks": worst,
    }


def suggest_latency(w, X, y, eng, M, reps=3):
    """Median wall time of AcquisitionFunction.suggest(gp, space, n_random=M, n_smart=0|10, fit_gp=True) through
    FloatSpace + HipGPR + the fused acquisition classes (the seams bayes_opt itself calls)."""
    import warnings

    from sklearn.gaussian_process.kernels import Matern, RBF

    from bayesianoptimization_amd import fused_acquisition as A
    from bayesianoptimization_amd.float_space import FloatSpace
    from bayesianoptimization_amd.gpr import HipGPR

    sp = FloatSpace(w.pbounds())
    sp.register_bulk(X, y)
    kern = Matern(nu=2.5, length_scale=w.length_scale) if w.kernel == W.MATERN25 else RBF(length_scale=w.length_scale)
    gp = HipGPR(kernel=kern, alpha=w.noise, normalize_y=True, optimizer=None, engine=eng, incremental=False)  # every call refits
    fn = {W.UCB: lambda: A.UpperConfidenceBound(kappa=w.acq_param), W.EI: lambda: A.ExpectedImprovement(xi=w.acq_param),
          W.POI: lambda: A.ProbabilityOfImprovement(xi=w.acq_param)}[w.acq]()
    res = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for n_smart in (0, 10):
            ts = []
            for rep in range(reps + 1):
                t0 = time.perf_counter()
                fn.suggest(gp, sp, n_random=M, n_smart=n_smart, fit_gp=True, random_state=np.random.RandomState(7 + rep))
                ts.append((time.perf_counter() - t0) * 1e3)
            res[f"n_smart_{n_smart}"] = float(np.median(ts[1:]))      # first call: allocations
    res["note"] = "median of 3 after one warm-up; fixed theta; candidates = the reference's RandomState stream, generated on the device"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None,
                    help="BASELINE.json config; default: C3 on one GPU, C4 (= C3's GP with EI, 2^20 candidates per GPU, "
                         "8 x 2^20 at --gpus 8) when sharded")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: using WORLD_SIZE")
    n_gpus = world

    dist = None
    if world > 1:
        import torch.distributed as dist  # rendezvous plumbing only
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    w = W.ALL[args.config or ("C3" if world == 1 else "C4")]
    X, y, c = W.make_observations(w)
    y_mean, y_std = float(np.mean(y)), float(np.std(y))
    yn = (y - y_mean) / y_std
    y_max = W.feasible_y_max(w, y, c)
    # per-GPU candidate count: C3 = its own M; C4/C5 are quoted as 8-GPU jobs -> one eighth per GPU (weak scaling)
    M = w.M // 8 if w.name in ("C4", "C5") else w.M
    prec = 1 if w.dtype == "f32" else 0
    n_gp = 2 if w.constrained else 1
    if w.constrained:
        c_mean, c_std = float(np.mean(c)), float(np.std(c))
        cn = (c - c_mean) / c_std
        lb_c, ub_c = [-np.inf], [w.constraint_ub]
    else:
        lb_c = ub_c = None
    Xc = W.make_candidates(w.bounds_array(), M, 7 + rank)  # rank r: shard r of a weak-scaled candidate set

    # GPBO_BENCH_DEVICE pins every rank to one device (single-GPU rehearsal of the N > 1 flow; RCCL then
    # refuses the duplicate GPU and the gloo fallback carries the 176-byte exchange)
    dev = int(os.environ.get("GPBO_BENCH_DEVICE", local_rank))
    eng = GpEngine(dev)
    collective = "none"
    allgather = None
    if world > 1:
        # RCCL bootstrap guarded by a watchdog: a hung ncclCommInitRank must not take the scaling run down.
        import threading

        ids = [GpEngine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        state = {"ok": False, "err": None}

        def _init():
            try:
                eng.comm_init(ids[0], world, rank)
                state["ok"] = True
            except Exception as e:  # noqa: BLE001
                state["err"] = repr(e)

        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("GPBO_RCCL_INIT_TIMEOUT", "120")))
        if state["ok"]:
            collective = "rccl-allgather"
        else:
            log(f"[bench] RCCL init failed/hung on rank {rank}: {state['err']}; using gloo for the 176-byte exchange")
            collective = "gloo-allgather(fallback)"
        flags = [None] * world
        dist.all_gather_object(flags, collective)
        if any(f != "rccl-allgather" for f in flags):
            collective = "gloo-allgather(fallback)"
            import torch

            def allgather(vals, idxs):
                tv = [torch.zeros(len(vals), dtype=torch.float64) for _ in range(world)]
                ti = [torch.zeros(len(idxs), dtype=torch.int64) for _ in range(world)]
                dist.all_gather(tv, torch.from_numpy(np.ascontiguousarray(vals)))
                dist.all_gather(ti, torch.from_numpy(np.ascontiguousarray(idxs)))
                return torch.cat(tv).numpy(), torch.cat(ti).numpy()

    sh = ShardedAcquisition(eng, world, rank, allgather)
    sh.set_candidates_local(Xc, offset=rank * M)

    post_ms = [0.0]

    def step():
        eng.fit(X, yn, w.kernel, w.length_scale, w.noise, slot=0, precision=prec)
        eng.posterior(0, y_mean, y_std, fetch=False)
        post_ms[0] = eng.last_timings()["posterior_main"]
        if w.constrained:   # constraint GP in slot 1 (bayes_opt/constraint.py:132-151, 199-221)
            eng.fit(X, cn, W.MATERN25, w.constraint_length_scale, w.noise, slot=1, precision=prec)
            eng.posterior(1, c_mean, c_std, fetch=False)
            post_ms[0] += eng.last_timings()["posterior_main"]
        return sh.argbest(w.acq, w.acq_param, 0.0 if y_max is None else y_max, lb_c, ub_c, k_seeds=10)

    def barrier():
        eng.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    kern_ms = {"fit": 0.0, "posterior_main": 0.0, "posterior_finalize": 0.0, "acq_argbest": 0.0, "kmat": 0.0,
               "cholesky": 0.0, "trtri": 0.0}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best = step()
        tm = eng.last_timings()  # HIP events recorded on the engine's stream around each kernel group
        for k_ in kern_ms:
            kern_ms[k_] += tm[k_] * (n_gp if k_ in ("fit", "kmat", "cholesky", "trtri") else 1)
        kern_ms["posterior_main"] += post_ms[0] - tm["posterior_main"]   # both GPs' posterior launches
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        steps = args.steps
        ms_per_step = elapsed / steps * 1e3
        value = n_gpus * M * steps / elapsed
        main_ms = kern_ms["posterior_main"] / steps
        fl = flops_per_candidate(w.N, w.d, n_gp) * M
        achieved = fl / (main_ms * 1e-3) / 1e12
        peak = FP32_MFMA_PEAK_TFLOPS if prec else FP64_MFMA_PEAK_TFLOPS
        out = {
            "metric": "acquisition candidates/sec (suggest step: GP fit at fixed theta + posterior + acquisition + arg-best)",
            "value": value, "unit": "candidates/s", "n_gpus": n_gpus, "steps": steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if prec else "f64", "data": "synthetic",
            "config": {"workload": f"{w.name}: d={w.d} N={w.N} {W.KERNEL_NAMES[w.kernel]} {W.ACQ_NAMES[w.acq]} "
                                   f"M={M} candidates per GPU, {n_gp} GP(s), fixed length_scale={w.length_scale}, "
                                   f"alpha={w.noise}, k_seeds=10; BASELINE.json config {w.name}",
                       "N": w.N, "d": w.d, "M_per_gpu": M, "M_total": M * n_gpus, "collective": collective},
            "roofline": {"bound": "mfma", "kernel": ("kstar_gen_f32_kernel + posterior_kernel_f32" if prec else
                                                     "kstar_gen_kernel + posterior_kernel_v2<GEN=2>") + " (k* slab + MFMA GEMM)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": None, "avg_launch_ms": main_ms, "flops_per_launch_algorithmic": fl},
            "step_breakdown_ms": {k_: v / steps for k_, v in kern_ms.items()},
            "best": {"index": int(best[0]), "value": float(best[1])},
        }
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this very command
        # (scripts/profile_pmc.sh; FETCH_SIZE and WRITE_SIZE cannot share a pass), summarised in profiles/.
        ppath = os.path.join(ROOT, "profiles", f"r01_pmc_{w.name}_final.json")
        if os.path.exists(ppath):
            try:
                pm = json.load(open(ppath))
                keys = [k_ for k_ in pm if "posterior_kernel_v2" in k_ or "kstar_gen_kernel" in k_]
                key = [k_ for k_ in keys if "posterior_kernel_v2" in k_][0]
                out["roofline"]["traffic"] = sum(pm[k_]["fetch_bytes_corrected_x2"] + pm[k_]["write_bytes"] for k_ in keys)
                out["roofline"]["traffic_note"] = (
                    "HBM-side bytes per launch = 2*FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE from "
                    f"profiles/{os.path.basename(ppath)}; algorithmic compulsory bytes = "
                    f"{(w.d + 1) * 8 * M + w.N * w.N * 4:.3g}; the excess is the k* slab (written once, N*M*8 B, and "
                    "re-read by every row chunk that needs it) plus W re-streamed per candidate tile from L2/Infinity "
                    "Cache — ~1.3 TB/s, a fraction of HBM bandwidth: the path stays MFMA-bound")
                out["roofline"]["mfma_pipe_busy_frac_pmc"] = pm[key].get("mfma_pipe_busy_frac")
            except Exception as e:  # noqa: BLE001
                log(f"[bench] could not read {ppath}: {e!r}")
        gpath = os.path.join(ROOT, "tests", "golden", f"{w.name}.npz")
        if n_gpus == 1 and os.path.exists(gpath) and int(np.load(gpath)["M_evaluated"]) == M and not prec:
            g = np.load(gpath)
            out["parity"] = {"argmin_equals_reference": bool(int(best[0]) == int(g["argmin"])),
                             "top10_equals_reference": bool(np.array_equal(best[2], g["topk_idx"][:10])),
                             "min_rel_err": float(abs(best[1] - float(g["min"])) / abs(float(g["min"])))}
        if n_gpus == 1 and not args.no_cpu_baseline and not w.constrained:
            try:
                _, _, _, _, gpu_ys = eng.acq_argbest(w.acq, w.acq_param, 0.0 if y_max is None else y_max,
                                                     k_seeds=0, return_values=True)
                out["cpu_baseline"] = cpu_baseline(w, X, y, Xc, y_max, gpu_ys)
            except Exception as e:
                log(f"[bench] cpu_baseline failed: {e!r}")
                out["cpu_baseline"] = None
        if n_gpus == 1 and not w.constrained:
            # the other half of BASELINE.json's metric, ms/suggest: whole suggest() calls through the drop-in seams
            # (refit at fixed theta, candidates drawn from the caller's RandomState on the device, posterior, acquisition,
            # arg-best; with the reference's default 10 local searches and without) — outside the timed region above
            try:
                out["suggest_ms"] = suggest_latency(w, X, y, eng, M)
            except Exception as e:  # noqa: BLE001
                log(f"[bench] suggest latency failed: {e!r}")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        if th.is_alive():  # a hung RCCL bootstrap thread: leave without running its destructors
            sys.stdout.flush()
            os._exit(0)
    eng.close()


if __name__ == "__main__":
    main()
