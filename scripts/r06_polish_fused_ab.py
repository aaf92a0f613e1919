"""gpbo_polish_seeds: the runs as ONE launch (csrc/polish_fused.hip: polish_rows_kernel, thread = training point; W in LDS up to
NP = 128, streamed from memory above) against the lockstep rounds (csrc/polish.hip), same seeds, on libgpbo_dbg.so.  (The
eight-wave kernel of round 5 was measured beside both until it was removed: profiles/r06_polish_fused_ab_three_arms.json.)

    python scripts/r06_polish_fused_ab.py > profiles/r06_polish_fused_ab.json

Per (N, d, acquisition): 10 seeds = the best of 10 000 uniform candidates (what suggest() hands the stage), wall ms of the call
(median of 15 after 3 warm-ups), the longest run's evaluations (= lockstep rounds), and whether the two paths returned the same bits.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd import workloads as W  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

UCB, EI = 0, 1


def eval_us(eng, acq, param, y_max, ym, ys, pts, repeat=200):
    """microseconds per evaluation inside the one launch: `repeat` evaluations of each point in one kernel (debug entry)"""
    from bayesianoptimization_amd import _lib
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    n, d = pts.shape
    out = np.empty((n, 4 + 3 * d))
    ts = []
    for rep in (1, repeat, repeat, repeat):
        t0 = time.perf_counter()
        eng._check(eng._lib.gpbo_debug_polish_eval(eng._h, int(acq), float(param), float(y_max), float(ym), float(ys), _lib.dptr(pts), n, d,
                                                   rep, _lib.dptr(out)))
        ts.append((time.perf_counter() - t0) * 1e6)
    return (min(ts[1:]) - ts[0]) / (repeat - 1)


def main():
    os.environ["GPBO_POLISH_FUSED_MAX_NP"] = "512"      # (the product's limit is 384: this table is where that number comes from)
    eng = GpEngine(0, debug=True)
    rows = []
    for d in (4, 8):
        for N in (17, 32, 64, 100, 128, 143, 256, 384, 512):
            rng = np.random.RandomState(N + d)
            X = rng.uniform(size=(N, d))
            y = np.exp(-np.sum((X - 0.5) ** 2, axis=1)) + 0.01 * rng.standard_normal(N)
            ym, ys = float(y.mean()), float(y.std())
            yn = (y - ym) / ys
            ls = 0.3 * np.sqrt(d)
            eng.fit(X, yn, W.MATERN25, ls, 1e-6, slot=0)
            cand = rng.uniform(size=(10_000, d))
            eng.set_candidates(cand)
            box = np.array([[0.0, 1.0]] * d)
            for acq, param, y_max in ((UCB, 2.576, 0.0), (EI, 0.01, float(y.max()))):
                eng.posterior(slot=0, y_mean=ym, y_std=ys, fetch=False)
                best = eng.acq_argbest(acq, param, y_max, None, None, k_seeds=10)
                seeds = np.ascontiguousarray(cand[np.asarray(best[2], dtype=np.int64)])
                out = {}
                for name, env in (("lockstep", "0"), ("one_launch", None)):
                    if env is None:
                        os.environ.pop("GPBO_POLISH_FUSED", None)
                    else:
                        os.environ["GPBO_POLISH_FUSED"] = env
                    ts = []
                    for it in range(18):
                        t0 = time.perf_counter()
                        res = eng.polish_seeds(acq, param, y_max, None, None, [ym], [ys], seeds, box)
                        ts.append((time.perf_counter() - t0) * 1e3)
                    out[name] = {"ms": float(np.median(ts[3:])), "rounds": int(res[3]), "evals_mean": float(np.mean(eng.last_polish["nfev"])),
                                 "res": res}
                os.environ.pop("GPBO_POLISH_FUSED", None)
                a, b = out["lockstep"]["res"], out["one_launch"]["res"]
                same = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]))
                ev = eval_us(eng, acq, param, y_max, ym, ys, seeds) if N <= 512 else None
                rows.append({"N": N, "d": d, "acq": "ucb" if acq == UCB else "ei", "one_launch_us_per_bare_evaluation": None if ev is None else round(ev, 2),
                             "lockstep_ms": round(out["lockstep"]["ms"], 4),
                             "one_launch_evals_mean": out["one_launch"]["evals_mean"],
                             "one_launch_ms": round(out["one_launch"]["ms"], 4), "rounds": out["lockstep"]["rounds"],
                             "one_launch_longest_run_evals": out["one_launch"]["rounds"], "evals_mean": out["lockstep"]["evals_mean"],
                             "same_bits": same, "best_f": [float(a[1].min()), float(b[1].min())],
                             "us_per_round_lockstep": round(1e3 * out["lockstep"]["ms"] / max(out["lockstep"]["rounds"], 1), 2),
                             "us_per_eval_one_launch": round(1e3 * out["one_launch"]["ms"] / max(out["one_launch"]["rounds"], 1), 2)})
                print(rows[-1], file=sys.stderr, flush=True)
    print(json.dumps({"what": __doc__.strip().split("\n")[0], "note": "the product takes the one launch up to NP = 384 (N = 512: a 48-evaluation EI run loses to the lockstep rounds)",
                      "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
