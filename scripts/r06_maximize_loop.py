"""The loop users actually run: BayesianOptimization.maximize() — one suggest() per new observation with the reference's defaults
(n_random = 10 000 candidates, n_smart = 10 local searches, the theta search of GaussianProcessRegressor(Matern(2.5), alpha = 1e-6,
normalize_y = True, n_restarts_optimizer = 5) in EVERY fit) while N grows by one per step, here from 16 to 528.

    python scripts/r06_maximize_loop.py [--no-warm] > profiles/r06_maximize_loop.json

What it replaces: /root/reference/bayes_opt/bayesian_optimization.py:348-391 (maximize -> suggest -> acquisition.suggest, :323-333,
acquisition.py:116-169) at the defaults of :124-130.  The GPU box has no /root/reference: the loop runs through the seams bayes_opt
calls (FloatSpace + HipGPR + the fused acquisition class), as bench.py's suggest_ms does.  Beside it, on the host
cores: the same acquisition class over scikit-learn's own GaussianProcessRegressor with the same configuration (what the reference
runs) on a SAMPLE of the steps (every 32nd N; the whole CPU loop would be minutes), from the device loop's observations at that N.
Round 6: the first-use costs are paid before the loop by dropin.warm_up (what accelerate(optimizer) does by default; --no-warm
leaves them inside the loop as in round 5), `spikes` lists every step above 3x its band's median, and the host column is
measured twice — BLAS on all host threads (what a default sklearn does; oversubscribed on sub-millisecond problems) and held to 8
threads (threadpoolctl.threadpool_limits(8)).
Per step: wall ms of the suggest() call (fit with theta search + candidates + posterior + acquisition + local searches), lockstep
rounds of the theta search and the length scale found.
"""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sklearn.gaussian_process import GaussianProcessRegressor  # noqa: E402
from sklearn.gaussian_process.kernels import Matern  # noqa: E402

from bayesianoptimization_amd import fused_acquisition as A  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402
from bayesianoptimization_amd.float_space import FloatSpace  # noqa: E402
from bayesianoptimization_amd.gpr import HipGPR  # noqa: E402

D = 4
N0, N1 = 16, 528
CPU_EVERY = 32


def black_box(x):
    x = np.asarray(x, dtype=np.float64)
    return float(-np.sum((x - 0.3) ** 2) + 0.5 * np.sin(5.0 * x[0]) * np.cos(3.0 * x[1]))


def main():
    warnings.simplefilter("ignore")
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 1      # (1: the run profiles/ quotes)
    eng = GpEngine(0, debug="--debug" in sys.argv)      # --debug: libgpbo_dbg.so (reads the A/B switches, e.g. GPBO_LML_GRAPH=0)
    no_cpu = "--no-cpu" in sys.argv
    warm_s = None
    if "--no-warm" not in sys.argv:
        from bayesianoptimization_amd.dropin import warm_up
        warm_s = warm_up(eng, np.array([[0.0, 1.0]] * D))
    pb = {f"x{j}": (0.0, 1.0) for j in range(D)}
    sp = FloatSpace(pb)
    rng = np.random.RandomState(seed)
    X0 = rng.uniform(size=(N0, D))
    sp.register_bulk(X0, np.array([black_box(x) for x in X0]))
    gp = HipGPR(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5,
                random_state=np.random.RandomState(seed), engine=eng)
    fn = A.UpperConfidenceBound(kappa=2.576)                       # the reference's default acquisition
    rs = np.random.RandomState(6 + seed)
    rows, cpu_rows = [], []
    t_all = time.perf_counter()
    while len(sp) < N1:
        N = len(sp)
        t0 = time.perf_counter()
        x = fn.suggest(gp, sp, n_random=10_000, n_smart=10, fit_gp=True, random_state=rs)
        ms = (time.perf_counter() - t0) * 1e3
        rows.append({"N": N, "ms": round(ms, 3), "theta_search_rounds": int(getattr(gp, "theta_search_rounds_", 0)),
                     "lml_evaluations": int(getattr(gp, "theta_search_evals_", 0)),
                     "length_scale": float(np.exp(gp.kernel_.theta[0]))})
        if (N - N0) % CPU_EVERY == 0 and not no_cpu:
            # the reference's own estimator on the host cores, same observations, same configuration, same acquisition class
            sk = GaussianProcessRegressor(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5,
                                          random_state=np.random.RandomState(1))
            fn_c = A.UpperConfidenceBound(kappa=2.576)
            fn_c.device_polish = False
            t0 = time.perf_counter()
            fn_c.suggest(sk, sp, n_random=10_000, n_smart=10, fit_gp=True, random_state=np.random.RandomState(7))
            row = {"N": N, "ms": round((time.perf_counter() - t0) * 1e3, 2), "length_scale": float(np.exp(sk.kernel_.theta[0]))}
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=8):
                sk8 = GaussianProcessRegressor(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5,
                                               random_state=np.random.RandomState(1))
                fn_8 = A.UpperConfidenceBound(kappa=2.576)
                fn_8.device_polish = False
                t0 = time.perf_counter()
                fn_8.suggest(sk8, sp, n_random=10_000, n_smart=10, fit_gp=True, random_state=np.random.RandomState(7))
                row["ms_8_threads"] = round((time.perf_counter() - t0) * 1e3, 2)
            cpu_rows.append(row)
            print(N, rows[-1], cpu_rows[-1], file=sys.stderr, flush=True)
        sp.register(x, black_box(x))
    total_s = time.perf_counter() - t_all - sum(r["ms"] + r.get("ms_8_threads", 0.0) for r in cpu_rows) * 1e-3
    ms = np.array([r["ms"] for r in rows])
    Ns = np.array([r["N"] for r in rows])

    def band(lo, hi):
        m = (Ns >= lo) & (Ns < hi)
        if not m.any():
            return {"N": f"{lo}..{hi - 1}", "steps": 0}
        return {"N": f"{lo}..{hi - 1}", "steps": int(m.sum()), "median_ms": float(np.median(ms[m])), "mean_ms": float(np.mean(ms[m])),
                "max_ms": float(np.max(ms[m])), "max_over_median": float(np.max(ms[m]) / np.median(ms[m]))}

    def spikes():
        out = []
        for lo, hi in ((16, 65), (65, 129), (129, 257), (257, 385), (385, 528)):
            m = (Ns >= lo) & (Ns < hi)
            if m.any():
                med = float(np.median(ms[m]))
                out += [{"N": int(n), "ms": float(t), "band_median_ms": med} for n, t in zip(Ns[m], ms[m]) if t > 3.0 * med]
        return out

    if no_cpu:
        print(json.dumps({"device": {"steps": len(rows), "total_s": float(np.sum(ms) * 1e-3), "warm_up_s": warm_s, "spikes_over_3x_band_median": spikes(),
                                     "bands": [band(16, 65), band(65, 129), band(129, 257), band(257, 385), band(385, 528)]}}, indent=1))
        return
    cpu_at = {r["N"]: r["ms"] for r in cpu_rows}
    cpu8_at = {r["N"]: r["ms_8_threads"] for r in cpu_rows}
    dev_at = {r["N"]: r["ms"] for r in rows}
    out = {
        "what": __doc__.strip().split("\n")[0],
        "config": {"d": D, "N": [N0, N1], "n_random": 10_000, "n_smart": 10, "acquisition": "UCB(kappa=2.576)",
                   "gp": "Matern(2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5 (theta search in every fit)",
                   "black_box": "-sum((x - 0.3)^2) + 0.5 sin(5 x0) cos(3 x1) on [0, 1]^4"},
        "device": {"steps": len(rows), "total_s": float(np.sum(ms) * 1e-3), "loop_wall_s_incl_registering": float(total_s),
                   "warm_up_s": warm_s, "spikes_over_3x_band_median": spikes(),
                   "bands": [band(16, 65), band(65, 129), band(129, 257), band(257, 385), band(385, 528)]},
        "host_sklearn_sampled": {"every": CPU_EVERY, "cores": os.cpu_count(), "rows": cpu_rows,
                                 "ratio_host_over_device": [{"N": n, "host_ms": cpu_at[n], "host_ms_8_threads": cpu8_at[n], "device_ms": dev_at[n],
                                                             "ratio": round(cpu_at[n] / dev_at[n], 1),
                                                             "ratio_8_threads": round(cpu8_at[n] / dev_at[n], 1)} for n in sorted(cpu_at)],
                                 "extrapolated_total_s": float(np.interp(Ns, sorted(cpu_at), [cpu_at[n] for n in sorted(cpu_at)]).sum() * 1e-3),
                                 "extrapolated_total_s_8_threads": float(np.interp(Ns, sorted(cpu8_at), [cpu8_at[n] for n in sorted(cpu8_at)]).sum() * 1e-3)},
        "per_step": rows,
        "best_target": float(np.max(sp.target)),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
