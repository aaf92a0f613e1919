"""First-light diagnostics on a real MI355X: layout probe, peaks, parity vs the oracle, timings.

    python tests/gpu_first_light.py [--big]      (writes gpurun_out/first_light.json)
Development aid (not part of the product or the test-suite); uses the oracle as the checker.
"""
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianoptimization_amd import workloads as W  # noqa: E402
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402

out = {}


def rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def section(name):
    def deco(fn):
        t0 = time.time()
        try:
            out[name] = fn()
        except Exception as e:  # keep going: one call must report as much as possible
            out[name] = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
        out[name + "_wall_s"] = round(time.time() - t0, 3)
        print(name, json.dumps(out[name], default=str)[:1500], flush=True)
        return fn
    return deco


eng = GpEngine(0, debug=True)      # gpbo_debug_gemm: the debug build (same sources as libgpbo.so)


@section("device")
def _():
    return eng.device_info()


@section("gemm_layout")
def _():
    rng = np.random.RandomState(0)
    res = {}
    for (m, n, k, bt) in [(64, 64, 16, False), (128, 192, 64, False), (128, 64, 48, True)]:
        A = rng.randn(m, k); B = rng.randn(n, k) if bt else rng.randn(k, n); C0 = rng.randn(m, n)
        ref = 0.5 * (A @ (B.T if bt else B)) - 2.0 * C0
        got = eng.debug_gemm(A, B, C0, alpha=0.5, beta=-2.0, b_trans=bt)
        res[f"{m}x{n}x{k}_bt{int(bt)}"] = rel(got, ref)
    return res


@section("peaks")
def _():
    return {"mfma_f64_tflops": [eng.mfma_f64_peak(20000) for _ in range(3)],
            "hbm_copy_gbps": [eng.hbm_copy_peak(1 << 30) for _ in range(2)]}


def parity_case(w, ls, M, check_fit=True, c_ls=None):
    X, y, c = W.make_observations(w)
    Xc = W.make_candidates(w.bounds_array(), M, 7)
    r = {"N": w.N, "d": w.d, "M": M}
    t0 = time.time(); gp = O.fit_fixed_theta(w.kernel, X, y, ls, w.noise); r["oracle_fit_s"] = round(time.time() - t0, 3)
    yn, ym, ys_ = O.normalize_targets(y)
    t0 = time.time(); eng.fit(X, yn, w.kernel, ls, w.noise, slot=0); r["gpu_fit_wall_s"] = round(time.time() - t0, 4)
    r["fit_timings_ms"] = eng.last_timings()
    if check_fit:
        K = O.kernel_matrix(w.kernel, X, None, ls); K[np.diag_indices_from(K)] += w.noise
        r["K_rel"] = rel(eng.get_K(w.N), K)
        r["L_rel"] = rel(eng.get_L(w.N), gp.L)
        Li = np.linalg.inv(gp.L)
        r["Linv_rel"] = rel(eng.get_Linv(w.N), Li)
    r["alpha_rel"] = rel(eng.get_alpha(w.N), gp.alpha)
    Mo = min(M, 8192)
    t0 = time.time(); mu_o, sd_o = O.predict(gp, Xc[:Mo]); r["oracle_predict_s"] = round(time.time() - t0, 3)
    eng.set_candidates(Xc)
    t0 = time.time(); mu, sd = eng.posterior(0, ym, ys_); r["gpu_post_wall_s"] = round(time.time() - t0, 4)
    tm = eng.last_timings(); r["post_main_ms"] = tm["posterior_main"]; r["post_final_ms"] = tm["posterior_finalize"]
    fl = O.flops_per_candidate(w.N, w.d) * M
    r["post_tflops_algorithmic"] = fl / (tm["posterior_main"] * 1e-3) / 1e12
    r["mu_rel"] = rel(mu[:Mo], mu_o); r["sd_rel"] = rel(sd[:Mo], sd_o)
    r["mu_maxabs"] = float(np.max(np.abs(mu[:Mo] - mu_o))); r["sd_maxabs"] = float(np.max(np.abs(sd[:Mo] - sd_o)))
    y_max = W.feasible_y_max(w, y, c)
    cons = None; lb = ub = None
    if w.constrained:
        cgp = O.fit_fixed_theta(W.MATERN25, X, c, c_ls, w.noise)
        cn, cm, cs = O.normalize_targets(c)
        eng.fit(X, cn, W.MATERN25, c_ls, w.noise, slot=1)
        eng.posterior(1, cm, cs, fetch=False)
        cons = ([cgp], [-np.inf], [w.constraint_ub]); lb = [-np.inf]; ub = [w.constraint_ub]
    ys_o = O.neg_acquisition(gp, Xc[:Mo], w.acq, w.acq_param, y_max if y_max is not None else 0.0, cons)
    t0 = time.time()
    bi, bv, si, sv, ys = eng.acq_argbest(w.acq, w.acq_param, y_max, lb, ub, k_seeds=10, return_values=True)
    r["gpu_acq_wall_s"] = round(time.time() - t0, 4); r["acq_ms"] = eng.last_timings()["acq_argbest"]
    r["ys_rel"] = rel(ys[:Mo], ys_o)
    r["argmin_gpu"] = int(bi); r["argmin_self"] = int(np.argmin(ys)); r["min_gpu"] = float(bv)
    r["topk_match_self"] = bool(np.array_equal(si, np.argsort(ys, kind="stable")[:10]))
    if Mo == M:
        oi, ov, os_ = O.arg_best(ys_o, 10)
        r["argmin_oracle"] = oi; r["topk_match_oracle"] = bool(np.array_equal(si, os_))
    return r


@section("parity_small")
def _():
    res = {}
    for name, ls, M, cls in [("P1", 0.4, 4096, None), ("P2", 0.6, 4096, None), ("C5S", 0.5, 8192, 0.7)]:
        res[name] = parity_case(W.ALL[name], ls, M, True, cls)
    return res


@section("parity_C2")
def _():
    return parity_case(W.C2, 1.0, W.C2.M, True)


@section("golden_C1_F1")
def _():
    res = {}
    for name in ("C1", "F1", "C2"):
        w = W.ALL[name]
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        X, y, c = W.make_observations(w)
        yn, ym, ys_ = O.normalize_targets(y)
        eng.fit(X, yn, w.kernel, g["length_scale"], w.noise)
        Xc = W.make_candidates(w.bounds_array(), w.M, 7)
        eng.set_candidates(Xc)
        mu, sd = eng.posterior(0, ym, ys_)
        S = len(g["mu"])
        bi, bv, si, sv, ys = eng.acq_argbest(w.acq, w.acq_param, float(np.max(y)), k_seeds=16, return_values=True)
        res[name] = {"alpha_rel": rel(eng.get_alpha(w.N), g["alpha"]), "mu_rel": rel(mu[:S], g["mu"]),
                     "sd_rel": rel(sd[:S], g["sd"]), "ys_rel": rel(ys[:S], g["ys"]),
                     "argmin": [int(bi), int(g["argmin"])], "topk_equal": bool(np.array_equal(si, g["topk_idx"]))}
    return res


if "--big" in sys.argv:
    @section("C3_timing")
    def _():
        w = W.C3
        X, y, c = W.make_observations(w)
        yn, ym, ys_ = O.normalize_targets(y)
        res = {}
        for rep in range(2):
            t0 = time.time(); eng.fit(X, yn, w.kernel, w.length_scale, w.noise); res[f"fit_wall_s_{rep}"] = round(time.time() - t0, 4)
            res[f"fit_ms_{rep}"] = eng.last_timings()
        gp = O.fit_fixed_theta(w.kernel, X, y, w.length_scale, w.noise)
        res["alpha_rel"] = rel(eng.get_alpha(w.N), gp.alpha)
        res["L_rel"] = rel(eng.get_L(w.N), gp.L)
        for M in (1 << 14, 1 << 17, 1 << 20):
            Xc = W.make_candidates(w.bounds_array(), M, 7)
            eng.set_candidates(Xc)
            for rep in range(2):
                t0 = time.time(); mu, sd = eng.posterior(0, ym, ys_); wall = time.time() - t0
                tm = eng.last_timings()
                fl = O.flops_per_candidate(w.N, w.d) * M
                res[f"M{M}_rep{rep}"] = {"wall_s": round(wall, 4), "main_ms": tm["posterior_main"], "final_ms": tm["posterior_finalize"],
                                         "tflops": fl / (tm["posterior_main"] * 1e-3) / 1e12,
                                         "cand_per_s": M / (tm["posterior_main"] * 1e-3)}
            mu_o, sd_o = O.predict(gp, Xc[:4096])
            res[f"M{M}_mu_rel"] = rel(mu[:4096], mu_o); res[f"M{M}_sd_rel"] = rel(sd[:4096], sd_o)
            bi, bv, si, sv, _ys = eng.acq_argbest(w.acq, w.acq_param, 0.0, k_seeds=10)
            res[f"M{M}_acq_ms"] = eng.last_timings()["acq_argbest"]; res[f"M{M}_argmin"] = int(bi)
        g = np.load(os.path.join(ROOT, "tests", "golden", "C3.npz")) if os.path.exists(os.path.join(ROOT, "tests", "golden", "C3.npz")) else None
        if g is not None:
            res["golden_argmin"] = int(g["argmin"]); res["golden_topk"] = g["topk_idx"][:10].tolist(); res["gpu_topk"] = si.tolist()
        return res

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "first_light.json"), "w"), indent=1, default=str)
print("DONE")
