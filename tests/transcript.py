"""Replay of the call transcripts under tests/golden/transcript_*.npz (written by oracle/gen_transcript.py: the REAL bayes_opt
driving accelerate() over a recording, oracle-backed engine).  Test infrastructure: `replay` feeds every recorded call, in order
and with the recorded arguments, to an engine — the real GpEngine on the GPU (tests/test_gpu_transcript.py) or a fresh oracle
engine on the CPU (tests/test_transcript_host.py, which pins the fixture itself) — and hands what came back, next to what the
driver saw when the transcript was recorded, to a checker per call kind."""
import hashlib
import json
import os

import numpy as np

from conftest import GOLDEN_DIR, elementwise_err

NAMES = ("float_ucb", "constrained_ei", "mixed_space", "gphedge", "constant_liar")


class Transcript:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, f"transcript_{name}.npz"))
        self.name = name
        self.calls = json.loads(bytes(z["__calls__"]).decode())
        self.meta = json.loads(bytes(z["__meta__"]).decode())
        self.arrays = {k: z[k] for k in z.files if not k.startswith("__")}

    # -- decoding ----------------------------------------------------------------------------------------------------------
    def array(self, ref):
        a = np.array(self.arrays[ref["ref"]], dtype=np.dtype(ref["dtype"])).reshape(ref["shape"])
        if ref["order"] == "F":
            return np.asfortranarray(a)
        if ref["order"] == "strided" and a.ndim >= 1 and a.shape[-1] > 0:
            wide = np.zeros(a.shape[:-1] + (2 * a.shape[-1],), dtype=a.dtype)      # same values, every second slot of a wider buffer
            wide[..., ::2] = a
            return wide[..., ::2]
        return np.ascontiguousarray(a)

    def value(self, v):
        if v is None or isinstance(v, (bool, int, str)):
            return v
        if "f" in v:
            return float.fromhex(v["f"])
        if "rng" in v:
            rs = np.random.RandomState()
            rs.set_state(("MT19937", self.array(v["rng"]), v["pos"], v["has_gauss"], float.fromhex(v["cached"])))
            return rs
        if "seq" in v:
            items = [self.value(x) for x in v["seq"]]
            return tuple(items) if v["tuple"] else items
        return self.array(v)


def same_rng(a, b) -> bool:
    sa, sb = a.get_state(legacy=True), b.get_state(legacy=True)
    return bool(np.array_equal(sa[1], sb[1]) and sa[2] == sb[2])


# ---- checkers: bars per call kind ------------------------------------------------------------------------------------------
class Bars:
    """The bar of every kind of return value: L 1e-10, alpha 1e-8, mu / sd 1e-5 per element and 1e-8 in
    the max norm, LML 1e-10 / gradient 1e-7, arg-best and seed indices exact where the recorded values' gaps exceed twice the
    value bound, candidates and RandomState positions bitwise.  `exact=True` (the oracle replaying itself): everything bitwise.
    Bars scale with kappa = cond_2(K) above 1e6 (recorded with every fit and every theta of the search): what fp64 holds of K^-1 y."""

    def __init__(self, exact=False):
        self.exact = exact
        self.worst = {}
        self.exact_argbest = 0
        self.loose_argbest = 0

    def note(self, kind, err):
        self.worst[kind] = max(self.worst.get(kind, 0.0), float(err))

    def close(self, kind, got, want, bar, scale=None):
        got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
        assert got.shape == want.shape, (kind, got.shape, want.shape)
        if self.exact:
            assert np.array_equal(got, want, equal_nan=True), kind
            return
        if want.size == 0:
            return
        s = float(np.max(np.abs(want))) if scale is None else float(scale)
        err = float(np.max(np.abs(got - want))) / max(s, 1e-300)
        self.note(kind, err / bar)
        assert err <= bar, f"{kind}: {err:.3e} > {bar:.1e}"


def _kappa(T, extra):
    return float(extra["kappa"])


def replay(T: Transcript, eng, bars: Bars, oracle=None):
    """Run every recorded call on `eng`.  `oracle`: module used to re-evaluate a local search's returned points (None: skip)."""
    factors = {}     # slot -> kappa of the factor the slot holds
    for n_call, c in enumerate(T.calls):
        name, a, ex = c["name"], {k: T.value(v) for k, v in c["args"].items()}, c["extra"]
        want = T.value(c["ret"])
        where = f"{T.name}[{n_call}] {name}"
        if name in ("fit", "fit_append"):
            getattr(eng, name)(**a)
            slot, N = a.get("slot", 0), ex["N"]
            kap = _kappa(T, ex)
            factors[slot] = kap
            amp = max(1.0, kap / 1e6)
            L = eng.get_L(N, slot)
            assert not np.triu(L, 1).any(), where
            bars.close("L_diag", np.diag(L), T.array(ex["L_diag"]), 1e-10 * amp)
            bars.close("L_lastrow", L[N - 1], T.array(ex["L_lastrow"]), 1e-10 * amp, scale=np.max(np.abs(T.array(ex["L_diag"]))))
            fro = float(np.linalg.norm(L))     # (a sum: its last bit depends on the memory order of the array it is taken over)
            assert abs(fro - float.fromhex(ex["L_fro"]["f"])) <= (1e-14 if bars.exact else 1e-10 * amp) * fro, where
            if "L" in ex:
                bars.close("L", L, T.array(ex["L"]), 1e-10 * amp)
            bars.close("alpha", eng.get_alpha(N, slot), T.array(ex["alpha"]), 1e-8 * amp)
        elif name in ("lml", "lml_batch"):
            got = getattr(eng, name)(**a)
            pairs = zip(got, want) if name == "lml_batch" else [(got, want)]
            for ((v, g), (v0, g0)), kap in zip(pairs, ex["kappa"]):
                if not np.isfinite(kap) or not np.isfinite(v0):
                    assert not np.isfinite(v) or bars.exact is False, where      # K not positive definite for the oracle: -inf there
                    continue
                amp = max(1.0, kap / 1e6)
                bars.close("lml", [v], [v0], 1e-10 * amp, scale=max(abs(v0), 1.0))
                bars.close("lml_grad", g, g0, 1e-7 * amp, scale=max(float(np.max(np.abs(g0))), 1.0))
        elif name in ("get_L", "get_alpha"):
            amp = max(1.0, factors.get(a.get("slot", 0), 1.0) / 1e6)
            bars.close(name, getattr(eng, name)(**a), want, (1e-10 if name == "get_L" else 1e-8) * amp)
        elif name == "set_candidates":
            eng.set_candidates(a["Xc"])
        elif name == "generate_candidates_like":
            rs = a["random_state"]
            start = np.random.RandomState()
            start.set_state(rs.get_state())
            eng.generate_candidates_like(a["M"], a["lo"], a["hi"], rs)
            after = T.value(c["rng_after"]["random_state"])
            assert same_rng(rs, after), f"{where}: the caller's RandomState is not where the reference leaves it"
            M, d = ex["M"], ex["d"]
            rows = np.concatenate([eng.get_candidate_rows(np.arange(lo_, min(lo_ + 4096, M)), d) for lo_ in range(0, M, 4096)])
            ref = np.column_stack([start.uniform(a["lo"][t], a["hi"][t], M) for t in range(d)])      # target_space.py:565-603
            assert np.array_equal(rows, ref), f"{where}: candidates are not the reference's stream"
            assert hashlib.sha1(np.ascontiguousarray(rows).tobytes()).hexdigest() == ex["checksum"], where
        elif name == "get_candidate_rows":
            got = eng.get_candidate_rows(a["idx"], a["d"])
            assert np.array_equal(got, want), where
        elif name == "posterior":
            slot, amp = a.get("slot", 0), max(1.0, factors.get(a.get("slot", 0), 1.0) / 1e6)
            if a.get("fetch", True):
                mu, sd = eng.posterior(**a)
                mu0, sd0 = want
            else:
                mu, sd = eng.posterior(slot, a["y_mean"], a["y_std"], True)
                idx = T.array(ex["sample_idx"])
                mu, sd, mu0, sd0 = mu[idx], sd[idx], T.array(ex["sample_mu"]), T.array(ex["sample_sd"])
            _check_posterior(bars, mu, sd, mu0, sd0, a["y_std"], amp)
        elif name == "predict":
            amp = max(1.0, factors.get(a.get("slot", 0), 1.0) / 1e6)
            mu, sd = eng.predict(**a)
            _check_posterior(bars, mu, sd, want[0], want[1], a["y_std"], amp)
        elif name == "predict_cov":
            amp = max(1.0, factors.get(a.get("slot", 0), 1.0) / 1e6)
            mu, cov = eng.predict_cov(**a)
            bars.close("cov_mu", mu, want[0], 1e-8 * amp, scale=max(float(np.max(np.abs(want[0]))), a["y_std"]))
            bars.close("cov", cov, want[1], 1e-8 * amp, scale=a["y_std"] ** 2)
        elif name == "predict_grad":
            amp = max(1.0, factors.get(a.get("slot", 0), 1.0) / 1e6)
            got = eng.predict_grad(**a)
            _check_posterior(bars, got[0], got[1], want[0], want[1], a["y_std"], amp)
            bars.close("dmu", got[2], want[2], 1e-7 * amp, scale=max(float(np.max(np.abs(want[2]))), a["y_std"]))
            bars.close("dsd", got[3], want[3], 1e-6 * amp, scale=max(float(np.max(np.abs(want[3]))), a["y_std"]))
        elif name == "acq_argbest":
            got = eng.acq_argbest(**a)
            _check_argbest(T, bars, got, want, a, ex, where)
        elif name == "polish_seeds":
            xs, fs, status, _ = eng.polish_seeds(**a)
            _check_polish(bars, xs, fs, status, want, a, eng, where, oracle)
        elif name == "take_negative_variance_flag":
            got = eng.take_negative_variance_flag()
            if bars.exact:
                assert got == want, where
        else:      # pragma: no cover
            raise AssertionError(f"transcript call {name!r} has no replay")


def _check_posterior(bars, mu, sd, mu0, sd0, y_std, amp):
    if bars.exact:
        assert np.array_equal(mu, mu0) and np.array_equal(sd, sd0)
        return
    bars.close("mu", mu, mu0, 1e-8 * amp, scale=max(float(np.max(np.abs(mu0))), y_std))
    bars.close("sd", sd, sd0, 1e-8 * amp, scale=max(float(np.max(np.abs(sd0))), 1e-300))
    # north_star's bound per element; a sigma below sqrt(eps) * s_y is rounding of 1 - sum v^2 (a candidate ON a training point:
    # the local searches' iterates converge to such points), which no arithmetic reproduces to a relative bound
    live = np.asarray(sd0) > 1e-6 * y_std
    e_sd, e_mu = elementwise_err(np.asarray(sd)[live], np.asarray(sd0)[live], mu, mu0, y_std)
    bars.note("sd_elementwise", e_sd / (1e-5 * amp))
    bars.note("mu_elementwise", e_mu / (1e-5 * amp))
    assert e_sd <= 1e-5 * amp and e_mu <= 1e-5 * amp, (e_sd, e_mu)


def _check_argbest(T, bars, got, want, a, ex, where):
    bi, bv, picks, vals, _ = got
    bi0, bv0, picks0, vals0, _ = want
    if bars.exact:
        assert bi == bi0 and (bv == bv0 or (np.isnan(bv) and np.isnan(bv0))) and np.array_equal(picks, picks0)
        assert np.array_equal(vals, vals0, equal_nan=True)
        return
    rng = max(float.fromhex(ex["range"]["f"]), 1e-300)
    bound = 1e-8 * rng
    head = T.array(ex["head_val"])
    k = a.get("k_seeds", 0)
    if ex["n_nan"] == 0 and head.size >= 2:
        gaps = np.diff(head[:k + 2])
        # a gap is decidable when it is wide (> twice the value bound) or an exact tie of bitwise equal values, which both sides
        # break by the lowest index (a plateau: a theta search that ended at the lower bound leaves K = I, every candidate the same)
        if head.size >= 2 and (gaps[0] > 2 * bound or (gaps[0] == 0.0 and vals[0] == vals[min(1, len(vals) - 1)])):
            assert bi == bi0, f"{where}: arg-best {bi} != {bi0} (gap {gaps[0]:.3e})"
        # seeds: exact wherever every gap up to and including the one behind the k-th value is decidable
        tie_ok = np.concatenate([np.asarray(vals)[1:] == np.asarray(vals)[:-1], [True]])[:k] if k else np.zeros(0, bool)
        if k and gaps.size >= k and np.all((gaps[:k] > 2 * bound) | ((gaps[:k] == 0.0) & tie_ok)):
            assert np.array_equal(np.asarray(picks), np.asarray(picks0)), where
            bars.exact_argbest += 1
        else:
            bars.loose_argbest += 1
        bars.close("argbest_value", [bv], [bv0], 1e-8, scale=rng)
        if k:
            bars.close("seed_values", vals, vals0, 1e-8, scale=rng)
    else:
        assert np.isnan(bv) == np.isnan(bv0), where


def _check_polish(bars, xs, fs, status, want, a, eng, where, oracle=None):
    """Local searches: parity is statistical — the best converged run is as good as the best of SciPy's L-BFGS-B
    runs the driver saw (to 1e-5 of the acquisition's scale), every returned point lies in the box, and the value returned with a
    point is the acquisition at that point."""
    xs0, fs0, status0, _ = want
    if bars.exact:
        assert np.array_equal(xs, xs0) and np.array_equal(fs, fs0) and np.array_equal(status, status0)
        return
    box = np.asarray(a["box"], dtype=np.float64)
    assert xs.shape == xs0.shape and np.all(xs >= box[:, 0] - 1e-12) and np.all(xs <= box[:, 1] + 1e-12), where
    ok, ok0 = (status < 2) & np.isfinite(fs), (status0 < 2) & np.isfinite(fs0)
    if ok0.any():
        assert ok.any(), f"{where}: no local search converged (the reference's did)"
        best, best0 = float(np.min(fs[ok])), float(np.min(fs0[ok0]))
        scale = max(abs(best0), float(np.max(np.abs(fs0[ok0]))), 1e-12)
        bars.note("polish_best_minus_reference", max(best - best0, 0.0) / (1e-5 * scale))
        bars.polish = getattr(bars, "polish", []) + [(best, best0)]
        assert best <= best0 + 1e-5 * scale, f"{where}: best local search {best:.12g} worse than the reference's {best0:.12g}"
    if oracle is not None and ok.any():
        # f(x) as returned = -acquisition [x constraint probability] at the returned x, from the engine's own posterior there
        O = oracle
        ym = 0.0 if a["y_max"] is None else float(a["y_max"])
        pts = np.ascontiguousarray(xs[ok])
        mu, sd = eng.predict(pts, slot=0, y_mean=float(a["y_means"][0]), y_std=float(a["y_stds"][0]))
        with np.errstate(all="ignore"):
            f = -1 * O.base_acq(a["acq"], mu, sd, a["param"], ym)
            if a["lb"] is not None:
                for j in range(len(a["lb"])):
                    cm, cs = eng.predict(pts, slot=j + 1, y_mean=float(a["y_means"][j + 1]), y_std=float(a["y_stds"][j + 1]))
                    pl = O._cdf_loc_scale(a["lb"][j], cm, cs) if a["lb"][j] != -np.inf else 0.0
                    pu = O._cdf_loc_scale(a["ub"][j], cm, cs) if a["ub"][j] != np.inf else 1.0
                    f = f * (pu - pl)
        fin = np.isfinite(f)
        if fin.any():
            bars.close("polish_value_at_x", fs[ok][fin], f[fin], 1e-6, scale=max(float(np.max(np.abs(f[fin]))), 1e-12))
