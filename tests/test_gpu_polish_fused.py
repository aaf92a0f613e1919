"""GPU (-m gpu): the local searches of gpbo_polish_seeds as ONE launch (csrc/polish_fused.hip: polish_rows_kernel — a workgroup
per run, thread = training point, the evaluations and the optimiser inside it; NP <= 512) against
the lockstep path it replaces there (csrc/polish.hip: six launches and a stream synchronisation per round, the optimiser on the
host), which is its CHECKER, not its twin.

What it replaces: AcquisitionFunction._smart_minimize (bayes_opt/acquisition.py:322-420).  SURVEY.md section 8 f2: "parity is
statistical (same or better acquisition value), not bit-wise" — the kernel sums in other orders than the six kernels, so
  * one evaluation — f, mu, sd and the three gradients — agrees with gpbo_predict_grad's kernels to 2e-11 of the values' scale
    (measured 3e-12: sums that cancel, |alpha| ~ cond(K) |y|);
  * a whole search ends at the lockstep path's value or a better one: run by run to 1e-8 in at least 8 of 10 runs (a rounding may tip
    one line-search test), for the best run always; no run unconverged that converged there; the value returned is the objective
    at the point returned;
  * EI / POI evaluations follow the host formulas over that posterior to rounding.
The switches (GPBO_POLISH_FUSED=0, GPBO_POLISH_FUSED_MAX_NP) and the single-evaluation entry exist in libgpbo_dbg.so only; the last
test pins the product library to the debug build's bits."""
import os

import numpy as np
import pytest

from bayesianoptimization_amd import _lib
from oracle import gp_oracle as O


pytestmark = pytest.mark.gpu


@pytest.fixture
def lockstep_only():
    old = os.environ.get("GPBO_POLISH_FUSED")

    def switch(on):
        if on:
            os.environ["GPBO_POLISH_FUSED"] = "0"
        else:
            os.environ.pop("GPBO_POLISH_FUSED", None)

    yield switch
    if old is None:
        os.environ.pop("GPBO_POLISH_FUSED", None)
    else:
        os.environ["GPBO_POLISH_FUSED"] = old


def _problem(N, d, seed, kernel=O.MATERN25, ls=None):
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(N, d))
    y = np.sin(3 * X.sum(1)) + 0.05 * rng.randn(N)
    if ls is None:
        ls = 0.25 * np.sqrt(d)
    return X, y, ls


def _fit(eng, X, y, kernel, ls):
    yn, ym, ys = O.normalize_targets(y)
    eng.fit(X, yn, kernel, ls, 1e-6, slot=0)
    return ym, ys


def _polish_eval(eng, acq, param, y_max, ym, ys, pts, repeat=1):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    n, d = pts.shape
    out = np.empty((n, 4 + 3 * d))
    rc = eng._lib.gpbo_debug_polish_eval(eng._h, int(acq), float(param), float(y_max), float(ym), float(ys), _lib.dptr(pts), n, d,
                                         int(repeat), _lib.dptr(out))
    eng._check(rc)
    return {"f": out[:, 0], "mu": out[:, 1], "sd": out[:, 2], "g": out[:, 4:4 + d], "dmu": out[:, 4 + d:4 + 2 * d],
            "dsd": out[:, 4 + 2 * d:4 + 3 * d]}


SHAPES = [(25, 2), (64, 4), (65, 3), (100, 5), (120, 40), (128, 16), (130, 8), (200, 2), (250, 32), (256, 64), (300, 6), (384, 3), (448, 16), (512, 8)]


@pytest.fixture
def any_size():
    """The one launch serves NP <= 512, the kernel's own limit (the product's since the end of round 6; 384 before): the tests pin the
    limit there whatever the default (GPBO_POLISH_FUSED_MAX_NP, read per call by the debug build only)."""
    old = os.environ.get("GPBO_POLISH_FUSED_MAX_NP")
    os.environ["GPBO_POLISH_FUSED_MAX_NP"] = "512"
    yield
    if old is None:
        os.environ.pop("GPBO_POLISH_FUSED_MAX_NP", None)
    else:
        os.environ["GPBO_POLISH_FUSED_MAX_NP"] = old


@pytest.mark.parametrize("kernel", [O.MATERN25, O.RBF])
@pytest.mark.parametrize("N,d", SHAPES)
def test_one_evaluation_is_the_six_kernels_to_rounding(debug_engine, any_size, kernel, N, d):
    eng = debug_engine
    X, y, ls = _problem(N, d, 11 + N + d, kernel)
    if d % 2:           # per-dimension length scales on the odd ones
        ls = ls * np.linspace(0.7, 1.4, d)
    ym, ys = _fit(eng, X, y, kernel, ls)
    rng = np.random.RandomState(5)
    pts = np.vstack([rng.uniform(size=(7, d)), X[:2] + 1e-6, np.full((1, d), 0.5)])
    kappa = 2.576
    got = _polish_eval(eng, O.UCB, kappa, 0.0, ym, ys, pts)
    mu, sd, dmu, dsd = eng.predict_grad(pts, slot=0, y_mean=ym, y_std=ys)
    # Two correct summations of the same terms differ by ~ n eps sum |terms|, and the terms cancel: mu = y_std sum_k k*_k alpha_k with
    # |alpha| ~ cond(K) |y| (d = 2, N = 200: sum |k* alpha| ~ 1e4 |mu|), v = W k* likewise.  So the bar is 512 eps times the sum of
    # the terms' magnitudes (from the device's own alpha and W), not a multiple of the result.
    eps = np.finfo(np.float64).eps
    Kst = O.kernel_matrix(kernel, pts, X, np.atleast_1d(ls))
    alpha, Wm = eng.get_alpha(N), eng.get_Linv(N)
    term_mu = ys * (np.abs(Kst) @ np.abs(alpha))
    assert np.all(np.abs(got["mu"] - mu) <= 512 * eps * term_mu + 1e-15)
    V, absV = Kst @ Wm.T, np.abs(Kst) @ np.abs(Wm).T
    term_var = ys * ys * 2.0 * np.sum(np.abs(V) * absV, axis=1)
    assert np.all(np.abs(got["sd"] ** 2 - sd ** 2) <= 512 * eps * term_var + 1e-15 * ys * ys)
    amp_mu = max(1.0, float(np.max(term_mu)) / max(float(np.abs(mu).max()), ys))
    amp_w = max(1.0, float(np.max(term_var)) / (ys * ys))
    assert np.max(np.abs(got["dmu"] - dmu)) <= 2e-13 * amp_mu * float(np.abs(dmu).max())
    far = sd > 1e-3 * ys                                   # d sd = -(...) / sd: amplified without bound as sd -> 0
    if far.any():
        assert np.max(np.abs(got["dsd"][far] - dsd[far])) <= 1e-11 * amp_w * amp_w * float(np.abs(dsd[far]).max())
    assert np.max(np.abs(got["f"] + (got["mu"] + kappa * got["sd"]))) <= 1e-15 * max(float(np.abs(mu).max()), ys)
    assert np.allclose(got["g"], -(got["dmu"] + kappa * got["dsd"]), rtol=1e-14, atol=0)
    # and they are the oracle's values (the six kernels' own parity: tests/test_gpu_parity.py)
    gp = O.fit_fixed_theta(kernel, X, y, ls, 1e-6)
    mu_o, sd_o = O.predict(gp, pts)
    assert np.allclose(got["mu"], mu_o, rtol=0, atol=1e-7 * max(1.0, float(np.abs(mu_o).max())))
    assert np.allclose(got["sd"], sd_o, rtol=0, atol=1e-6 * float(sd_o.max()))


def test_ei_and_poi_evaluations_agree_with_the_host_formula_to_rounding(debug_engine, any_size):
    from scipy.stats import norm

    eng = debug_engine
    X, y, ls = _problem(300, 4, 3)
    ym, ys = _fit(eng, X, y, O.MATERN25, ls)
    pts = np.random.RandomState(2).uniform(size=(12, 4))
    mu, sd, dmu, dsd = eng.predict_grad(pts, slot=0, y_mean=ym, y_std=ys)
    y_max, xi = float(np.max(y)), 0.01
    a = mu - y_max - xi
    z = a / sd
    for acq in (O.EI, O.POI):
        got = _polish_eval(eng, acq, xi, y_max, ym, ys, pts)
        if acq == O.EI:
            f = -(a * norm.cdf(z) + sd * norm.pdf(z))
            g = -(norm.cdf(z)[:, None] * dmu + norm.pdf(z)[:, None] * dsd)
        else:
            f = -norm.cdf(z)
            g = -((norm.pdf(z) / sd)[:, None] * dmu + (-norm.pdf(z) * z / sd)[:, None] * dsd)
        # (a Phi(z) + sd phi(z) cancels in the lower tail: relative to the terms, not to their difference)
        scale_f = np.abs(a) * norm.cdf(z) + sd * norm.pdf(z) if acq == O.EI else norm.cdf(z)
        # (mu, sd themselves agree to ~3e-12 of their scale; in the tails exp(-z^2 / 2) turns that into z^2 times as much)
        assert np.all(np.abs(got["f"] - f) <= 1e-11 * (1.0 + z * z) * scale_f + 1e-300)
        scale_g = norm.cdf(z)[:, None] * np.abs(dmu) + norm.pdf(z)[:, None] * np.abs(dsd) if acq == O.EI else \
            (norm.pdf(z) / sd)[:, None] * (np.abs(dmu) + np.abs(z)[:, None] * np.abs(dsd))
        assert np.all(np.abs(got["g"] - g) <= 1e-10 * (1.0 + z * z)[:, None] * scale_g + 1e-300)


def _both(eng, switch, acq, param, y_max, ym, ys, seeds, box, max_iter=0):
    switch(True)
    ref = eng.polish_seeds(acq, param, y_max, None, None, [ym], [ys], seeds, box, max_iter=max_iter)
    ref_counts = {k: np.array(v) for k, v in eng.last_polish.items()}
    switch(False)
    got = eng.polish_seeds(acq, param, y_max, None, None, [ym], [ys], seeds, box, max_iter=max_iter)
    got_counts = {k: np.array(v) for k, v in eng.last_polish.items()}
    return ref, ref_counts, got, got_counts


def _same_or_better(ref, got, gp, acq, param, y_max):
    """The lockstep path as the checker (see the module's docstring)."""
    scale = max(abs(float(ref[1].min())), 1e-12)
    close = np.abs(got[1] - ref[1]) <= 1e-8 * scale
    assert close.sum() >= min(8, len(ref[1])), (got[1], ref[1])
    assert float(got[1].min()) <= float(ref[1].min()) + 1e-8 * scale
    assert np.all(got[2][ref[2] < 2] < 2)
    f_at = O.neg_acquisition(gp, got[0], acq, param, y_max, None)
    assert np.all(np.abs(got[1] - f_at) <= 1e-6 * np.abs(f_at) + 1e-9)


@pytest.mark.parametrize("kernel", [O.MATERN25, O.RBF])
@pytest.mark.parametrize("N,d", [(25, 2), (64, 4), (100, 5), (128, 12), (200, 2), (300, 6), (384, 16), (512, 8)])
def test_a_whole_ucb_search_against_the_lockstep_path(debug_engine, lockstep_only, any_size, kernel, N, d):
    eng = debug_engine
    X, y, ls = _problem(N, d, 100 + N, kernel)
    ym, ys = _fit(eng, X, y, kernel, ls)
    gp = O.fit_fixed_theta(kernel, X, y, ls, 1e-6)
    rng = np.random.RandomState(9)
    cand = rng.uniform(size=(3000, d))
    vals = O.neg_acquisition(gp, cand, O.UCB, 2.576, 0.0, None)
    seeds = cand[np.argsort(vals)[:10]].copy()
    seeds[0] = np.clip(seeds[0] + 0.7, -0.5, 1.5)      # a seed outside the box (clipped into it first) ...
    seeds[1, 0] = 0.0                                  # ... and one on a bound
    box = np.array([[0.0, 1.0]] * d)
    ref, rc, got, gc = _both(eng, lockstep_only, O.UCB, 2.576, 0.0, ym, ys, seeds, box)
    _same_or_better(ref, got, gp, O.UCB, 2.576, 0.0)
    assert got[3] == int(np.max(gc["nfev"]))           # "rounds" of the one launch = its longest run's evaluations
    assert np.all(got[0] >= 0.0) and np.all(got[0] <= 1.0)
    assert np.all(got[1] <= O.neg_acquisition(gp, np.clip(seeds, 0.0, 1.0), O.UCB, 2.576, 0.0, None) + 1e-9)


def test_iteration_limit_and_a_single_seed(debug_engine, lockstep_only):
    eng = debug_engine
    X, y, ls = _problem(150, 3, 4)
    ym, ys = _fit(eng, X, y, O.MATERN25, ls)
    gp = O.fit_fixed_theta(O.MATERN25, X, y, ls, 1e-6)
    box = np.array([[0.0, 1.0]] * 3)
    seeds = np.random.RandomState(1).uniform(size=(1, 3))
    ref, rc, got, gc = _both(eng, lockstep_only, O.UCB, 2.576, 0.0, ym, ys, seeds, box, max_iter=2)
    # two accepted steps of one optimiser over one objective: the same iterates to rounding, the same counts
    assert gc["nit"][0] <= 2 and np.array_equal(gc["nit"], rc["nit"]) and np.array_equal(gc["nfev"], rc["nfev"])
    assert np.array_equal(got[2], ref[2]) and np.allclose(got[0], ref[0], rtol=0, atol=1e-9) and np.allclose(got[1], ref[1], rtol=1e-9)
    seeds = np.random.RandomState(2).uniform(size=(64, 3))          # the ABI's maximum
    ref, rc, got, gc = _both(eng, lockstep_only, O.UCB, 1.0, 0.0, ym, ys, seeds, box)
    scale = max(abs(float(ref[1].min())), 1e-12)
    assert (np.abs(got[1] - ref[1]) <= 1e-8 * scale).sum() >= 58 and float(got[1].min()) <= float(ref[1].min()) + 1e-8 * scale
    assert np.all(got[2][ref[2] < 2] < 2)


@pytest.mark.parametrize("acq", [O.EI, O.POI])
@pytest.mark.parametrize("N,d", [(60, 2), (300, 6), (512, 8)])
def test_ei_and_poi_searches_agree_with_the_lockstep_path_to_rounding(debug_engine, lockstep_only, any_size, acq, N, d):
    eng = debug_engine
    X, y, ls = _problem(N, d, 7 + N)
    ym, ys = _fit(eng, X, y, O.MATERN25, ls)
    gp = O.fit_fixed_theta(O.MATERN25, X, y, ls, 1e-6)
    y_max = float(np.max(y))
    cand = np.random.RandomState(9).uniform(size=(3000, d))
    vals = O.neg_acquisition(gp, cand, acq, 0.01, y_max, None)
    seeds = cand[np.argsort(vals)[:10]]
    box = np.array([[0.0, 1.0]] * d)
    ref, rc, got, gc = _both(eng, lockstep_only, acq, 0.01, y_max, ym, ys, seeds, box)
    # one rounding of erfc / exp may tip a line-search test: the runs agree in value almost always, the best one always
    scale = max(abs(float(ref[1].min())), 1e-12)
    close = np.abs(got[1] - ref[1]) <= 1e-8 * scale
    assert close.sum() >= 8, (got[1], ref[1])
    assert abs(float(got[1].min()) - float(ref[1].min())) <= 1e-8 * scale
    assert np.all(got[0] >= 0.0) and np.all(got[0] <= 1.0)
    f_at = O.neg_acquisition(gp, got[0], acq, 0.01, y_max, None)
    assert np.all(np.abs(got[1] - f_at) <= 1e-6 * np.abs(f_at) + 1e-9)


def test_above_the_size_limit_and_with_constraints_the_lockstep_path_serves(debug_engine, lockstep_only, any_size):
    eng = debug_engine
    box = np.array([[0.0, 1.0]] * 3)
    seeds = np.random.RandomState(1).uniform(size=(6, 3))
    X, y, ls = _problem(800, 3, 4)                                   # NP = 832 > 512
    ym, ys = _fit(eng, X, y, O.MATERN25, ls)
    ref, rc, got, gc = _both(eng, lockstep_only, O.UCB, 2.576, 0.0, ym, ys, seeds, box)
    for a, b in zip(got[:3], ref[:3]):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):                                  # ... and the single-evaluation entry says so
        _polish_eval(eng, O.UCB, 2.576, 0.0, ym, ys, seeds)
    X, y, ls = _problem(200, 3, 5)
    c = np.cos(2 * X.sum(1))
    yn, ym, ys = O.normalize_targets(y)
    cn, cm, cs = O.normalize_targets(c)
    eng.fit(X, yn, O.MATERN25, ls, 1e-6, slot=0)
    eng.fit(X, cn, O.MATERN25, 0.7, 1e-6, slot=1)
    out = []
    for on in (True, False):
        lockstep_only(on)
        out.append(eng.polish_seeds(O.EI, 0.01, float(np.max(y[c <= 0.5])), [-np.inf], [0.5], [ym, cm], [ys, cs], seeds, box))
    for a, b in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("N", [60, 100, 190, 380])        # W in LDS (one wave, two waves), W in memory (three, six waves)
def test_the_product_library_runs_the_same_search(engine, debug_engine, N):
    X, y, ls = _problem(N, 5, 21)
    box = np.array([[0.0, 1.0]] * 5)
    seeds = np.random.RandomState(3).uniform(size=(10, 5))
    res = []
    for eng in (engine, debug_engine):
        ym, ys = _fit(eng, X, y, O.MATERN25, ls)
        res.append(eng.polish_seeds(O.UCB, 2.576, 0.0, None, None, [ym], [ys], seeds, box))
    for a, b in zip(res[0][:3], res[1][:3]):
        assert np.array_equal(a, b)
    assert res[0][3] == res[1][3]
