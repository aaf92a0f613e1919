"""The one-launch fit / LML evaluation of small problems (csrc/fused_small.hip) against the multi-launch sequence it replaces:
BITWISE — the fused kernel runs the same device bodies in the same order (fit_bodies.h, chol_bodies.h, gemm_tile.h, lml_bodies.h).

The debug build reads GPBO_FUSED_MAX_NP per call: 0 = the multi-launch path, 64 = the product's rule (above it the strip path of
csrc/mid_fit.hip is faster: profiles/r05_small_fit_timing.json), 128 = one diagonal workgroup with its two 64-blocks, 512 = the fused
kernel's general schedule (several diagonal blocks, panel solves, trailing tiles, ragged trtri levels) — exercised here although the
product only sends NP <= 64 there.  What it replaces in the reference: GaussianProcessRegressor.fit at fixed theta and
log_marginal_likelihood (sklearn _gpr.py:296-364, 575-652) at the sizes a maximize() loop lives at.
"""
import numpy as np
import pytest

from tests import helpers as H
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def fused_max_np(v):
    """The one-workgroup kernel up to NP = v, the multi-launch sequence above (the strip path of csrc/mid_fit.hip stays off: this
    file compares against the launches the fused kernel replaces body for body)."""
    return H.fit_paths(fused=v, mid=0)


def problem(N, d, seed, per_dim=False):
    rng = np.random.RandomState(seed)
    X = rng.uniform(-2.0, 3.0, size=(N, d))
    y = np.sin(X.sum(axis=1)) + 0.1 * rng.standard_normal(N)
    yn = (y - y.mean()) / y.std()
    ls = rng.uniform(0.6, 1.7, size=d) if per_dim else np.array([0.9 + 0.05 * d])
    Xc = rng.uniform(-2.0, 3.0, size=(5000, d))
    return X, yn, ls, Xc


def fit_state(eng, X, yn, kernel, ls, Xc, slot=0):
    N = X.shape[0]
    eng.fit(X, yn, kernel, ls, 1e-6, slot=slot)
    L, Wm, al = eng.get_L(N, slot), eng.get_Linv(N, slot), eng.get_alpha(N, slot)
    mu_big, sd_big = eng.predict(Xc, slot=slot, y_mean=0.3, y_std=1.7)            # MFMA path: reads the packed W
    mu_small, sd_small = eng.predict(Xc[:7], slot=slot, y_mean=0.3, y_std=1.7)    # GEMV path: reads W
    return dict(L=L, W=Wm, alpha=al, mu_big=mu_big, sd_big=sd_big, mu_small=mu_small, sd_small=sd_small)


CASES = [  # N, d, kernel, per-dimension length scales, fused limit
    (5, 2, O.RBF, False, 128), (25, 2, O.RBF, False, 128), (64, 8, O.MATERN25, True, 128), (65, 3, O.MATERN25, False, 128),
    (100, 17, O.RBF, True, 128), (128, 8, O.MATERN25, False, 128),
    # the general schedule (not what the product dispatches)
    (129, 4, O.MATERN25, False, 512), (192, 8, O.RBF, True, 512), (256, 5, O.MATERN25, False, 512), (300, 8, O.MATERN25, True, 512),
    (448, 16, O.MATERN25, False, 512), (512, 8, O.MATERN25, False, 512),
]


@pytest.mark.parametrize("N,d,kernel,per_dim,limit", CASES)
def test_fused_fit_is_bitwise_the_multi_launch_fit(debug_engine, N, d, kernel, per_dim, limit):
    X, yn, ls, Xc = problem(N, d, 1000 + N, per_dim)
    with fused_max_np(0):
        ref = fit_state(debug_engine, X, yn, kernel, ls, Xc)
    with fused_max_np(limit):
        got = fit_state(debug_engine, X, yn, kernel, ls, Xc)
    for k in ref:
        assert np.array_equal(ref[k], got[k]), (k, float(np.max(np.abs(ref[k] - got[k]))))
    # and against the oracle (the same bar as test_fit_parity)
    Lo = O.fit_fixed_theta(kernel, X, yn, ls, 1e-6, normalize_y=False).L
    assert np.max(np.abs(got["L"] - np.tril(Lo))) / np.max(np.abs(Lo)) < 1e-10


@pytest.mark.parametrize("N,d,kernel,per_dim,limit", CASES)
def test_fused_lml_is_bitwise_the_multi_launch_lml(debug_engine, N, d, kernel, per_dim, limit):
    X, yn, ls, _ = problem(N, d, 2000 + N, per_dim)
    thetas = np.stack([ls, ls * 1.7, ls * 0.4])
    with fused_max_np(0):
        ref1 = debug_engine.lml(X, yn, kernel, ls, 1e-6, eval_gradient=True)
        ref0 = debug_engine.lml(X, yn, kernel, ls, 1e-6, eval_gradient=False)
        refb = debug_engine.lml_batch(X, yn, kernel, thetas, 1e-6)
    with fused_max_np(limit):
        got1 = debug_engine.lml(X, yn, kernel, ls, 1e-6, eval_gradient=True)
        got0 = debug_engine.lml(X, yn, kernel, ls, 1e-6, eval_gradient=False)
        gotb = debug_engine.lml_batch(X, yn, kernel, thetas, 1e-6)
        gotb2 = debug_engine.lml_batch(X, yn, kernel, thetas[::-1].copy(), 1e-6, reuse_inputs=True)
    assert ref1[0] == got1[0] and np.array_equal(ref1[1], got1[1])
    assert ref0 == got0 == ref1[0]
    for (rv, rg), (gv, gg) in zip(refb, gotb):
        assert rv == gv and np.array_equal(rg, gg)
    for (rv, rg), (gv, gg) in zip(refb[::-1], gotb2):
        assert rv == gv and np.array_equal(rg, gg)
    assert gotb[0][0] == got1[0] and np.array_equal(gotb[0][1], got1[1])      # a lane = the single evaluation
    v, g = O.log_marginal_likelihood(kernel, X, yn, ls, 1e-6)
    assert abs(got1[0] - v) <= 1e-10 * max(1.0, abs(v))
    assert np.max(np.abs(got1[1] - g)) <= 1e-7 * max(1.0, float(np.max(np.abs(g))))


def test_fused_overlapped_fits_append_and_not_pd(debug_engine):
    eng = debug_engine
    X, yn, ls, Xc = problem(90, 6, 7, True)
    X2, yn2, ls2, _ = problem(120, 6, 8, False)
    with fused_max_np(0):
        ref_a = fit_state(eng, X, yn, O.MATERN25, ls, Xc, slot=0)
        ref_b = fit_state(eng, X2, yn2, O.RBF, ls2, Xc, slot=1)
    with fused_max_np(128):
        with eng.overlapped_fits():                      # gpbo_fit_begin on two slot streams: two one-workgroup launches side by side
            eng.fit(X, yn, O.MATERN25, ls, 1e-6, slot=0)
            eng.fit(X2, yn2, O.RBF, ls2, 1e-6, slot=1)
        for slot, ref, n in ((0, ref_a, 90), (1, ref_b, 120)):
            assert np.array_equal(eng.get_L(n, slot), ref["L"])
            assert np.array_equal(eng.get_Linv(n, slot), ref["W"])
            assert np.array_equal(eng.get_alpha(n, slot), ref["alpha"])
            mu, sd = eng.predict(Xc, slot=slot, y_mean=0.3, y_std=1.7)
            assert np.array_equal(mu, ref["mu_big"]) and np.array_equal(sd, ref["sd_big"])
    # gpbo_fit_append: 20 new rows at once re-run the factorisation from the resident inputs (the fused kernel's src = 1 form)
    rng = np.random.RandomState(3)
    Xn = rng.uniform(-2, 3, size=(20, 6))
    y_all = np.concatenate([yn, rng.standard_normal(20) * 0.3])
    out = {}
    for limit in (0, 128):
        with fused_max_np(limit):
            eng.fit(X, yn, O.MATERN25, ls, 1e-6, slot=0)
            eng.fit_append(Xn, y_all, slot=0)
            out[limit] = (eng.get_L(110, 0), eng.get_Linv(110, 0), eng.get_alpha(110, 0), eng.predict(Xc, slot=0))
    for a, b in zip(out[0][:3], out[128][:3]):
        assert np.array_equal(a, b)
    assert np.array_equal(out[0][3][0], out[128][3][0]) and np.array_equal(out[0][3][1], out[128][3][1])
    Lo = O.fit_fixed_theta(O.MATERN25, np.vstack([X, Xn]), y_all, ls, 1e-6, normalize_y=False).L
    assert np.max(np.abs(out[128][0] - np.tril(Lo))) / np.max(np.abs(Lo)) < 1e-10
    # not positive definite: the same LAPACK-style order from both paths
    Xd = np.vstack([X[:40], X[:40]])
    orders = []
    for limit in (0, 128):
        with fused_max_np(limit):
            with pytest.raises(np.linalg.LinAlgError) as ei:
                eng.fit(Xd, np.zeros(80), O.RBF, [1.0], 0.0, slot=0)
            orders.append(str(ei.value))
            v = eng.lml(Xd, np.zeros(80), O.RBF, [1.0], 0.0, eval_gradient=True)
            assert v[0] == -np.inf and np.all(v[1] == 0)
    assert orders[0] == orders[1]


def test_product_library_uses_the_fused_kernel_and_agrees_with_the_debug_build(engine, debug_engine):
    """The product has no switch: its NP <= 64 fits ARE the fused kernel.  Same bits as the debug build's multi-launch path."""
    X, yn, ls, Xc = problem(57, 5, 11, True)
    with fused_max_np(0):
        ref = fit_state(debug_engine, X, yn, O.MATERN25, ls, Xc)
        ref_l = debug_engine.lml(X, yn, O.MATERN25, ls, 1e-6)
    got = fit_state(engine, X, yn, O.MATERN25, ls, Xc)
    got_l = engine.lml(X, yn, O.MATERN25, ls, 1e-6)
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k
    assert ref_l[0] == got_l[0] and np.array_equal(ref_l[1], got_l[1])
    t = engine.last_timings()
    assert t["kmat"] < 0 and t["cholesky"] < 0      # no per-phase events: the fit was one kernel
