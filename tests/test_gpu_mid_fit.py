"""The strip path of mid-size fits / LML evaluations (csrc/mid_fit.hip: 64 < NP <= 768 in the product, up to 1024 by switch)
against the multi-launch sequence it replaces and against the oracle.

What must be BITWISE the multi-launch path: K (kmat_q_kernel is kmat_kernel's arithmetic element for element) and L with its pivot
order (the factorisation is the same launches).  What may differ in rounding only: W = L^-1 (forward substitution by column strips
instead of recursive doubling), alpha and everything computed from them — checked against the oracle at the bars of test_fit_parity /
test_lml_parity and against the multi-launch results at a tighter one.  What must be bitwise WITHIN the path: a lane of
gpbo_lml_batch = gpbo_lml, overlapped fits = sequential fits, the product library = the debug build.

Replaces in the reference: GaussianProcessRegressor.fit at fixed theta and log_marginal_likelihood (sklearn _gpr.py:296-364,
575-652; the triangular solves of _gpr.py:454-456) at the sizes BASELINE config 2 and the later steps of a maximize() loop have.
"""
import numpy as np
import pytest

from tests import helpers as H
from tests.conftest import elementwise_err, rel_err
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


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
    K, L, Wm, al = eng.get_K(N, slot), eng.get_L(N, slot), eng.get_Linv(N, slot), eng.get_alpha(N, slot)
    mu_big, sd_big = eng.predict(Xc, slot=slot, y_mean=0.3, y_std=1.7)            # MFMA path: reads the packed W
    mu_small, sd_small = eng.predict(Xc[:7], slot=slot, y_mean=0.3, y_std=1.7)    # GEMV path: reads W
    return dict(K=K, L=L, W=Wm, alpha=al, mu_big=mu_big, sd_big=sd_big, mu_small=mu_small, sd_small=sd_small)


CASES = [  # N, d, kernel, per-dimension length scales, strip-path limit
    (129, 4, O.MATERN25, False, 512), (192, 8, O.RBF, True, 512), (200, 2, O.RBF, False, 512), (256, 5, O.MATERN25, False, 512),
    (300, 8, O.MATERN25, True, 512), (384, 33, O.RBF, True, 512), (449, 16, O.MATERN25, False, 512), (512, 8, O.MATERN25, False, 512),
    # beyond the product's rule (the strip's LDS image allows NP <= 1024)
    (640, 8, O.MATERN25, True, 1024), (1000, 6, O.RBF, False, 1024), (1024, 16, O.MATERN25, False, 1024),
    # below it (NP <= 128 is the one-workgroup kernel's in the product; the strip path itself has no lower limit)
    (40, 3, O.RBF, False, 512), (128, 8, O.MATERN25, True, 512),
]


@pytest.mark.parametrize("N,d,kernel,per_dim,limit", CASES)
def test_strip_fit_against_the_multi_launch_fit_and_the_oracle(debug_engine, N, d, kernel, per_dim, limit):
    X, yn, ls, Xc = problem(N, d, 3000 + N, per_dim)
    with H.fit_paths(fused=0, mid=0):
        ref = fit_state(debug_engine, X, yn, kernel, ls, Xc)
    with H.fit_paths(fused=0, mid=limit):
        got = fit_state(debug_engine, X, yn, kernel, ls, Xc)
        again = fit_state(debug_engine, X, yn, kernel, ls, Xc)
    assert np.array_equal(ref["K"], got["K"]) and np.array_equal(ref["L"], got["L"])
    for k in got:
        assert np.array_equal(got[k], again[k]), k                   # deterministic
    assert np.all(np.triu(got["W"], 1) == 0.0)
    gp = O.fit_fixed_theta(kernel, X, yn, ls, 1e-6, normalize_y=False)
    Winv = np.linalg.inv(gp.L)
    assert rel_err(got["W"], Winv) < 1e-8 and rel_err(got["alpha"], gp.alpha) < 1e-8        # test_fit_parity's bars
    # the two device algorithms: closer to each other than either has to be to LAPACK's
    assert rel_err(got["W"], ref["W"]) < 1e-10 and rel_err(got["alpha"], ref["alpha"]) < 1e-9
    # W L = I to rounding (the residual of the inverse itself, independent of any reference)
    assert np.max(np.abs(got["W"] @ got["L"] - np.eye(N))) < 1e-11 * max(1.0, float(np.max(np.abs(got["W"]))))
    # the posterior, per candidate: north_star's 1e-5 against the oracle; the two device algorithms a decade closer (what separates
    # them is kappa(K) x rounding: 2e-8 at N = 200, d = 2, RBF)
    gp.y_mean, gp.y_std = 0.3, 1.7
    for big, Xq in (("big", Xc), ("small", Xc[:7])):
        mu_o, sd_o = O.predict(gp, Xq)
        e_sd, e_mu = elementwise_err(got["sd_" + big], sd_o, got["mu_" + big], mu_o, 1.7)
        assert e_sd < 1e-5 and e_mu < 1e-5, (big, e_sd, e_mu)
        e_sd, e_mu = elementwise_err(got["sd_" + big], ref["sd_" + big], got["mu_" + big], ref["mu_" + big], 1.7)
        assert e_sd < 1e-6 and e_mu < 1e-6, (big, e_sd, e_mu)


@pytest.mark.parametrize("N,d,kernel,per_dim,limit", CASES)
def test_strip_lml_against_the_multi_launch_lml_and_the_oracle(debug_engine, N, d, kernel, per_dim, limit):
    X, yn, ls, _ = problem(N, d, 4000 + N, per_dim)
    thetas = np.stack([ls, ls * 1.7, ls * 0.4])
    with H.fit_paths(fused=0, mid=0):
        ref1 = debug_engine.lml(X, yn, kernel, ls, 1e-6, eval_gradient=True)
    with H.fit_paths(fused=0, mid=limit):
        got1 = debug_engine.lml(X, yn, kernel, ls, 1e-6, eval_gradient=True)
        got0 = debug_engine.lml(X, yn, kernel, ls, 1e-6, eval_gradient=False)
        gotb = debug_engine.lml_batch(X, yn, kernel, thetas, 1e-6)
        gotb2 = debug_engine.lml_batch(X, yn, kernel, thetas[::-1].copy(), 1e-6, reuse_inputs=True)
        gotb3 = debug_engine.lml_batch(X, yn, kernel, thetas, 1e-6, reuse_inputs=True)      # (third call of a shape: graph replay)
    assert got0 == got1[0]
    for (av, ag), (bv, bg) in zip(gotb, gotb2[::-1]):
        assert av == bv and np.array_equal(ag, bg)                   # a lane's result does not depend on its position
    for (av, ag), (bv, bg) in zip(gotb, gotb3):
        assert av == bv and np.array_equal(ag, bg)
    assert gotb[0][0] == got1[0] and np.array_equal(gotb[0][1], got1[1])      # a lane = the single evaluation
    v, g = O.log_marginal_likelihood(kernel, X, yn, ls, 1e-6)
    # the bars of test_lml_parity; the ill-conditioned case (RBF, d = 2, N = 200: kappa(K) ~ 2e8, y^T K^-1 y ~ 2e6) gets the two
    # decades its condition number takes from ANY algorithm (the multi-launch path is 1.5e-10 from LAPACK there as well)
    slack = 100.0 if (N, d) == (200, 2) else 1.0
    assert abs(got1[0] - v) <= slack * 1e-10 * max(1.0, abs(v))
    assert np.max(np.abs(got1[1] - g)) <= slack * 1e-7 * max(1.0, float(np.max(np.abs(g))))
    assert abs(got1[0] - ref1[0]) <= slack * 1e-11 * max(1.0, abs(v))
    assert np.max(np.abs(got1[1] - ref1[1])) <= slack * 1e-8 * max(1.0, float(np.max(np.abs(g))))


def test_strip_overlapped_fits_append_and_not_pd(debug_engine):
    eng = debug_engine
    X, yn, ls, Xc = problem(300, 6, 7, True)
    X2, yn2, ls2, _ = problem(450, 6, 8, False)
    with H.fit_paths(fused=128, mid=512):
        seq_a = fit_state(eng, X, yn, O.MATERN25, ls, Xc, slot=0)
        seq_b = fit_state(eng, X2, yn2, O.RBF, ls2, Xc, slot=1)
        with eng.overlapped_fits():                      # gpbo_fit_begin on two slot streams: own staging windows, own pivot words
            eng.fit(X, yn, O.MATERN25, ls, 1e-6, slot=0)
            eng.fit(X2, yn2, O.RBF, ls2, 1e-6, slot=1)
        for slot, ref, n in ((0, seq_a, 300), (1, seq_b, 450)):
            assert np.array_equal(eng.get_L(n, slot), ref["L"])
            assert np.array_equal(eng.get_Linv(n, slot), ref["W"])
            assert np.array_equal(eng.get_alpha(n, slot), ref["alpha"])
            mu, sd = eng.predict(Xc, slot=slot, y_mean=0.3, y_std=1.7)
            assert np.array_equal(mu, ref["mu_big"]) and np.array_equal(sd, ref["sd_big"])
    # gpbo_fit_append: 20 new rows at once re-run the factorisation from the resident inputs; one row grows it by a rank-one step
    rng = np.random.RandomState(3)
    Xn = rng.uniform(-2, 3, size=(21, 6))
    y_all = np.concatenate([yn, rng.standard_normal(21) * 0.3])
    out = {}
    for mid in (0, 512):
        with H.fit_paths(fused=128, mid=mid):
            eng.fit(X, yn, O.MATERN25, ls, 1e-6, slot=0)
            eng.fit_append(Xn[:20], y_all[:320], slot=0)
            eng.fit_append(Xn[20:], y_all, slot=0)
            out[mid] = (eng.get_L(321, 0), eng.get_Linv(321, 0), eng.get_alpha(321, 0), eng.predict(Xc, slot=0))
    gp = O.fit_fixed_theta(O.MATERN25, np.vstack([X, Xn]), y_all, ls, 1e-6, normalize_y=False)
    assert rel_err(out[512][0], np.tril(gp.L)) < 1e-10
    assert rel_err(out[512][1], np.linalg.inv(gp.L)) < 1e-8 and rel_err(out[512][2], gp.alpha) < 1e-8
    assert rel_err(out[512][3][0], out[0][3][0]) < 1e-9 and rel_err(out[512][3][1], out[0][3][1]) < 1e-9
    # not positive definite: the same LAPACK-style order from both paths, -inf / zero gradient from the LML entry points
    Xd = np.vstack([X[:150], X[:150]])
    orders = []
    for mid in (0, 512):
        with H.fit_paths(fused=128, mid=mid):
            with pytest.raises(np.linalg.LinAlgError) as ei:
                eng.fit(Xd, np.zeros(300), O.RBF, [1.0], 0.0, slot=0)
            orders.append(str(ei.value))
            v = eng.lml(Xd, np.zeros(300), O.RBF, [1.0], 0.0, eval_gradient=True)
            assert v[0] == -np.inf and np.all(v[1] == 0)
            vb = eng.lml_batch(Xd, np.zeros(300), O.RBF, np.array([[1.0], [2.0]]), 0.0)
            assert all(x[0] == -np.inf for x in vb)
    assert orders[0] == orders[1]


def test_product_library_uses_the_strip_path_and_agrees_with_the_debug_build(engine, debug_engine):
    """The product has no switch: its 128 < NP <= 512 fits ARE the strip path.  Same bits as the debug build's."""
    X, yn, ls, Xc = problem(333, 5, 11, True)
    with H.fit_paths(fused=None, mid=None):
        ref = fit_state(debug_engine, X, yn, O.MATERN25, ls, Xc)
        ref_l = debug_engine.lml(X, yn, O.MATERN25, ls, 1e-6)
    got = fit_state(engine, X, yn, O.MATERN25, ls, Xc)
    got_l = engine.lml(X, yn, O.MATERN25, ls, 1e-6)
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k
    assert ref_l[0] == got_l[0] and np.array_equal(ref_l[1], got_l[1])
    t = engine.last_timings()
    assert t["kmat"] < 0 and t["cholesky"] < 0 and t["fit"] > 0      # no per-phase events on this path
    with H.fit_paths(fused=0, mid=0):
        old = fit_state(debug_engine, X, yn, O.MATERN25, ls, Xc)
    assert np.array_equal(old["L"], got["L"]) and not np.array_equal(old["W"], got["W"])      # (the switch did switch)
