"""GPU (-m gpu): the host-side mirror of the reference interface on the real engine — HipGPR (seam B2),
the fused acquisition classes (seam B1), HipConstraintModel and FloatSpace — reading like the
reference's own tests (tests/test_acquisition.py, tests/test_bayesian_optimization.py::test_predict*)."""
import numpy as np
import pytest
from sklearn.gaussian_process import GaussianProcessRegressor
from sklearn.gaussian_process.kernels import RBF, Matern

from bayesianoptimization_amd import fused_acquisition as A
from bayesianoptimization_amd import workloads as W
from bayesianoptimization_amd.constraint_model import HipConstraintModel
from bayesianoptimization_amd.gpr import HipGPR
from bayesianoptimization_amd.float_space import FloatSpace
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _space(w, with_constraint=False, engine=None):
    X, y, c = W.make_observations(w)
    cons = None
    if with_constraint:
        cons = HipConstraintModel(None, -np.inf, w.constraint_ub, engine=engine, random_state=1)
    sp = FloatSpace(w.pbounds(), constraint=cons)
    sp.register_bulk(X, y, c)
    return sp


def test_hipgpr_matches_sklearn_fixed_theta(engine):
    w = W.P1
    sp = _space(w)
    k = Matern(nu=2.5, length_scale=0.4)
    sk = GaussianProcessRegressor(kernel=k, alpha=1e-6, normalize_y=True, optimizer=None).fit(sp.params, sp.target)
    gp = HipGPR(kernel=k, alpha=1e-6, normalize_y=True, optimizer=None, engine=engine).fit(sp.params, sp.target)
    Xc = sp.random_sample(500, 3)
    m1, s1 = sk.predict(Xc, return_std=True)
    m2, s2 = gp.predict(Xc, return_std=True)
    assert rel_err(m2, m1) < 1e-9 and rel_err(s2, s1) < 1e-9
    assert rel_err(gp.predict(Xc), m1) < 1e-9
    assert gp.X_train_.shape == (w.N, w.d) and gp.n_features_in_ == w.d
    assert gp.L_.flags["F_CONTIGUOUS"] and rel_err(gp.L_, sk.L_) < 1e-10 and rel_err(gp.alpha_, sk.alpha_) < 1e-8
    assert rel_err(gp.predict(Xc[:20], return_cov=True)[1], sk.predict(Xc[:20], return_cov=True)[1]) < 1e-6
    with pytest.raises(RuntimeError):
        gp.predict(Xc, return_std=True, return_cov=True)
    with pytest.raises(ValueError):
        gp.predict(Xc[:, :2])


def test_hipgpr_theta_search_consumes_rng_like_sklearn(engine):
    """Default bayes_opt GP config (bayesian_optimization.py:124-130): 5 restarts drawn from the shared
    RandomState; HipGPR must land on the same theta and leave the stream in the same state."""
    w = W.F1
    sp = _space(w)
    r1, r2 = np.random.RandomState(3), np.random.RandomState(3)
    kw = dict(alpha=1e-6, normalize_y=True, n_restarts_optimizer=5)
    sk = GaussianProcessRegressor(kernel=Matern(nu=2.5), random_state=r1, **kw).fit(sp.params, sp.target)
    gp = HipGPR(kernel=Matern(nu=2.5), random_state=r2, engine=engine, lml_on_device=False, **kw).fit(sp.params, sp.target)
    assert np.array_equal(gp.kernel_.theta, sk.kernel_.theta)
    assert r1.uniform() == r2.uniform()
    g = load_golden("F1")
    assert np.allclose(gp.kernel_.length_scale, g["length_scale"], rtol=1e-12)
    assert gp.log_marginal_likelihood_value_ == pytest.approx(sk.log_marginal_likelihood_value_, rel=1e-12)
    Xc = sp.random_sample(300, 5)
    m1, s1 = sk.predict(Xc, return_std=True)
    m2, s2 = gp.predict(Xc, return_std=True)
    assert rel_err(m2, m1) < 1e-8 and rel_err(s2, s1) < 1e-8


def test_predict_at_training_points_and_prior(engine):
    """tests/test_bayesian_optimization.py:639-690 of the reference."""
    w = W.F1
    sp = _space(w)
    gp = HipGPR(kernel=Matern(nu=2.5, length_scale=0.6), alpha=1e-6, normalize_y=True, optimizer=None, engine=engine)
    prior_m, prior_s = gp.predict(sp.params[:5], return_std=True)      # unfitted prior
    assert np.allclose(prior_m, 0) and np.all(prior_s > 1e-2)
    gp.fit(sp.params, sp.target)
    m, s = gp.predict(sp.params, return_std=True)
    assert np.allclose(m, sp.target, atol=1e-3) and np.all(s < 0.02)


@pytest.mark.parametrize("name", ["C1", "F1", "P1", "P2", "C2"])
def test_fused_suggest_random_stage_equals_reference(engine, name):
    """acq.suggest(..., n_smart=0) through HipGPR + fused classes == the reference's own suggestion."""
    w = W.ALL[name]
    g = load_golden(name)
    sp = _space(w)
    k = RBF(length_scale=g["length_scale"]) if w.kernel == W.RBF else Matern(nu=2.5, length_scale=g["length_scale"])
    gp = HipGPR(kernel=k, alpha=w.noise, normalize_y=True, optimizer=None, engine=engine)
    fn = {W.UCB: A.UpperConfidenceBound(kappa=w.acq_param), W.EI: A.ExpectedImprovement(xi=w.acq_param),
          W.POI: A.ProbabilityOfImprovement(xi=w.acq_param)}[w.acq]
    x = fn.suggest(gp, sp, n_random=int(g["suggest_nsmart0_nrandom"]), n_smart=0, random_state=np.random.RandomState(7))
    assert np.array_equal(x, g["suggest_nsmart0_x"])
    assert fn.i == 1


def test_fused_constrained_ei_equals_reference(engine):
    w = W.C5S
    g = load_golden("C5S")
    sp = _space(w, with_constraint=True, engine=engine)
    sp.constraint._model[0].set_params(kernel=Matern(nu=2.5, length_scale=w.constraint_length_scale), optimizer=None)
    gp = HipGPR(kernel=Matern(nu=2.5, length_scale=w.length_scale), alpha=w.noise, normalize_y=True, optimizer=None, engine=engine)
    fn = A.ExpectedImprovement(xi=w.acq_param)
    x = fn.suggest(gp, sp, n_random=int(g["suggest_nsmart0_nrandom"]), n_smart=0, random_state=np.random.RandomState(7))
    assert np.array_equal(x, g["suggest_nsmart0_x"])
    assert fn.y_max == float(g["y_max"])
    S = len(g["p_c"])
    Xc = W.make_candidates(w.bounds_array(), w.M, 7)
    assert rel_err(sp.constraint.predict(Xc[:S]), g["p_c"]) < 1e-8
    assert rel_err(sp.constraint.approx(Xc[:S]), g["c_mu"]) < 1e-8
    with pytest.raises(A.ConstraintNotSupportedError):
        A.UpperConfidenceBound().suggest(gp, sp)


def test_smart_stage_improves_and_stays_in_bounds(engine):
    """Default suggest (n_smart=10): L-BFGS-B from the device-selected seeds; x_min must be a seed, the
    result must be in bounds and at least as good as the random-stage winner."""
    w = W.P2
    sp = _space(w)
    gp = HipGPR(kernel=RBF(length_scale=0.6), alpha=w.noise, normalize_y=True, optimizer=None, engine=engine)
    fn = A.ExpectedImprovement(xi=0.01)
    fn.y_max = sp._target_max()
    fn._fit_gp(gp, sp)
    acq = fn._get_acq(gp)
    from bayesianoptimization_amd.fused_acquisition import _fused_models
    fn._fused = _fused_models(gp, None)
    x_min, min_acq, seeds = fn._random_sample_minimize(acq, sp, np.random.RandomState(7), n_random=4096, n_x_seeds=10)
    assert any(np.array_equal(x_min, s) for s in seeds)
    assert acq(x_min)[0] == pytest.approx(min_acq, rel=1e-9)
    x = fn.suggest(gp, sp, n_random=4096, n_smart=10, fit_gp=False, random_state=np.random.RandomState(7))
    assert np.all(x >= sp.bounds[:, 0]) and np.all(x <= sp.bounds[:, 1])
    assert acq(x)[0] <= min_acq + 1e-12


def test_no_valid_point_and_decay(engine):
    w = W.C5S
    sp = _space(w, with_constraint=True, engine=engine)
    sp._constraint_values = np.full_like(sp._constraint_values, 10.0)   # nothing feasible
    gp = HipGPR(kernel=Matern(nu=2.5, length_scale=0.5), alpha=1e-6, normalize_y=True, optimizer=None, engine=engine)
    with pytest.raises(A.NoValidPointRegisteredError):
        A.ExpectedImprovement(xi=0.01).suggest(gp, sp)
    sp2 = _space(W.P1)
    ucb = A.UpperConfidenceBound(kappa=2.0, exploration_decay=0.5, exploration_decay_delay=2)
    for expect in (2.0, 1.0, 0.5):
        ucb.suggest(HipGPR(kernel=Matern(nu=2.5, length_scale=0.4), alpha=1e-6, normalize_y=True, optimizer=None, engine=engine),
                    sp2, n_random=256, n_smart=0, random_state=1)
        assert ucb.kappa == expect


def test_device_theta_search_matches_sklearn_optimum(engine):
    """lml_on_device=True: L-BFGS-B on the host, every objective evaluation on the GPU.  Optimiser end
    points are not bit-comparable (SURVEY.md §7 'theta parity'), so check the optimum: same LML value to
    1e-8 relative, same RandomState consumption, same predictions to 1e-5."""
    w = W.F1
    sp = _space(w)
    r1, r2 = np.random.RandomState(3), np.random.RandomState(3)
    kw = dict(alpha=1e-6, normalize_y=True, n_restarts_optimizer=5)
    sk = GaussianProcessRegressor(kernel=Matern(nu=2.5), random_state=r1, **kw).fit(sp.params, sp.target)
    gp = HipGPR(kernel=Matern(nu=2.5), random_state=r2, engine=engine, lml_on_device=True, **kw).fit(sp.params, sp.target)
    assert r1.uniform() == r2.uniform()
    assert gp.log_marginal_likelihood_value_ == pytest.approx(sk.log_marginal_likelihood_value_, rel=1e-8)
    assert np.allclose(gp.kernel_.theta, sk.kernel_.theta, rtol=1e-4, atol=1e-4)
    th = sk.kernel_.theta
    v1, g1 = sk.log_marginal_likelihood(th, eval_gradient=True)
    v2, g2 = gp.log_marginal_likelihood(th, eval_gradient=True)          # on a fitted model: fit is restored
    assert v2 == pytest.approx(v1, rel=1e-10) and np.allclose(g2, g1, rtol=1e-6, atol=1e-9)
    Xc = sp.random_sample(200, 5)
    m1, s1 = sk.predict(Xc, return_std=True)
    m2, s2 = gp.predict(Xc, return_std=True)
    assert rel_err(m2, m1) < 1e-5 and rel_err(s2, s1) < 1e-5


def test_batched_fd_smart_stage_is_bitwise_the_per_point_path(debug_engine, monkeypatch):
    """The smart stage in its shapes — all runs in lockstep (one device batch of n_seeds * (d + 1) points per
    round; SciPy's setulb driven directly, or one thread per run around the public minimize), one batched (d + 1)-point call per L-BFGS-B iteration of each run, and the reference-shaped per-point
    path — returns exactly the same point: the GEMV kernel evaluates a candidate identically alone or in any
    batch (the path is pinned — GPBO_SMALL_MAX, a debug-build switch; across the GEMV/MFMA switch the agreement is to
    rounding, next test)."""
    engine = debug_engine
    monkeypatch.setenv("GPBO_SMALL_MAX", "1024")
    w = W.P2
    sp = _space(w)
    res = {}
    for mode in ("lockstep", "threads", "batched", "per_point"):
        gp = HipGPR(kernel=RBF(length_scale=0.6), alpha=w.noise, normalize_y=True, optimizer=None, engine=engine)
        fn = A.ExpectedImprovement(xi=0.01)
        fn.device_polish = False                 # the reference-shaped stage (SciPy's iterates) is what this test compares
        fn.batched_fd = mode != "per_point"
        fn.lockstep = {"lockstep": True, "threads": "threads"}.get(mode, False)   # True: SciPy's setulb driven directly
        n0 = [0]
        orig = engine.set_candidates

        def counting(Xc, _o=orig, _n=n0):
            _n[0] += 1
            return _o(Xc)

        engine.set_candidates = counting
        try:
            x = fn.suggest(gp, sp, n_random=2048, n_smart=5, random_state=np.random.RandomState(7))
        finally:
            del engine.set_candidates
        res[mode] = (x, n0[0])
    assert np.array_equal(res["batched"][0], res["per_point"][0])
    assert np.array_equal(res["lockstep"][0], res["per_point"][0])
    assert np.array_equal(res["threads"][0], res["per_point"][0]), (res["threads"], res["per_point"])
    assert res["threads"][1] == res["lockstep"][1], (res["threads"][1], res["lockstep"][1])
    assert res["batched"][1] * 3 < res["per_point"][1]
    assert res["lockstep"][1] * 2 < res["batched"][1]


def test_lockstep_rounds_across_the_kernel_switch(engine):
    """With the default dispatch a lockstep round may run on the MFMA path and a late round (few live runs) on
    the GEMV path; values agree to rounding, so the polished point is the sequential one to optimiser precision."""
    w = W.C2
    sp = _space(w)
    xs = {}
    for lockstep in (True, False):
        gp = HipGPR(kernel=Matern(nu=2.5, length_scale=w.length_scale), alpha=w.noise, normalize_y=True,
                    optimizer=None, engine=engine)
        fn = A.UpperConfidenceBound(kappa=2.576)
        fn.device_polish = False                 # SciPy's L-BFGS-B runs, in lockstep or one after another
        fn.lockstep = lockstep
        xs[lockstep] = fn.suggest(gp, sp, n_random=4096, n_smart=10, random_state=np.random.RandomState(7))
    acq = fn._get_acq(gp)
    f_lock, f_seq = acq(xs[True])[0], acq(xs[False])[0]
    assert abs(f_lock - f_seq) <= 1e-6 * abs(f_seq)
    assert np.allclose(xs[True], xs[False], rtol=0, atol=1e-3)


def test_device_sampling_throughput_mode(engine):
    """gpbo_generate_candidates: Philox4x32-10 on the device, checked bit for bit against a NumPy restatement
    (known-answer), and used end to end by the fused acquisition (labelled non-parity: not the reference's
    RandomState stream)."""
    from helpers import philox4x32_10_uniform

    lo, hi = np.array([0.0, -2.0, 3.0]), np.array([1.0, 2.0, 3.5])
    M, seed = 10001, (123456789 << 31) | 987654321
    engine.generate_candidates(M, lo, hi, seed)
    ref = philox4x32_10_uniform(M, 3, lo, hi, seed)
    rows = engine.get_candidate_rows(np.arange(0, M, 97), 3)
    assert np.array_equal(rows, ref[::97])
    assert np.all(ref >= lo) and np.all(ref < hi) and abs(ref[:, 0].mean() - 0.5) < 0.01
    assert np.all(np.isnan(engine.get_candidate_rows([M + 5, -1], 3)))
    # end to end: the fused random stage on device-generated candidates agrees with the same candidates uploaded
    w = W.P1
    sp = _space(w)
    gp = HipGPR(kernel=Matern(nu=2.5, length_scale=0.4), alpha=w.noise, normalize_y=True, optimizer=None, engine=engine)
    fn = A.UpperConfidenceBound(kappa=2.0)
    fn.device_sampling = True
    rs = np.random.RandomState(5)
    x = fn.suggest(gp, sp, n_random=5000, n_smart=0, random_state=rs)
    rs2 = np.random.RandomState(5)
    seed2 = int(rs2.randint(0, 2**31 - 1)) | (int(rs2.randint(0, 2**31 - 1)) << 31)
    Xc = philox4x32_10_uniform(5000, w.d, sp.bounds[:, 0], sp.bounds[:, 1], seed2)
    m, s = gp.predict(Xc, return_std=True)
    assert np.array_equal(x, Xc[np.argmin(-(m + 2.0 * s))])


def test_incremental_refit_in_a_maximize_loop(engine):
    """optimizer=None: every suggest() refits on the observations so far.  With incremental=True the refits after
    the first are gpbo_fit_append calls; the suggested points are those of from-scratch refits."""
    w = W.C2
    X, y, _ = W.make_observations(w)
    picks = {}
    for incremental in (True, False):
        sp = FloatSpace(w.pbounds())
        sp.register_bulk(X[:300], y[:300])
        gp = HipGPR(kernel=Matern(nu=2.5, length_scale=w.length_scale), alpha=w.noise, normalize_y=True,
                    optimizer=None, engine=engine, incremental=incremental)
        fn = A.UpperConfidenceBound(kappa=2.576)
        calls = []
        orig_fit, orig_app = engine.fit, engine.fit_append
        engine.fit = lambda *a, **k: (calls.append("fit"), orig_fit(*a, **k))[1]
        engine.fit_append = lambda *a, **k: (calls.append("append"), orig_app(*a, **k))[1]
        try:
            out = []
            for it in range(70):                                   # crosses a 64-row padding boundary
                x = fn.suggest(gp, sp, n_random=2048, n_smart=0, random_state=np.random.RandomState(100 + it))
                out.append(x)
                sp.register(x, float(np.sin(3 * x.sum())))
        finally:
            del engine.fit, engine.fit_append
        picks[incremental] = np.array(out)
        assert calls.count("fit") == (1 if incremental else 70)
        assert calls.count("append") == (69 if incremental else 0)
    assert np.array_equal(picks[True], picks[False])


@pytest.mark.parametrize("streams", [2, 7, 64, 256])
def test_device_candidate_substreams_by_jump_ahead_are_the_sequential_stream(debug_engine, streams, monkeypatch):
    """The sub-stream generator (csrc/mt_jump.hip: S start states by polynomial jump-ahead, S workgroups) against the
    reference stream, EVERY value, for sub-stream counts that do and do not divide the block count, an odd stream
    position, doubles that straddle sub-stream boundaries, and the state handed back.  (GPBO_MT_STREAMS: debug build.)"""
    engine = debug_engine
    monkeypatch.setenv("GPBO_MT_STREAMS", str(streams))
    M, d = 150001, 5
    lo = np.linspace(-1.0, 2.0, d)
    hi = lo + np.linspace(0.5, 3.0, d)
    ref, dev = np.random.RandomState(99), np.random.RandomState(99)
    for r in (ref, dev):
        r.randint(0, 2**31 - 1, size=1001)
    want = np.column_stack([ref.uniform(lo[t], hi[t], M) for t in range(d)])
    engine.generate_candidates_like(M, lo, hi, dev)
    got = np.vstack([engine.get_candidate_rows(np.arange(s0, min(M, s0 + 4096)), d) for s0 in range(0, M, 4096)])
    assert np.array_equal(got, want)
    assert np.array_equal(dev.get_state()[1], ref.get_state()[1]) and dev.get_state()[2] == ref.get_state()[2]
    assert np.array_equal(dev.uniform(size=700), ref.uniform(size=700))


@pytest.mark.parametrize("M,d,burn", [(1, 1, 0), (311, 2, 1), (313, 3, 623), (5000, 7, 0), (70000, 16, 12345), (100, 64, 7),
                                      (1 << 17, 8, 3)])
def test_device_candidates_are_the_reference_stream(engine, M, d, burn):
    """gpbo_generate_candidates_mt19937: the resident candidate matrix equals, bit for bit, the per-column
    RandomState.uniform draws of TargetSpace.random_sample (target_space.py:593-600) from any stream position, and the
    RandomState handed back continues the reference's stream."""
    lo = np.linspace(-3.7, 2.0, d)
    hi = lo + np.linspace(0.5, 11.3, d)
    ref, dev = np.random.RandomState(42), np.random.RandomState(42)
    for r in (ref, dev):
        if burn:
            r.randint(0, 2**31 - 1, size=burn)
    want = np.column_stack([ref.uniform(lo[t], hi[t], M) for t in range(d)])
    engine.generate_candidates_like(M, lo, hi, dev)
    idx = np.unique(np.concatenate([np.arange(0, M, max(1, M // 1500)), [M - 1]]))
    assert np.array_equal(engine.get_candidate_rows(idx, d), want[idx])
    assert np.array_equal(dev.uniform(size=700), ref.uniform(size=700))
    assert dev.standard_normal() == ref.standard_normal()


def test_suggest_with_device_stream_is_the_host_stream_suggest(engine):
    """device_sampling="auto" (candidates generated on the device from the caller's RandomState) and False (host
    random_sample + upload) give the same suggestion and leave the RandomState at the same position."""
    w = W.C2
    sp = _space(w)
    out = {}
    for mode in ("auto", False):
        gp = HipGPR(kernel=Matern(nu=2.5, length_scale=w.length_scale), alpha=w.noise, normalize_y=True,
                    optimizer=None, engine=engine)
        fn = A.UpperConfidenceBound(kappa=2.576)
        fn.device_sampling = mode
        rs = np.random.RandomState(11)
        x = fn.suggest(gp, sp, n_random=30000, n_smart=4, random_state=rs)
        out[mode] = (x, rs.uniform())
    assert np.array_equal(out["auto"][0], out[False][0]) and out["auto"][1] == out[False][1]


def test_theta_search_in_lockstep_on_the_device(engine):
    """Default bayes_opt GP configuration (5 restarts), LML on the device: the lockstep search (gpbo_lml_batch) ends
    at the same theta, LML value and RandomState position as the sequential one, bit for bit."""
    w = W.F1
    sp = _space(w)
    out = {}
    for lockstep in (True, False):
        rs = np.random.RandomState(3)
        gp = HipGPR(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5, random_state=rs,
                    engine=engine, lml_on_device=True, theta_lockstep=lockstep).fit(sp.params, sp.target)
        out[lockstep] = (gp.kernel_.theta.copy(), gp.log_marginal_likelihood_value_, rs.uniform())
    assert np.array_equal(out[True][0], out[False][0]) and out[True][1:] == out[False][1:]


def test_predict_warns_on_negative_variances_only_like_sklearn(engine):
    """sklearn warns when it clips a NEGATIVE predicted variance (_gpr.py:479-485) — not on a variance of zero.  The
    device's finalize kernels record the clip (gpbo_take_negative_variance_flag).  fp64: neither sklearn nor the device
    clips anything on this model, so neither warns.  The fp32 posterior mode at the training points of a model with
    noise 1e-8 (true variance far below the mode's 5e-6 error) does clip — flag, exact zeros and the warning appear
    together, and the flag is cleared by reading it."""
    import warnings

    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import Matern

    from bayesianoptimization_amd.gpr import HipGPR

    rs = np.random.RandomState(3)
    X = rs.uniform(size=(600, 3))
    y = np.sin(3 * X.sum(axis=1))
    Xq = np.concatenate([X] * 3)                     # 1800 queries: above every small-batch limit -> the MFMA paths
    k = Matern(nu=2.5, length_scale=0.7)

    def warned(gp):
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            _, sd = gp.predict(Xq, return_std=True)
        return any("Predicted variances smaller than 0" in str(r.message) for r in rec), sd

    sk_w, sd_s = warned(GaussianProcessRegressor(kernel=k, alpha=1e-8, normalize_y=True, optimizer=None).fit(X, y))
    d_w, sd_d = warned(HipGPR(kernel=k, alpha=1e-8, normalize_y=True, optimizer=None, engine=engine).fit(X, y))
    assert d_w == sk_w and not d_w
    assert not (sd_d == 0).any() and not (sd_s == 0).any()
    assert engine.take_negative_variance_flag() is False

    f_w, sd_f = warned(HipGPR(kernel=k, alpha=1e-8, normalize_y=True, optimizer=None, engine=engine, precision="f32").fit(X, y))
    assert f_w and (sd_f == 0).any()
    assert engine.take_negative_variance_flag() is False       # predict() consumed it
    # the small-batch (GEMV) kernels carry the flag too: a 3-point batch in fp64 does not clip
    gp = HipGPR(kernel=k, alpha=1e-8, normalize_y=True, optimizer=None, engine=engine).fit(X, y)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        gp.predict(X[:3], return_std=True)
    assert not rec


MIXED_PB = {"a": (0.0, 2.0), "n": (-3, 7, int), "b": (1.0, 4.0), "c": ("x", "y", "z"), "e": (5.0, 6.0), "f": (0.0, 1.0), "k": (0, 1, int)}


@pytest.mark.parametrize("M", [4099, 150001])
def test_mixed_space_candidates_and_kernel_transform_on_the_device(engine, M):
    """SURVEY.md §8 f3, second half: a space with Float + Int + Categorical parameters.  The float columns come
    from the device generator, the int / categorical ones are drawn on the host at the right position of the SAME MT19937
    stream, `TargetSpace.kernel_transform` (np.round, the reference's one-hot with its batch behaviour) runs as device
    kernels.  Checked bit for bit: the resident matrix == space.random_sample(M, rs) on EVERY row, the RandomState ends
    where the reference's loop leaves it (target_space.py:593-600), and the posterior over the device-transformed
    candidates == the posterior over space.kernel_transform(x) uploaded from the host (same kernels, so equal inputs give
    equal bits)."""
    from bayesianoptimization_amd.float_space import MixedSpace

    sp = MixedSpace(MIXED_PB)
    rng = np.random.RandomState(4)
    Xobs = sp.random_sample(200, rng)
    y = np.sin(Xobs[:, 0] + 0.3 * Xobs[:, 1]) + 0.1 * Xobs[:, 2] + Xobs[:, 4]
    gp = HipGPR(kernel=Matern(nu=2.5, length_scale=1.3), alpha=1e-6, normalize_y=True, optimizer=None, engine=engine,
                transform=sp.kernel_transform).fit(Xobs, y)
    groups = A._mixed_groups_on_device([gp], sp, np.random.RandomState(0), M)
    assert groups is not None
    ref, dev = np.random.RandomState(77), np.random.RandomState(77)
    for r in (ref, dev):
        r.randint(0, 2**31 - 1, size=333)
    want = sp.random_sample(M, ref)
    engine.generate_candidates_mixed(M, groups, dev)
    got = np.vstack([engine.get_candidate_rows(np.arange(s0, min(M, s0 + 4096)), sp.dim) for s0 in range(0, M, 4096)])
    assert np.array_equal(got, want)
    assert np.array_equal(dev.get_state()[1], ref.get_state()[1]) and dev.get_state()[2] == ref.get_state()[2]
    assert dev.uniform() == ref.uniform()
    engine.transform_candidates(groups)
    assert np.array_equal(engine.get_candidate_rows(np.arange(50), sp.dim), want[:50])     # still the rows as drawn
    gp._ensure_resident()
    mu_d, sd_d = engine.posterior(0, float(gp._y_train_mean), float(gp._y_train_std))
    mu_h, sd_h = engine.predict(sp.kernel_transform(want), y_mean=float(gp._y_train_mean), y_std=float(gp._y_train_std))
    assert np.array_equal(mu_d, mu_h) and np.array_equal(sd_d, sd_h)


def test_suggest_over_a_mixed_space_is_the_host_sampling_suggest(engine):
    """The fused random stage over a mixed space with the candidates assembled on the device (device_sampling="auto") and
    with host sampling + host kernel_transform + upload (False): the same suggestion, the same RandomState afterwards."""
    from bayesianoptimization_amd.float_space import MixedSpace

    sp = MixedSpace(MIXED_PB)
    rng = np.random.RandomState(4)
    Xobs = sp.random_sample(300, rng)
    sp.register_bulk(Xobs, np.sin(Xobs[:, 0] + 0.3 * Xobs[:, 1]) + 0.1 * Xobs[:, 2] + Xobs[:, 4])
    out = {}
    for mode in ("auto", False):
        gp = HipGPR(kernel=Matern(nu=2.5, length_scale=1.3), alpha=1e-6, normalize_y=True, optimizer=None, engine=engine,
                    transform=sp.kernel_transform)
        fn = A.UpperConfidenceBound(kappa=2.576)
        fn.device_sampling = mode
        used = []
        orig = engine.generate_candidates_mixed
        engine.generate_candidates_mixed = lambda *a, _o=orig, **k: (used.append(1), _o(*a, **k))[1]
        try:
            rs = np.random.RandomState(21)
            x = fn.suggest(gp, sp, n_random=60000, n_smart=0, random_state=rs)
        finally:
            del engine.generate_candidates_mixed
        out[mode] = (x, rs.uniform(), len(used))
    assert out["auto"][2] == 1 and out[False][2] == 0
    assert np.array_equal(out["auto"][0], out[False][0]) and out["auto"][1] == out[False][1]
