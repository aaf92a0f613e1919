"""CPU: the drop-in glue against the REAL bayes_opt (mounted only in the build container).  The GPU is
replaced by tests/helpers.FakeEngine (oracle-backed test double) so that what is checked here is the
host layer: accelerate() swaps, HipGPR's sklearn duck type inside the reference driver, RandomState
consumption, the fused _random_sample_minimize protocol, state save/load determinism."""
import warnings

import numpy as np
import pytest

from helpers import FakeEngine
from oracle.refenv import have_reference, import_reference

pytestmark = pytest.mark.skipif(not have_reference(), reason="reference not mounted (GPU box)")


def black_box(x, y):
    return -(x**2) - (y - 1) ** 2 + 1


PB = {"x": (2, 4), "y": (-3, 3)}


def _pair(seed=1, constraint=None, acq=None, acq2=None, lml_on_device=False):
    """(reference optimizer, accelerated twin over a FakeEngine).  The bit-for-bit comparisons of this file need theta
    bit for bit, i.e. sklearn's own LML arithmetic in the theta search: lml_on_device=False (accelerate()'s default, "auto",
    evaluates it on the engine, where theta agrees to rounding — test_default_theta_search_runs_on_the_engine_...)."""
    import_reference()
    from bayes_opt import BayesianOptimization

    from bayesianoptimization_amd import accelerate

    ref = BayesianOptimization(f=black_box, pbounds=PB, random_state=seed, verbose=0, constraint=constraint,
                               acquisition_function=acq)
    mine = BayesianOptimization(f=black_box, pbounds=PB, random_state=seed, verbose=0, constraint=constraint,
                                acquisition_function=acq2)
    eng = FakeEngine()
    accelerate(mine, engine=eng, lml_on_device=lml_on_device)
    return ref, mine, eng


def test_accelerate_swaps_gp_and_acquisition():
    from bayesianoptimization_amd import fused_acquisition as A
    from bayesianoptimization_amd.gpr import HipGPR

    ref, mine, eng = _pair()
    assert isinstance(mine._gp, HipGPR) and mine._gp.slot == 0 and mine._gp.engine is eng
    assert mine._gp.random_state is mine._random_state          # the SAME RandomState object (RNG coupling)
    assert isinstance(mine._acquisition_function, A.UpperConfidenceBound)
    assert mine._acquisition_function.kappa == ref._acquisition_function.kappa == 2.576
    assert mine._gp.transform is None                           # all-float space: identity transform skipped
    p1, p2 = ref._gp.get_params(), mine._gp.get_params()
    for k in ("alpha", "normalize_y", "n_restarts_optimizer"):
        assert p1[k] == p2[k]


def test_maximize_trajectory_equals_reference_random_stage():
    """Whole driver loop (suggest -> probe -> register): with the random stage only (n_smart has no knob in
    BayesianOptimization.suggest, so compare through the acquisition's suggest) the points coincide."""
    ref, mine, eng = _pair(seed=3)
    for o in (ref, mine):
        o.maximize(init_points=3, n_iter=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for n_random in (2000, 3000, 2500):     # 2 columns: 3000 and 2500 rows take the device-stream branch
            xr = ref._acquisition_function.suggest(ref._gp, ref._space, n_random=n_random, n_smart=0, fit_gp=True,
                                                   random_state=ref._random_state)
            xm = mine._acquisition_function.suggest(mine._gp, mine._space, n_random=n_random, n_smart=0, fit_gp=True,
                                                    random_state=mine._random_state)
            assert np.array_equal(xr, xm)
            assert np.array_equal(ref._gp.kernel_.theta, mine._gp.kernel_.theta)
            for o, x in ((ref, xr), (mine, xm)):
                o.probe(o._space.array_to_params(x), lazy=False)
    assert ref._random_state.uniform() == mine._random_state.uniform()
    assert [c[1] for c in eng.calls if c[0] == "generate_candidates_like"] == [3000, 2500]
    kinds = [c[0] for c in eng.calls]
    assert "acq_argbest" in kinds and "posterior" in kinds        # the fused path was taken


def test_default_theta_search_runs_on_the_engine_and_agrees_to_rounding():
    """accelerate()'s default: the theta search's LML evaluations go to the engine at every N (profiles/r04_lml_crossover.json)
    — lockstep lanes through lml_batch — and land on the reference's optimum to rounding, with the shared RandomState
    consumed identically (the restarts' starting points are drawn from it, sklearn _gpr.py:325-334)."""
    ref, mine, eng = _pair(seed=3, lml_on_device="auto")
    for o in (ref, mine):
        o.maximize(init_points=4, n_iter=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xr = ref._acquisition_function.suggest(ref._gp, ref._space, n_random=2000, n_smart=0, fit_gp=True, random_state=ref._random_state)
        xm = mine._acquisition_function.suggest(mine._gp, mine._space, n_random=2000, n_smart=0, fit_gp=True, random_state=mine._random_state)
    assert any(c[0] == "lml_batch" for c in eng.calls)                      # N = 4: the engine, not sklearn's host code
    assert np.allclose(ref._gp.kernel_.theta, mine._gp.kernel_.theta, rtol=1e-5, atol=1e-6)
    assert mine._gp.log_marginal_likelihood_value_ == pytest.approx(ref._gp.log_marginal_likelihood_value_, rel=1e-8)
    assert np.allclose(xr, xm, atol=1e-4)
    assert ref._random_state.uniform() == mine._random_state.uniform()


def test_full_maximize_runs_and_improves():
    ref, mine, eng = _pair(seed=5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mine.maximize(init_points=2, n_iter=4)
        ref.maximize(init_points=2, n_iter=4)
    assert len(mine.space) == 6
    assert mine.max["target"] > -10
    # identical init points and identical first suggestion region: same stream for the 2 init points
    assert np.array_equal(mine.space.params[:2], ref.space.params[:2])
    assert np.allclose(mine.space.params[2], ref.space.params[2], atol=1e-5)


def test_constrained_ei_uses_all_slots():
    import_reference()
    from scipy.optimize import NonlinearConstraint

    from bayesianoptimization_amd import fused_acquisition as A
    from bayesianoptimization_amd.gpr import HipGPR

    cons = NonlinearConstraint(lambda x, y: np.cos(x) * np.cos(y) - np.sin(x) * np.sin(y), -np.inf, 0.5)
    ref, mine, eng = _pair(seed=7, constraint=cons)
    assert isinstance(mine._acquisition_function, A.ExpectedImprovement)
    cm = mine._space.constraint._model
    assert len(cm) == 1 and isinstance(cm[0], HipGPR) and cm[0].slot == 1 and cm[0].engine is eng
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for o in (ref, mine):
            o.maximize(init_points=4, n_iter=0)
        xr = ref._acquisition_function.suggest(ref._gp, ref._space, n_random=1500, n_smart=0, random_state=ref._random_state)
        xm = mine._acquisition_function.suggest(mine._gp, mine._space, n_random=1500, n_smart=0, random_state=mine._random_state)
    assert np.array_equal(xr, xm)
    slots = {c[1] for c in eng.calls if c[0] == "posterior"}
    assert slots == {0, 1}


def test_predict_and_state_roundtrip(tmp_path):
    """bayesian_optimization.py:176-260 (predict) and :409-524 (save/load): the accelerated optimizer keeps
    the reference behaviour — after a reload the next suggestion is exactly reproduced."""
    import_reference()
    from bayes_opt import BayesianOptimization

    from bayesianoptimization_amd import accelerate

    ref, mine, eng = _pair(seed=11)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mine.maximize(init_points=2, n_iter=2)
        m, s = mine.predict({"x": 3.0, "y": 0.5}, return_std=True)
        assert np.isscalar(m) or np.ndim(m) == 0 or np.shape(m) == ()
        path = tmp_path / "state.json"
        mine.save_state(path)
        nxt = mine.suggest()
        fresh = BayesianOptimization(f=black_box, pbounds=PB, random_state=11, verbose=0)
        accelerate(fresh, engine=FakeEngine(), lml_on_device=False)     # as `mine` (_pair)
        fresh.load_state(path)
        nxt2 = fresh.suggest()
    assert nxt == nxt2


def test_hipgpr_device_lml_override_semantics():
    """HipGPR.log_marginal_likelihood routes L-BFGS-B's objective to the engine (here the oracle-backed fake)
    with sklearn's conventions: clone_kernel=False mutates kernel_.theta, theta=None returns the stored value,
    unsupported kernels / lml_on_device=False fall through to sklearn's own code, a call on a fitted model
    restores the fit, and the theta search reaches sklearn's optimum with the same RandomState consumption."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import ConstantKernel, Matern, RBF

    from bayesianoptimization_amd.gpr import HipGPR

    rng = np.random.RandomState(2)
    X = rng.uniform(size=(40, 3))
    y = np.sin(2 * X.sum(1))
    eng = FakeEngine()
    r1, r2 = np.random.RandomState(5), np.random.RandomState(5)
    kw = dict(alpha=1e-6, normalize_y=True, n_restarts_optimizer=3)
    sk = GaussianProcessRegressor(kernel=Matern(nu=2.5), random_state=r1, **kw).fit(X, y)
    gp = HipGPR(kernel=Matern(nu=2.5), random_state=r2, engine=eng, lml_on_device=True, **kw).fit(X, y)
    assert any(c[0] in ("lml", "lml_batch") for c in eng.calls) and eng.calls[-1][0] == "fit"
    assert r1.uniform() == r2.uniform()
    assert gp.log_marginal_likelihood_value_ == pytest.approx(sk.log_marginal_likelihood_value_, rel=1e-9)
    assert gp.log_marginal_likelihood() == gp.log_marginal_likelihood_value_
    th = np.log([0.7])
    v_s, g_s = sk.log_marginal_likelihood(th, eval_gradient=True)
    n_fit = sum(c[0] == "fit" for c in eng.calls)
    v, g = gp.log_marginal_likelihood(th, eval_gradient=True)
    assert v == pytest.approx(v_s, rel=1e-10) and np.allclose(g, g_s, rtol=1e-7)
    assert sum(c[0] == "fit" for c in eng.calls) == n_fit + 1          # the slot's fit was restored
    assert not np.allclose(gp.kernel_.theta, th)                          # clone_kernel=True: kernel_ untouched
    gp.predict(X[:3], return_std=True)                                    # and the model still predicts
    # anisotropic RBF under a fixed unit constant factor is supported; a free constant is not
    gp2 = HipGPR(kernel=ConstantKernel(1.0, "fixed") * RBF([1.0, 1.0, 1.0]), engine=eng, lml_on_device=True,
                 alpha=1e-6, optimizer=None).fit(X, y)
    assert gp2._device_lml_ok(gp2.kernel_)
    gp3 = HipGPR(kernel=Matern(nu=2.5), engine=eng, lml_on_device=False, alpha=1e-6, optimizer=None).fit(X, y)
    n_lml = sum(c[0] == "lml" for c in eng.calls)
    gp3.log_marginal_likelihood(th, eval_gradient=True)
    assert sum(c[0] == "lml" for c in eng.calls) == n_lml
    # "auto": the device at every N since round 4 (the measured crossover: profiles/r04_lml_crossover.json) ...
    assert HipGPR(kernel=Matern(nu=2.5), engine=eng, lml_on_device="auto", alpha=1e-6,
                  optimizer=None).fit(X, y)._device_lml_ok(Matern(nu=2.5))
    # ... for the kernels the device evaluates: anything else keeps sklearn's host code
    from sklearn.gaussian_process.kernels import WhiteKernel
    assert not HipGPR(kernel=Matern(nu=2.5), engine=eng, lml_on_device="auto", alpha=1e-6,
                      optimizer=None).fit(X, y)._device_lml_ok(Matern(nu=2.5) + WhiteKernel())


def test_meta_acquisitions_run_through_the_seams():
    """GPHedge and ConstantLiar (bayes_opt/acquisition.py:952-1360) stay reference code; accelerate() swaps the GP and
    the stock policies they delegate to, so every base `suggest(..., fit_gp=False)` runs the fused random stage — and the
    optimisation visits exactly the points the un-accelerated reference visits (same RandomState consumption)."""
    import_reference()
    from bayes_opt import BayesianOptimization, acquisition

    from bayesianoptimization_amd import accelerate
    from bayesianoptimization_amd import fused_acquisition as A
    from bayesianoptimization_amd.gpr import HipGPR

    for make in (lambda: acquisition.GPHedge([acquisition.UpperConfidenceBound(kappa=2.0),
                                               acquisition.ExpectedImprovement(xi=0.01)]),
                 lambda: acquisition.ConstantLiar(acquisition.UpperConfidenceBound(kappa=2.0))):
        ref = BayesianOptimization(f=black_box, pbounds=PB, random_state=4, verbose=0, acquisition_function=make())
        opt = BayesianOptimization(f=black_box, pbounds=PB, random_state=4, verbose=0, acquisition_function=make())
        eng = FakeEngine()
        accelerate(opt, engine=eng, lml_on_device=False)      # bit-for-bit trajectories: sklearn's LML arithmetic
        assert isinstance(opt._gp, HipGPR)
        meta = opt._acquisition_function
        assert type(meta).__module__.startswith("bayes_opt")                         # meta policy untouched
        bases = getattr(meta, "base_acquisitions", None) or [meta.base_acquisition]
        assert all(isinstance(b, A.AcquisitionFunction) for b in bases)               # ... its delegates are fused
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref.maximize(init_points=2, n_iter=3)
            opt.maximize(init_points=2, n_iter=3)
        assert len(opt.space) == 5
        # (the local searches end within optimiser precision of each other: the engine double and sklearn differ in the
        # last bits of mu/sigma and L-BFGS-B amplifies that to ~1e-5 in x)
        assert np.allclose(opt.space.params, ref.space.params, rtol=0, atol=1e-4)
        assert ref._random_state.uniform() == opt._random_state.uniform()
        kinds = [c[0] for c in eng.calls]
        assert "fit" in kinds and "acq_argbest" in kinds                              # the device random stage ran


def test_integer_parameters_use_the_host_transform():
    """Non-float parameters (bayes_opt/parameter.py:236-320): the engine sees kernel-transformed coordinates
    (rounded integers) through HipGPR.transform, and the random stage still matches the reference exactly."""
    import_reference()
    from bayes_opt import BayesianOptimization

    from bayesianoptimization_amd import accelerate

    def f(x, k):
        return -(x - 3) ** 2 - (k - 4) ** 2

    pb = {"x": (2.0, 4.0), "k": (0, 10, int)}
    ref = BayesianOptimization(f=f, pbounds=pb, random_state=9, verbose=0)
    mine = BayesianOptimization(f=f, pbounds=pb, random_state=9, verbose=0)
    eng = FakeEngine()
    accelerate(mine, engine=eng, lml_on_device=False)
    assert mine._gp.transform is not None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for o in (ref, mine):
            o.maximize(init_points=4, n_iter=0)
        xr = ref._acquisition_function.suggest(ref._gp, ref._space, n_random=800, n_smart=0, random_state=ref._random_state)
        xm = mine._acquisition_function.suggest(mine._gp, mine._space, n_random=800, n_smart=0, random_state=mine._random_state)
    assert np.array_equal(xr, xm)
    fit_X = [c for c in eng.calls if c[0] == "fit"]
    assert fit_X and fit_X[-1][2] == (4, 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mine.maximize(init_points=0, n_iter=2)      # mixed space: differential evolution smart stage on the host
    assert len(mine.space) == 6


def test_refit_with_appended_rows_goes_through_fit_append():
    """HipGPR.fit at fixed theta (optimizer=None) on the previous inputs plus new rows grows the held factorisation
    (gpbo_fit_append) instead of refactorising; anything that invalidates the held model — another theta, a
    changed earlier row, an LML evaluation on the slot, another estimator on the slot — falls back to gpbo_fit."""
    from sklearn.gaussian_process.kernels import Matern

    from bayesianoptimization_amd.gpr import HipGPR

    rng = np.random.RandomState(0)
    X = rng.uniform(size=(30, 3))
    y = np.sin(X.sum(1))
    eng = FakeEngine()
    gp = HipGPR(kernel=Matern(nu=2.5, length_scale=0.7), alpha=1e-6, normalize_y=True, optimizer=None, engine=eng)

    def last():
        return [c[0] for c in eng.calls if c[0] in ("fit", "fit_append", "lml")][-1]

    gp.fit(X[:20], y[:20]); assert last() == "fit"
    gp.fit(X[:21], y[:21]); assert last() == "fit_append" and eng.calls[-1][2] == (1, 3)
    gp.fit(X[:25], y[:25]); assert last() == "fit_append" and eng.calls[-1][2] == (4, 3)
    gp.fit(X[:25], 2 * y[:25]); assert last() == "fit_append" and eng.calls[-1][2] == (0, 3)   # new targets only
    fresh = HipGPR(kernel=Matern(nu=2.5, length_scale=0.7), alpha=1e-6, normalize_y=True, optimizer=None,
                   engine=FakeEngine()).fit(X[:25], 2 * y[:25])
    q = rng.uniform(size=(7, 3))
    for a, b in zip(gp.predict(q, return_std=True), fresh.predict(q, return_std=True)):
        assert np.allclose(a, b, rtol=1e-12, atol=1e-14)
    assert np.allclose(gp.alpha_, fresh.alpha_, rtol=1e-10)
    X2 = X.copy(); X2[3, 0] += 0.1
    gp.fit(X2[:26], y[:26]); assert last() == "fit"                          # an earlier row changed
    gp.fit(X2[:27], y[:27]); assert last() == "fit_append"
    gp.log_marginal_likelihood(gp.kernel_.theta)                             # host LML: slot untouched
    gp.fit(X2[:28], y[:28]); assert last() == "fit_append"
    gp.set_params(kernel=Matern(nu=2.5, length_scale=0.9))
    gp.fit(X2[:29], y[:29]); assert last() == "fit"                          # another theta
    other = HipGPR(kernel=Matern(nu=2.5, length_scale=0.9), alpha=1e-6, normalize_y=True, optimizer=None, engine=eng)
    other.fit(X2[:10], y[:10])
    gp.fit(X2[:30], y[:30]); assert last() == "fit"                          # the slot was taken by someone else
    gp.set_params(incremental=False)
    gp.fit(X2[:30], y[:30]); assert last() == "fit"


def test_theta_search_runs_in_lockstep_equal_the_sequential_runs(monkeypatch):
    """HipGPR.fit with restarts: the independent L-BFGS-B runs advanced together over batched LML evaluations
    (gpbo_lml_batch) end at the same theta, the same LML and the same RandomState position as one run after another."""
    from sklearn.gaussian_process.kernels import Matern

    from bayesianoptimization_amd.gpr import HipGPR

    rng = np.random.RandomState(4)
    X = rng.uniform(size=(40, 3))
    y = np.sin(3 * X.sum(1)) + 0.05 * rng.standard_normal(40)
    from bayesianoptimization_amd import lbfgsb_lockstep

    fits = {}
    for lockstep in (True, "threads", False):
        if lockstep == "threads":       # the public-API fallback: one thread per run around sklearn's own optimiser call
            monkeypatch.setattr(lbfgsb_lockstep, "driver_available", lambda: False)
        eng = FakeEngine()
        rs = np.random.RandomState(8)
        gp = HipGPR(kernel=Matern(nu=2.5, length_scale=1.0), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5,
                    random_state=rs, engine=eng, lml_on_device=True, theta_lockstep=bool(lockstep)).fit(X, y)
        fits[lockstep] = (gp.kernel_.theta.copy(), gp.log_marginal_likelihood_value_, rs.uniform(), eng.calls)
    for mode in (True, "threads"):
        assert np.array_equal(fits[mode][0], fits[False][0]) and fits[mode][1] == fits[False][1]
        assert fits[mode][2] == fits[False][2]
    assert [c for c in fits["threads"][3] if c[0] == "lml_batch"] == [c for c in fits[True][3] if c[0] == "lml_batch"]
    batches = [c[1] for c in fits[True][3] if c[0] == "lml_batch"]
    singles = [c for c in fits[False][3] if c[0] == "lml"]
    assert batches and batches[0] == 6 and sum(batches) == len(singles) and len(batches) * 2 < len(singles)
    assert not [c for c in fits[True][3] if c[0] == "lml"]


def test_fused_gphedge_is_the_reference_gphedge_and_can_share_one_posterior_pass():
    """fused_acquisition.GPHedge: in its default mode it IS bayes_opt's GPHedge (acquisition.py:1181-1360)
    — same nominees, gains, RandomState position through a maximize() loop, same parameter dict; with
    share_candidates=True all base policies are served by ONE candidate set and ONE posterior pass per suggest()."""
    import_reference()
    from bayes_opt import BayesianOptimization, acquisition

    from bayesianoptimization_amd import accelerate
    from bayesianoptimization_amd import fused_acquisition as A

    ref = BayesianOptimization(f=black_box, pbounds=PB, random_state=6, verbose=0, acquisition_function=acquisition.GPHedge(
        [acquisition.UpperConfidenceBound(kappa=2.0), acquisition.ExpectedImprovement(xi=0.01),
         acquisition.ProbabilityOfImprovement(xi=0.02)]))
    fused = A.GPHedge([A.UpperConfidenceBound(kappa=2.0), A.ExpectedImprovement(xi=0.01), A.ProbabilityOfImprovement(xi=0.02)])
    opt = BayesianOptimization(f=black_box, pbounds=PB, random_state=6, verbose=0, acquisition_function=fused)
    eng = FakeEngine()
    accelerate(opt, engine=eng, lml_on_device=False)
    assert opt._acquisition_function is fused
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref.maximize(init_points=2, n_iter=4)
        opt.maximize(init_points=2, n_iter=4)
    assert np.allclose(opt.space.params, ref.space.params, rtol=0, atol=1e-4)
    assert np.allclose(fused.gains, ref._acquisition_function.gains, rtol=0, atol=1e-4)
    assert ref._random_state.uniform() == opt._random_state.uniform()
    pr, pf = ref._acquisition_function.get_acquisition_params(), fused.get_acquisition_params()
    assert sorted(pr) == sorted(pf) and len(pf["base_acquisitions_params"]) == 3
    clone = A.GPHedge([A.UpperConfidenceBound(), A.ExpectedImprovement(xi=0.5), A.ProbabilityOfImprovement(xi=0.5)])
    clone.set_acquisition_params(pf)
    assert np.array_equal(clone.gains, fused.gains) and clone.base_acquisitions[1].xi == 0.01
    with pytest.raises(TypeError, match="ambiguous"):
        fused.base_acq(0.0, 1.0)

    # shared mode: one candidate draw, one posterior pass per GP, one arg-best pass per policy
    shared = A.GPHedge([A.UpperConfidenceBound(kappa=2.0), A.ExpectedImprovement(xi=0.01), A.ProbabilityOfImprovement(xi=0.02)],
                       share_candidates=True)
    opt2 = BayesianOptimization(f=black_box, pbounds=PB, random_state=6, verbose=0, acquisition_function=shared)
    eng2 = FakeEngine()
    accelerate(opt2, engine=eng2, lml_on_device=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        opt2.maximize(init_points=2, n_iter=1)
        eng2.calls.clear()
        x = opt2.suggest()
    kinds = [c[0] for c in eng2.calls]
    start = kinds.index("generate_candidates_like")           # (before it: the gains update, a 3-point predict)
    rest = kinds[start + 1:]
    stop = rest.index("set_candidates") if "set_candidates" in rest else len(rest)      # first local-search predict
    stage = rest[:stop]
    assert kinds.count("generate_candidates_like") == 1
    assert stage.count("posterior") == 1 and stage.count("acq_argbest") == 3      # ONE pass, three selections
    assert all(PB[k][0] <= v <= PB[k][1] for k, v in x.items())
    assert shared.previous_candidates.shape == (3, 2) and [b.i for b in shared.base_acquisitions] == [2, 2, 2]


def test_unsupported_kernel_degrades_to_the_reference_trajectory():
    """`set_gp_params(kernel=Matern(nu=1.5))` on an accelerated optimizer (bayes_opt/bayesian_optimization.py:403-407) is a slower
    step, not an exception (SURVEY.md §2 "Third-party kernels"): the model runs scikit-learn's own fit / predict — one
    UserWarning — the fused acquisition runs over its `predict`, and the whole maximize() trajectory is the reference's bit for
    bit, RandomState position included.  Setting a supported kernel afterwards puts the model back on the engine."""
    from sklearn.gaussian_process.kernels import Matern

    ref, mine, eng = _pair(seed=11)
    ref.set_gp_params(kernel=Matern(nu=1.5), n_restarts_optimizer=2)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        mine.set_gp_params(kernel=Matern(nu=1.5), n_restarts_optimizer=2)          # said here: the fits run with warnings silenced
        mine.maximize(init_points=3, n_iter=3)
    said = [w for w in seen if issubclass(w.category, UserWarning) and "HIP path supports Matern(nu=2.5) only" in str(w.message)]
    assert len(said) == 1, [str(w.message) for w in seen]                      # once per estimator, not once per fit
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref.maximize(init_points=3, n_iter=3)
    assert mine._gp._host_mode and not [c for c in eng.calls if c[0] in ("fit", "posterior", "lml", "lml_batch")]
    assert np.array_equal(mine.space.params, ref.space.params)
    assert np.array_equal(mine.space.target, ref.space.target)
    assert np.array_equal(mine._gp.kernel_.theta, ref._gp.kernel_.theta)
    assert ref._random_state.uniform() == mine._random_state.uniform()
    # back on the engine with a kernel it evaluates
    mine.set_gp_params(kernel=Matern(nu=2.5))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mine.maximize(init_points=0, n_iter=1)
    assert not mine._gp._host_mode and any(c[0] == "fit" for c in eng.calls)


def test_accelerate_with_an_unsupported_kernel_warns_once_and_keeps_the_optimizer_usable():
    import_reference()
    from bayes_opt import BayesianOptimization
    from sklearn.gaussian_process.kernels import RationalQuadratic

    from bayesianoptimization_amd import accelerate

    ref = BayesianOptimization(f=black_box, pbounds=PB, random_state=4, verbose=0)
    mine = BayesianOptimization(f=black_box, pbounds=PB, random_state=4, verbose=0)
    for o in (ref, mine):
        o.set_gp_params(kernel=RationalQuadratic())
    eng = FakeEngine()
    with pytest.warns(UserWarning, match="the target GP"):
        accelerate(mine, engine=eng)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        mine.maximize(init_points=2, n_iter=2)
    assert not [w for w in seen if "HIP path" in str(w.message)]               # accelerate() said it already
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref.maximize(init_points=2, n_iter=2)
    assert np.array_equal(mine.space.params, ref.space.params)


def test_default_accelerate_path_reaches_the_references_acquisition_value():
    """accelerate()'s DEFAULT configuration — theta search on the engine (`lml_on_device="auto"`) and the local searches as one
    engine call (`local_search="auto"`, gpbo_polish_seeds) — is not bit for bit the reference: the same candidate
    stage, the same seeds, another optimiser.  What must hold: after identical histories the point it suggests has an acquisition
    value at least as good as the reference's own suggestion, up to the optimisers' tolerances (here: the oracle-backed engine
    runs SciPy's L-BFGS-B with the analytic gradient where the reference differentiates numerically)."""
    import_reference()
    from bayes_opt import BayesianOptimization

    from bayesianoptimization_amd import accelerate
    from oracle import gp_oracle as O

    ref = BayesianOptimization(f=black_box, pbounds=PB, random_state=5, verbose=0)
    mine = BayesianOptimization(f=black_box, pbounds=PB, random_state=5, verbose=0)
    eng = FakeEngine()
    accelerate(mine, engine=eng)                       # every default
    rng = np.random.RandomState(0)
    for _ in range(12):                                # identical histories
        p = {"x": float(rng.uniform(2, 4)), "y": float(rng.uniform(-3, 3))}
        t = black_box(**p)
        ref.register(params=p, target=t)
        mine.register(params=p, target=t)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x_ref = ref._space.params_to_array(ref.suggest())
        x_mine = mine._space.params_to_array(mine.suggest())
    assert any(c[0] == "polish_seeds" for c in eng.calls) and any(c[0] == "lml_batch" for c in eng.calls)    # the default path ran
    # the acquisition both maximise, under the reference's own fitted GP (UCB, kappa = 2.576: bayes_opt's default)
    gp = ref._gp
    def ucb(x):
        mu, sd = gp.predict(np.asarray(x, dtype=np.float64).reshape(1, -1), return_std=True)
        return float(mu[0] + 2.576 * sd[0])
    a_ref, a_mine = ucb(x_ref), ucb(x_mine)
    scale = max(1.0, abs(a_ref))
    assert a_mine >= a_ref - 1e-5 * scale, (a_mine, a_ref, x_mine, x_ref)
    # ... and theta agrees to rounding (the engine's LML is the oracle's arithmetic, sklearn's up to the last bits)
    assert np.allclose(mine._gp.kernel_.theta, ref._gp.kernel_.theta, rtol=1e-6, atol=1e-8)


def test_a_space_wider_than_the_engine_degrades_to_the_reference_trajectory():
    """Five 16-way CategoricalParameters are 80 columns in kernel space (one-hot: bayes_opt/parameter.py:434-449 through
    target_space.py:340-347) — more than GPBO_MAX_DIM = 64.  Such a model does not reach gpbo_fit (which would answer
    GPBO_ERR_INVALID from inside suggest()): HipGPR runs scikit-learn's own fit / predict with one UserWarning and the whole
    maximize() trajectory is the reference's, RandomState position included (VERDICT r5 missing #3)."""
    import_reference()
    from bayes_opt import BayesianOptimization

    from bayesianoptimization_amd import accelerate
    from bayesianoptimization_amd._lib import MAX_DIM

    cats = tuple(f"v{i:02d}" for i in range(16))
    pb = {f"c{j}": cats for j in range(5)}
    pb["x"] = (0.0, 1.0)

    def f(x, **c):
        return -(x - 0.3) ** 2 + 0.1 * sum(cats.index(v) == 3 + j for j, v in enumerate(c[k] for k in sorted(c)))

    ref = BayesianOptimization(f=f, pbounds=pb, random_state=5, verbose=0)
    mine = BayesianOptimization(f=f, pbounds=pb, random_state=5, verbose=0)
    eng = FakeEngine()
    with pytest.warns(UserWarning, match="81 columns in kernel space"):      # said by accelerate(): the fits run with warnings silenced
        accelerate(mine, engine=eng, lml_on_device=False)
    assert mine._gp._device_width(mine._space.random_sample(2, random_state=np.random.RandomState(0))) == 81 > MAX_DIM
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        mine.maximize(init_points=3, n_iter=3)
    assert not [w for w in seen if "HIP path" in str(w.message)]               # ... and only there
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref.maximize(init_points=3, n_iter=3)
    assert mine._gp._host_mode and not [c for c in eng.calls if c[0] in ("fit", "posterior", "lml", "lml_batch")]
    assert mine._gp._host_warned.startswith("HIP path supports up to 64 dimensions in kernel space")
    assert np.array_equal(mine.space.params, ref.space.params)
    assert np.array_equal(mine.space.target, ref.space.target)
    assert np.array_equal(mine._gp.kernel_.theta, ref._gp.kernel_.theta)
    assert ref._random_state.uniform() == mine._random_state.uniform()
