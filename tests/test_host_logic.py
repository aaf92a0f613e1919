"""CPU: host-side mirror of the reference interface (space, acquisition orchestration, merge rules),
checked against the reference itself when /root/reference is mounted."""
import numpy as np
import pytest

from bayesianoptimization_amd import fused_acquisition as A
from bayesianoptimization_amd import workloads as W
from bayesianoptimization_amd.distributed import merge_best, shard_range
from bayesianoptimization_amd.float_space import FloatSpace, ensure_rng
from oracle.refenv import have_reference, import_reference

needs_ref = pytest.mark.skipif(not have_reference(), reason="reference not mounted (GPU box)")


def test_ensure_rng():
    assert isinstance(ensure_rng(None), np.random.RandomState)
    r = np.random.RandomState(3)
    assert ensure_rng(r) is r
    assert ensure_rng(5).uniform() == np.random.RandomState(5).uniform()
    with pytest.raises(TypeError):
        ensure_rng("x")


def test_shard_range_partitions_exactly():
    for M in (1, 7, 128, 1 << 20, 1000003):
        for G in (1, 2, 3, 8):
            spans = [shard_range(M, G, r) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == M
            assert all(spans[i][1] == spans[i + 1][0] for i in range(G - 1))
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_merge_best_rules():
    # lowest index among equal minima; -0.0 == 0.0
    bi, bv, si, sv = merge_best([0.0, -0.0], [7, 3], [[0.0, 1.0], [-0.0, 2.0]], [[7, 8], [3, 4]], 3)
    assert bi == 3 and list(si) == [3, 7, 8]
    # first NaN wins arg-best, NaNs sort last among seeds, padding (-1) dropped
    bi, bv, si, sv = merge_best([1.0, np.nan], [2, 50], [[1.0, np.nan], [np.nan, np.nan]], [[2, 9], [50, -1]], 4)
    assert bi == 50 and np.isnan(bv) and list(si) == [2, 9, 50]
    # merge equals a global sort
    rng = np.random.RandomState(0)
    ys = rng.randn(1000)
    ys[rng.randint(0, 1000, 30)] = ys[5]  # ties
    k = 10
    parts = [(0, 400), (400, 1000)]
    sv_, si_, bv_, bi_ = [], [], [], []
    for s, e in parts:
        o = np.lexsort((np.arange(s, e), ys[s:e]))[:k]
        si_.append(o + s); sv_.append(ys[s:e][o]); bi_.append(o[0] + s); bv_.append(ys[s:e][o[0]])
    bi, bv, si, sv = merge_best(bv_, bi_, sv_, si_, k)
    ref = np.lexsort((np.arange(1000), ys))[:k]
    assert bi == ref[0] and np.array_equal(si, ref)


def test_floatspace_protocol_and_target_max():
    sp = FloatSpace({"a": (0, 1), "b": (-2, 2)})
    assert sp.empty and sp._target_max() is None
    sp.register([0.5, 0.0], 1.0)
    sp.register([0.5, 5.0], 9.0)  # out of bounds -> masked
    assert len(sp) == 2 and sp._target_max() == 1.0
    x = sp.random_sample(5, np.random.RandomState(1))
    assert x.shape == (5, 2) and np.all(x[:, 1] >= -2)
    assert sp.random_sample(0, 3).shape == (2,)


@needs_ref
def test_candidate_stream_equals_reference():
    import_reference()
    from bayes_opt.target_space import TargetSpace

    for w in (W.C1, W.C2, W.C5S):
        ts = TargetSpace(None, w.pbounds())
        a = ts.random_sample(777, np.random.RandomState(7))
        b = FloatSpace(w.pbounds()).random_sample(777, np.random.RandomState(7))
        c = W.make_candidates(w.bounds_array(), 777, 7)
        assert np.array_equal(a, b) and np.array_equal(a, c)
        assert np.array_equal(ts.bounds, FloatSpace(w.pbounds()).bounds)


def _mk_spaces(w, n=40):
    X, y, c = W.make_observations(w)
    sp = FloatSpace(w.pbounds())
    sp.register_bulk(X[:n], y[:n])
    return sp, X[:n], y[:n]


@needs_ref
@pytest.mark.parametrize("kind", ["ucb", "ei", "poi"])
@pytest.mark.parametrize("n_smart", [0, 3])
def test_host_orchestration_equals_reference(kind, n_smart):
    """Same GP object, same seeds: the restated suggest()/_acq_min/_smart_minimize give the SAME point
    as bayes_opt's (non-fused path, plain sklearn GP), and consume the RandomState identically."""
    import_reference()
    from bayes_opt import acquisition as RA
    from bayes_opt.target_space import TargetSpace
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import Matern

    w = W.P1
    sp, X, y = _mk_spaces(w)
    ts = TargetSpace(None, w.pbounds())
    for i in range(len(y)):
        ts.register(X[i], y[i])
    mine = {"ucb": A.UpperConfidenceBound(kappa=1.7, exploration_decay=0.9), "ei": A.ExpectedImprovement(xi=0.02),
            "poi": A.ProbabilityOfImprovement(xi=0.02)}[kind]
    ref = {"ucb": RA.UpperConfidenceBound(kappa=1.7, exploration_decay=0.9), "ei": RA.ExpectedImprovement(xi=0.02),
           "poi": RA.ProbabilityOfImprovement(xi=0.02)}[kind]
    r1, r2 = np.random.RandomState(11), np.random.RandomState(11)
    gp1 = GaussianProcessRegressor(kernel=Matern(nu=2.5, length_scale=0.4), alpha=1e-6, normalize_y=True, optimizer=None)
    gp2 = GaussianProcessRegressor(kernel=Matern(nu=2.5, length_scale=0.4), alpha=1e-6, normalize_y=True, optimizer=None)
    for _ in range(2):
        x1 = mine.suggest(gp1, sp, n_random=500, n_smart=n_smart, random_state=r1)
        x2 = ref.suggest(gp2, ts, n_random=500, n_smart=n_smart, random_state=r2)
        assert np.array_equal(x1, x2)
    assert r1.uniform() == r2.uniform()  # identical stream position
    assert mine.i == ref.i == 2
    assert mine.get_acquisition_params() == ref.get_acquisition_params()


def test_error_behaviour_matches_reference_contract():
    sp = FloatSpace({"a": (0, 1)})
    ucb = A.UpperConfidenceBound()
    with pytest.raises(A.TargetSpaceEmptyError):
        ucb.suggest(None, sp)
    with pytest.raises(ValueError):
        A.UpperConfidenceBound(kappa=-1)
    with pytest.raises(ValueError):
        A.ExpectedImprovement(xi=0.1, exploration_decay=1.5)
    with pytest.raises(ValueError):
        A.ExpectedImprovement(xi=0.1, exploration_decay_delay=-1)
    with pytest.warns(DeprecationWarning):
        A.UpperConfidenceBound(random_state=1)
    ei = A.ExpectedImprovement(xi=0.1)
    with pytest.raises(ValueError, match="y_max is not set"):
        ei.base_acq(np.zeros(2), np.ones(2))

    class Cons:  # any constraint makes UCB refuse (acquisition.py:524-529)
        pass

    sp2 = FloatSpace({"a": (0, 1)}, constraint=Cons())
    sp2._params = np.zeros((1, 1)); sp2._target = np.zeros(1); sp2._constraint_values = np.zeros(1)
    with pytest.raises(A.ConstraintNotSupportedError):
        ucb.suggest(None, sp2)
    sp.register([0.3], 1.0)
    with pytest.raises(ValueError, match="Either n_random or n_smart"):
        from sklearn.gaussian_process import GaussianProcessRegressor
        ucb.suggest(GaussianProcessRegressor(), sp, n_random=0, n_smart=0)


def test_mock_gp_duck_typing_and_seed_containment():
    """tests/test_acquisition.py:90-132 of the reference: an analytic bowl, x_min must be among the seeds."""
    class MockAcq(A.AcquisitionFunction):
        def _get_acq(self, gp, constraint=None):
            return lambda x: (3 - x[..., 0]) ** 2 + (1 - x[..., 1]) ** 2
        def base_acq(self, mean, std):
            pass

    sp = FloatSpace({"x": (1, 4), "y": (0, 3)})
    acq = MockAcq()
    rs = np.random.RandomState(0)
    x_min, min_acq, x_seeds = acq._random_sample_minimize(acq._get_acq(None), sp, rs, n_random=1000, n_x_seeds=5)
    assert any(np.array_equal(x_min, s) for s in x_seeds)
    best = acq._acq_min(acq._get_acq(None), sp, rs, n_random=1000, n_smart=5)
    assert np.allclose(best, [3, 1], atol=1e-4)


def test_hipgpr_fails_loudly_without_gpu():
    from bayesianoptimization_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible here")
    from sklearn.gaussian_process.kernels import Matern
    from bayesianoptimization_amd.gpr import HipGPR, describe_kernel
    from sklearn.gaussian_process.kernels import RationalQuadratic

    gp = HipGPR(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, optimizer=None)
    with pytest.raises(_lib.GpboError):
        gp.fit(np.random.rand(5, 2), np.random.rand(5))
    with pytest.raises(NotImplementedError):
        describe_kernel(RationalQuadratic())
    with pytest.raises(NotImplementedError):
        describe_kernel(Matern(nu=1.5))


def test_batched_finite_differences_reproduce_scipy_lbfgsb():
    """_fd_value_and_grad hands L-BFGS-B the forward differences SciPy would form itself (abs_step 1e-8, the
    one-sided bound adjustment): same iterates bit for bit, ~d times fewer acquisition calls."""
    from scipy.optimize import minimize

    from bayesianoptimization_amd.fused_acquisition import _fd_value_and_grad

    rng = np.random.RandomState(0)
    A = rng.randn(6, 6)
    A = A @ A.T + np.eye(6)
    b = rng.randn(6)

    def f1(x):
        return 0.5 * x @ A @ x - x @ b + np.sin(3 * x).sum()

    calls = [0]

    def acq(x):  # batch evaluation that is bitwise the per-point evaluation (as the device's small path is)
        calls[0] += 1
        return np.array([f1(r) for r in np.atleast_2d(x)])

    bounds = np.array([[-1.0, 1.0]] * 6)
    for x0 in (rng.uniform(-1, 1, 6), np.array([1.0, -1.0, 0.3, 1.0, 0, -1.0]), np.zeros(6)):
        calls[0] = 0
        r1 = minimize(f1, x0, bounds=bounds, method="L-BFGS-B")
        r2 = minimize(_fd_value_and_grad(acq, bounds), x0, jac=True, bounds=bounds, method="L-BFGS-B")
        assert np.array_equal(r1.x, r2.x) and r1.fun == r2.fun and r1.nit == r2.nit
        assert calls[0] * 5 < r1.nfev


def test_lockstep_runs_are_the_sequential_runs():
    """_polish_in_lockstep advances all L-BFGS-B runs together (one merged batch per round) and every run ends
    exactly where it ends alone; runs of different lengths retire without stalling the rest; an objective that
    raises propagates to the caller."""
    from scipy.optimize import minimize

    from bayesianoptimization_amd.fused_acquisition import _fd_value_and_grad, _polish_in_lockstep

    rng = np.random.RandomState(3)
    A = rng.randn(5, 5)
    A = A @ A.T + np.eye(5)

    def f1(x):
        return 0.5 * x @ A @ x + np.cos(4 * x).sum()

    calls = []

    def acq(x):
        calls.append(len(x))
        return np.array([f1(r) for r in np.atleast_2d(x)])

    bounds = np.array([[-2.0, 2.0]] * 5)
    seeds = np.vstack([rng.uniform(-2, 2, (7, 5)), np.full((1, 5), 2.0)])
    alone = [minimize(_fd_value_and_grad(acq, bounds), s, jac=True, bounds=bounds, method="L-BFGS-B") for s in seeds]
    n_alone = len(calls)
    calls.clear()
    together = _polish_in_lockstep(acq, seeds, bounds)
    for a, b in zip(alone, together):
        assert np.array_equal(a.x, b.x) and a.fun == b.fun and a.nit == b.nit and a.nfev == b.nfev
    assert len(calls) == max(r.nfev for r in alone) and len(calls) * 3 < n_alone
    assert calls[0] == len(seeds) * 6 and calls[-1] < calls[0]          # runs retire at different rounds

    def broken(x):
        if len(calls) > 3:
            raise FloatingPointError("boom")
        return acq(x)

    calls.clear()
    with pytest.raises(FloatingPointError):
        _polish_in_lockstep(broken, seeds, bounds)


@pytest.mark.parametrize("M,d,burn", [(1, 1, 0), (5, 3, 0), (311, 2, 1), (312, 2, 7), (313, 1, 623), (1000, 4, 624),
                                      (2000, 16, 12345)])
def test_mt19937_block_walk_reproduces_randomstate_uniform(M, d, burn):
    """The block walk of csrc/mt19937.hip (mirrored in NumPy): bit-identical candidates to the reference's
    per-column RandomState.uniform draws from any starting position (even/odd, fresh, mid-block), and the state
    handed back continues the stream."""
    from helpers import mt19937_device_mirror

    lo = np.linspace(-3.7, 2.0, d)
    hi = lo + np.linspace(0.5, 11.3, d)
    ref = np.random.RandomState(42)
    if burn:
        ref.randint(0, 2**31 - 1, size=burn)         # one 32-bit output each: odd burn -> odd position
    _, key, pos, hg, cg = ref.get_state()
    Xc, key2, pos2 = mt19937_device_mirror(key, pos, M, d, lo, hi)
    want = np.column_stack([ref.uniform(lo[t], hi[t], M) for t in range(d)])
    assert np.array_equal(Xc, want)
    cont = np.random.RandomState(0)
    cont.set_state(("MT19937", key2, pos2, hg, cg))
    assert np.array_equal(cont.uniform(size=700), ref.uniform(size=700))


def test_lbfgsb_driver_equals_scipy_minimize():
    """lbfgsb_lockstep.minimize_many drives SciPy's setulb for all starts at once over a batched objective; every run
    ends in the OptimizeResult `minimize(f, x0, bounds=..., method="L-BFGS-B")` (no jac: the reference's call,
    bayes_opt/acquisition.py:366) returns — x, fun, jac, nit, nfev, status — bit for bit, including starts on the bounds,
    active bounds at the optimum, half-open and unbounded boxes."""
    from scipy.optimize import minimize

    from bayesianoptimization_amd import lbfgsb_lockstep as LL

    assert LL.driver_available()          # scipy 1.15.x in this image; elsewhere the threaded path is used
    rng = np.random.RandomState(11)
    A = rng.randn(6, 6)
    A = A @ A.T + np.eye(6)
    b = rng.randn(6)

    def f1(x):
        return 0.5 * x @ A @ x - x @ b + np.sin(3 * x).sum()

    batches = []

    def acq(P):
        batches.append(len(P))
        return np.array([f1(r) for r in P])

    boxes = [np.array([[-1.0, 1.0]] * 6), np.array([[-0.2, 0.3]] * 6),
             np.array([[-np.inf, 0.5], [-1.0, np.inf], [-np.inf, np.inf], [-1, 1], [0.0, np.inf], [-np.inf, 0.0]])]
    for box in boxes:
        lo = np.where(np.isinf(box[:, 0]), -2.0, box[:, 0])
        hi = np.where(np.isinf(box[:, 1]), 2.0, box[:, 1])
        starts = np.vstack([rng.uniform(lo, hi, size=(7, 6)), lo[None], hi[None], np.zeros((1, 6))])
        batches.clear()
        got = LL.minimize_many(acq, starts, box)
        for x0, r in zip(starts, got):
            ref = minimize(f1, x0, bounds=box, method="L-BFGS-B")
            assert np.array_equal(r.x, ref.x) and r.fun == ref.fun and np.array_equal(r.jac, ref.jac)
            assert (r.nit, r.nfev, r.status, r.success) == (ref.nit, ref.nfev, ref.status, ref.success)
        assert len(batches) == max(r.nfev for r in got) // 7 and batches[0] == len(starts) * 7


def test_forward_difference_points_match_the_per_seed_closure():
    from bayesianoptimization_amd import lbfgsb_lockstep as LL
    from bayesianoptimization_amd.fused_acquisition import _fd_value_and_grad

    rng = np.random.RandomState(12)
    box = np.array([[0.0, 1.0], [-3.0, 3.0], [2.0, 4.0], [-np.inf, np.inf]])
    X0 = np.vstack([rng.uniform([0, -3, 2, -5], [1, 3, 4, 5], size=(6, 4)), [0.0, -3.0, 4.0, 0.0], [1.0, 3.0, 2.0, 1e9]])
    pts, steps = LL.forward_difference_points(X0, box[:, 0], box[:, 1])
    for x0, P, h in zip(X0, pts, steps):
        seen = []
        _fd_value_and_grad(lambda q: (seen.append(q.copy()), np.zeros(len(q)))[1], box)(x0)
        assert np.array_equal(seen[0], P) and np.array_equal(h, P[1:][np.arange(4), np.arange(4)] - x0)


def test_lbfgsb_driver_with_gradient_equals_scipy_minimize():
    """minimize_many_with_grad == scipy.optimize.minimize(fg, x0, jac=True, bounds=..., method="L-BFGS-B") per start (the
    call of sklearn's theta search, _gpr.py:656-668): x, fun, jac, nit, nfev, status, message bit for bit."""
    from scipy.optimize import minimize

    from bayesianoptimization_amd import lbfgsb_lockstep as LL

    rng = np.random.RandomState(13)
    A = rng.randn(4, 4)
    A = A @ A.T + np.eye(4)
    b = rng.randn(4)

    def fg(x):
        return 0.5 * x @ A @ x - x @ b + np.cos(2 * x).sum(), A @ x - b - 2 * np.sin(2 * x)

    def batch(X):
        out = [fg(x) for x in X]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out])

    box = np.array([[-11.5, 11.5], [-1.0, 0.2], [0.0, 3.0], [-0.5, 0.5]])
    starts = rng.uniform(box[:, 0], box[:, 1], size=(6, 4))
    for x0, r in zip(starts, LL.minimize_many_with_grad(batch, starts, box)):
        ref = minimize(fg, x0, jac=True, bounds=box, method="L-BFGS-B")
        assert np.array_equal(r.x, ref.x) and r.fun == ref.fun and np.array_equal(r.jac, ref.jac)
        assert (r.nit, r.nfev, r.status, r.success, r.message) == (ref.nit, ref.nfev, ref.status, ref.success, ref.message)


def test_engine_is_shared_by_copies_and_not_picklable():
    """sklearn.base.clone / copy.deepcopy of an estimator (bayes_opt's ConstantLiar deep-copies a constrained target
    space, constraint GPs included) must not duplicate the device context behind it."""
    import copy
    import pickle

    from sklearn.base import clone
    from sklearn.gaussian_process.kernels import Matern

    from bayesianoptimization_amd.engine import GpEngine
    from bayesianoptimization_amd.gpr import HipGPR

    class Handleless(GpEngine):            # no GPU here: skip context creation, keep the copy semantics
        def __init__(self):
            self._h = None
            self._serial = {}

    eng = Handleless()
    assert copy.deepcopy(eng) is eng and copy.copy(eng) is eng
    gp = HipGPR(kernel=Matern(nu=2.5), engine=eng, slot=3, precision="f32", incremental=False, theta_lockstep=False)
    twin = clone(gp)
    assert twin.engine is eng and twin.slot == 3 and twin.precision == "f32"
    assert twin.incremental is False and twin.theta_lockstep is False
    assert copy.deepcopy(gp).engine is eng
    with pytest.raises(TypeError):
        pickle.dumps(eng)


def test_blas_single_thread_is_reentrant_and_thread_safe():
    """the limit the lockstep drivers put on the host BLAS pools is set by the first user and lifted by the last —
    interleaved enters / exits from several threads (two optimizers, one context each) must leave the pools as they were."""
    import threading

    from threadpoolctl import threadpool_info

    from bayesianoptimization_amd import lbfgsb_lockstep as D

    def blas_threads():
        return [m["num_threads"] for m in threadpool_info() if m.get("user_api") == "blas"]

    before = blas_threads()
    with D.blas_single_thread():
        with D.blas_single_thread():
            assert all(n == 1 for n in blas_threads())
        assert all(n == 1 for n in blas_threads())          # the inner exit must not lift the outer limit
    assert blas_threads() == before
    # A enters, B enters, A leaves, B leaves (the order that used to leave the pools at 1)
    a_in, b_in, a_out = threading.Event(), threading.Event(), threading.Event()

    def a():
        with D.blas_single_thread():
            a_in.set()
            b_in.wait(10)
        a_out.set()

    def b():
        a_in.wait(10)
        with D.blas_single_thread():
            b_in.set()
            a_out.wait(10)
            assert all(n == 1 for n in blas_threads())

    ts = [threading.Thread(target=f) for f in (a, b)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(20)
    assert blas_threads() == before and D._BLAS_USERS == 0


def test_bare_length_scale_kernels_are_read_and_written_as_sklearn_does():
    """HipGPR.fit reads theta / bounds of a bare Matern / RBF from its attributes and writes the optimum back the same way
    (gpr._length_scale_theta / _set_length_scale_theta) instead of through Kernel.theta / .bounds — which walk dir() and
    inspect.signature on every access.  Same arrays, same attribute types as sklearn's own accessors (kernels.py:285-343)."""
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Hyperparameter, Matern

    from bayesianoptimization_amd import gpr as G

    class Wrapped(Matern):          # what bayes_opt.parameter.wrap_kernel makes: a dynamic subclass, same hyper-parameters
        pass

    class Extra(RBF):               # a subclass with a second hyper-parameter is NOT bare
        @property
        def hyperparameter_other(self):
            return Hyperparameter("other", "numeric", (1e-2, 1e2))

    for k in (Matern(nu=2.5), RBF(0.7), Matern(length_scale=[0.5, 2.0, 1.0], nu=2.5), RBF(length_scale=[0.3, 0.4], length_scale_bounds=(1e-3, 1e2)),
              Matern(length_scale=[1.5], nu=2.5), Wrapped(length_scale=0.2, nu=2.5),
              RBF(length_scale=[0.3, 0.4], length_scale_bounds=[(1e-3, 1e2), (1e-1, 1e1)])):
        assert G._bare_length_scale_kernel(k)
        theta, bounds = G._length_scale_theta(k)
        assert np.array_equal(theta, k.theta) and np.array_equal(bounds, k.bounds)
        new = theta + np.linspace(0.1, 0.3, theta.shape[0])
        ref = k.clone_with_theta(new)
        G._set_length_scale_theta(k, new)
        assert type(k.length_scale) is type(ref.length_scale) and np.array_equal(k.length_scale, ref.length_scale)
        assert np.array_equal(k.theta, ref.theta)
    for k in (RBF(1.0, length_scale_bounds="fixed"), ConstantKernel(1.0, "fixed") * RBF(1.0), Extra(1.0)):
        assert not G._bare_length_scale_kernel(k)


def test_an_optimum_on_a_bound_still_warns_as_sklearn_does():
    import warnings

    from sklearn.exceptions import ConvergenceWarning
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF

    import helpers as H
    from bayesianoptimization_amd.gpr import HipGPR

    rng = np.random.RandomState(0)
    X = rng.uniform(size=(40, 1))
    y = np.sin(25 * X[:, 0]) + 0.05 * rng.standard_normal(40)      # wiggly: the search runs into the lower bound 0.9
    eng = H.FakeEngine()
    gp = HipGPR(kernel=RBF(1.0, length_scale_bounds=(0.9, 1.1)), alpha=1e-6, normalize_y=True, n_restarts_optimizer=2,
                random_state=np.random.RandomState(1), engine=eng)
    sk = GaussianProcessRegressor(kernel=RBF(1.0, length_scale_bounds=(0.9, 1.1)), alpha=1e-6, normalize_y=True, n_restarts_optimizer=2,
                                  random_state=np.random.RandomState(1))
    msgs = []
    for est in (gp, sk):
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            est.fit(X, y)
        msgs.append(sorted(str(x.message) for x in w if issubclass(x.category, ConvergenceWarning) and "close to the specified" in str(x.message)))
    assert msgs[0] == msgs[1] and len(msgs[1]) >= 1
    assert gp.kernel_.length_scale == pytest.approx(sk.kernel_.length_scale, rel=1e-6)
