"""CPU: the analytic local-search gradient.  The oracle's `predict_grad` against central differences
of its own `predict`; the acquisition chain rule of fused_acquisition (`_value_and_grad`: UCB / EI / POI, constraint
probabilities) against central differences of the `_get_acq` objective it differentiates; a whole suggest() with
`analytic_gradient=True` ends at an acquisition value at least as good as the finite-difference search's."""
import warnings

import numpy as np
import pytest
from sklearn.gaussian_process.kernels import Matern

from bayesianoptimization_amd import fused_acquisition as A
from bayesianoptimization_amd import workloads as W
from bayesianoptimization_amd.constraint_model import HipConstraintModel
from bayesianoptimization_amd.float_space import FloatSpace
from bayesianoptimization_amd.gpr import HipGPR
from helpers import FakeEngine
from oracle import gp_oracle as O


def _central(fun, X, h=1e-6):
    X = np.asarray(X, dtype=np.float64)
    g = np.empty(X.shape)
    for t in range(X.shape[1]):
        e = np.zeros(X.shape[1]); e[t] = h
        g[:, t] = (fun(X + e) - fun(X - e)) / (2 * h)
    return g


@pytest.mark.parametrize("kind,ls", [(O.MATERN25, 0.7), (O.RBF, 0.5), (O.MATERN25, [0.4, 0.8, 1.3])])
def test_oracle_predict_grad_matches_central_differences(kind, ls):
    rng = np.random.RandomState(2)
    X = rng.uniform(size=(80, 3))
    y = np.sin(3 * X.sum(1)) + 0.05 * rng.randn(80)
    gp = O.fit_fixed_theta(kind, X, y, ls, 1e-6)
    Xq = rng.uniform(size=(9, 3))
    mu, sd, dmu, dsd = O.predict_grad(gp, Xq)
    m0, s0 = O.predict(gp, Xq)
    assert np.array_equal(mu, m0) and np.array_equal(sd, s0)
    # (central differences of an ill-conditioned GP — alpha ~ 1e5 at noise 1e-6 — carry ~1e-5 of cancellation noise)
    g_mu, g_sd = _central(lambda Z: O.predict(gp, Z)[0], Xq), _central(lambda Z: O.predict(gp, Z)[1], Xq)
    assert np.max(np.abs(dmu - g_mu)) < 3e-5 * np.max(np.abs(g_mu))
    assert np.max(np.abs(dsd - g_sd)) < 3e-4 * np.max(np.abs(g_sd))


def _setup(constrained, policy):
    w = W.C5S
    X, y, c = W.make_observations(w)
    eng = FakeEngine()
    cons = None
    if constrained:
        cons = HipConstraintModel(None, -0.3, w.constraint_ub, engine=eng)
        cons._model[0].set_params(kernel=Matern(nu=2.5, length_scale=0.7), optimizer=None)
    sp = FloatSpace(w.pbounds(), constraint=cons)
    sp.register_bulk(X, y, c if constrained else None)
    gp = HipGPR(kernel=Matern(nu=2.5, length_scale=0.5), alpha=1e-6, normalize_y=True, optimizer=None, engine=eng)
    fn = {"ucb": lambda: A.UpperConfidenceBound(kappa=2.0), "ei": lambda: A.ExpectedImprovement(xi=0.01),
          "poi": lambda: A.ProbabilityOfImprovement(xi=0.01)}[policy]()
    return w, sp, gp, fn


@pytest.mark.parametrize("policy,constrained", [("ucb", False), ("ei", False), ("poi", False), ("ei", True), ("poi", True)])
def test_acquisition_chain_rule_matches_central_differences(policy, constrained):
    w, sp, gp, fn = _setup(constrained, policy)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fn._fit_gp(gp, sp)
    if policy != "ucb":
        fn.y_max = sp._target_max()
    chain = A._fused_models(gp, sp.constraint)
    assert chain is not None
    fg = fn._value_and_grad(chain, sp.constraint)
    acq = fn._get_acq(gp, sp.constraint)
    Xq = np.random.RandomState(3).uniform(0.05, 0.95, size=(12, w.d))
    f, g = fg(Xq)
    assert np.allclose(f, acq(Xq), rtol=1e-12, atol=1e-15)
    gfd = _central(acq, Xq)
    scale = np.max(np.abs(gfd)) + 1e-12
    assert np.max(np.abs(g - gfd)) < 2e-5 * scale


@pytest.mark.parametrize("policy,constrained", [("ucb", False), ("ei", True)])
def test_suggest_with_analytic_gradient_is_at_least_as_good(policy, constrained):
    if constrained and policy == "ucb":
        pytest.skip("UCB takes no constraints")
    vals = {}
    for analytic in (False, True):
        w, sp, gp, fn = _setup(constrained, policy)
        fn.analytic_gradient = analytic
        fn.device_polish = False          # this test is about the Python driver over gpbo_predict_grad (the one-call stage has its own)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x = fn.suggest(gp, sp, n_random=3000, n_smart=6, fit_gp=True, random_state=np.random.RandomState(11))
            if policy != "ucb":
                fn.y_max = sp._target_max()
            vals[analytic] = float(fn._get_acq(gp, sp.constraint)(x[None])[0])
        assert np.all(x >= sp.bounds[:, 0]) and np.all(x <= sp.bounds[:, 1])
        if analytic:
            assert any(c[0] == "predict_grad" for c in gp.engine.calls)
    # negated acquisition: smaller is better; the exact gradient must not end noticeably worse than finite differences
    assert vals[True] <= vals[False] + 1e-6 * max(1.0, abs(vals[False]))
