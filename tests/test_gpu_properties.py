"""GPU (-m gpu): size-independent properties of the hot path AT BASELINE.json's full sizes (C3: N = 4096, d = 16,
M = 2^20; C5 shape: N = 8192, d = 32, two GPs), where a CPU oracle pass would take minutes.  Each property holds for the
reference's arithmetic (sklearn _gpr.py:443-494, bayes_opt/acquisition.py:198-217, 311-317) by construction:

  determinism      two fits / two passes give identical bits (fixed reduction orders)
  locality         a candidate's mu/sigma do not depend on its position or on the other candidates (rows of x_tries are
                   evaluated independently) -> permuting rows permutes the outputs bitwise
  sharding         arg-best / top-k of the full pass == merge of G shard passes, at G = 8
  linearity        mu is linear in the targets, sigma does not depend on them (_gpr.py:444, 474-475)
  interpolation    at the training points mu reproduces y and sigma ~ sqrt(alpha) (the reference's own loose pin,
                   tests/test_bayesian_optimization.py:639-665: +-1e-3, sigma < 0.02)
  closure identity ys == -(mu + kappa * sd) op for op (acquisition.py:207, 485), argmin/min/argsort[:k] of that array
  sortedness       the k seeds are the k smallest ys in non-decreasing order, ties by index
  fp32 mode        differs from fp64 by at most the documented 2e-5 * s_y^2 in variance
"""
import numpy as np
import pytest

from bayesianoptimization_amd import workloads as W
from bayesianoptimization_amd.distributed import merge_best, shard_range
from bayesianoptimization_amd.engine import F32, F64
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3(engine):
    w = W.C3
    X, y, _ = W.make_observations(w)
    yn, ym, ys = O.normalize_targets(y)
    Xc = W.make_candidates(w.bounds_array(), w.M, 7)
    return dict(w=w, X=X, y=y, yn=yn, ym=ym, ys=ys, Xc=Xc)


def _pass(engine, c, Xc=None, precision=F64, k=16, offset=0):
    w = c["w"]
    engine.fit(c["X"], c["yn"], w.kernel, w.length_scale, w.noise, precision=precision)
    engine.set_candidates(c["Xc"] if Xc is None else Xc)
    mu, sd = engine.posterior(0, c["ym"], c["ys"])
    bi, bv, si, sv, ys = engine.acq_argbest(w.acq, w.acq_param, 0.0, k_seeds=k, index_offset=offset, return_values=True)
    return mu, sd, bi, bv, si, sv, ys


def test_c3_full_size_determinism_closure_and_sortedness(engine, c3):
    w = c3["w"]
    mu, sd, bi, bv, si, sv, ys = _pass(engine, c3)
    L1 = engine.get_L(w.N)
    mu2, sd2, bi2, bv2, si2, sv2, ys2 = _pass(engine, c3)
    assert np.array_equal(L1, engine.get_L(w.N))
    assert np.array_equal(mu, mu2) and np.array_equal(sd, sd2) and np.array_equal(ys, ys2)
    assert (bi, bv) == (bi2, bv2) and np.array_equal(si, si2)
    # the closure, op for op (UCB: -1 * (mean + kappa * std))
    assert np.array_equal(ys, -1 * (mu + w.acq_param * sd))
    assert bi == int(ys.argmin()) and bv == ys.min()
    order = np.lexsort((np.arange(len(ys)), ys))[:16]
    assert np.array_equal(si, order) and np.array_equal(sv, ys[order]) and np.all(np.diff(sv) >= 0)
    assert np.all(sd >= 0) and np.all(sd <= c3["ys"] * (1 + 1e-12)) and np.all(np.isfinite(mu))


def test_c3_full_size_locality_under_row_permutation(engine, c3):
    mu, sd, *_ = _pass(engine, c3)
    perm = np.random.RandomState(5).permutation(c3["w"].M)
    mu_p, sd_p, *_ = _pass(engine, c3, Xc=c3["Xc"][perm])
    assert np.array_equal(mu_p, mu[perm]) and np.array_equal(sd_p, sd[perm])
    # and a candidate evaluated in a batch of 3000 (another kernel schedule: one partial slab) keeps its bits
    sub = perm[:3000]
    mu_s, sd_s, *_ = _pass(engine, c3, Xc=c3["Xc"][sub])
    assert np.array_equal(mu_s, mu[sub]) and np.array_equal(sd_s, sd[sub])


def test_c3_full_size_sharding_equals_single_pass(engine, c3):
    w = c3["w"]
    _, _, bi, bv, si, sv, ys = _pass(engine, c3, k=10)
    G = 8
    parts = []
    for r in range(G):
        s, e = shard_range(w.M, G, r)
        _, _, a, b, cidx, cval, ys_r = _pass(engine, c3, Xc=c3["Xc"][s:e], k=10, offset=s)
        assert np.array_equal(ys_r, ys[s:e])
        parts.append((b, a, cval, cidx))
    m = merge_best([p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts], [p[3] for p in parts], 10)
    assert m[0] == bi and m[1] == bv and np.array_equal(m[2], si) and np.array_equal(m[3], sv)


def test_c3_full_size_linearity_in_targets(engine, c3):
    """mu(a*y1 + b*y2) = a*mu(y1) + b*mu(y2) and sigma unchanged — targets swapped with gpbo_fit_append(n_new = 0),
    i.e. the factorisation is the same object in all three passes."""
    w = c3["w"]
    rng = np.random.RandomState(9)
    y1, y2 = c3["yn"], rng.standard_normal(w.N)
    Xc = c3["Xc"][: 1 << 17]
    engine.fit(c3["X"], y1, w.kernel, w.length_scale, w.noise)
    engine.set_candidates(Xc)
    mu1, sd1 = engine.posterior(0, 0.0, 1.0)
    engine.fit_append(np.empty((0, w.d)), y2)
    mu2, sd2 = engine.posterior(0, 0.0, 1.0)
    engine.fit_append(np.empty((0, w.d)), 0.7 * y1 - 1.3 * y2)
    mu3, sd3 = engine.posterior(0, 0.0, 1.0)
    assert np.array_equal(sd1, sd2) and np.array_equal(sd1, sd3)
    scale = max(np.max(np.abs(mu1)), np.max(np.abs(mu2)))
    assert np.max(np.abs(mu3 - (0.7 * mu1 - 1.3 * mu2))) <= 1e-9 * scale


def test_c3_interpolation_at_training_points(engine, c3):
    w = c3["w"]
    engine.fit(c3["X"], c3["yn"], w.kernel, w.length_scale, w.noise)
    mu, sd = engine.predict(c3["X"], y_mean=c3["ym"], y_std=c3["ys"])
    assert np.max(np.abs(mu - c3["y"])) < 1e-3                 # the reference's own pin: +-1e-3
    assert np.all(sd < 0.02)                                   # ... and sigma < 0.02
    assert np.all(sd <= 1.5 * np.sqrt(w.noise) * c3["ys"] + 1e-9)


def test_c3_full_size_fp32_mode_within_documented_bound(engine, c3):
    mu, sd, bi, *_ = _pass(engine, c3)
    mu32, sd32, bi32, _, _, _, ys32 = _pass(engine, c3, precision=F32)
    assert np.max(np.abs(mu32 - mu)) <= 1e-6 * np.max(np.abs(mu))
    assert np.max(np.abs(sd32**2 - sd**2)) <= 2e-5 * c3["ys"] ** 2
    assert ys32[bi] <= ys32[bi32] + 1e-4 * abs(ys32[bi32])     # the fp64 winner is (nearly) the fp32 winner


def test_c5_shape_two_gps_determinism_and_constraint_weighting(engine):
    """C5 shape (N = 8192, d = 32, target + one constraint GP, EI), one 2^16-candidate shard in fp32 mode (C5's dtype):
    deterministic; ys == -EI(mu, sd) * P(c <= ub) op for op on the host from the device's mu/sd (constraint.py:199-221,
    acquisition.py:847-849); arg-best of that array."""
    from scipy.special import ndtr

    w = W.C5
    X, y, c = W.make_observations(w)
    yn, ym, ys_ = O.normalize_targets(y)
    cn, cm, cs = O.normalize_targets(c)
    Xc = W.make_candidates(w.bounds_array(), 1 << 16, 7)
    y_max = W.feasible_y_max(w, y, c)
    outs = []
    for _ in range(2):
        engine.fit(X, yn, w.kernel, w.length_scale, w.noise, slot=0, precision=F32)
        engine.fit(X, cn, W.MATERN25, w.constraint_length_scale, w.noise, slot=1, precision=F32)
        engine.set_candidates(Xc)
        mu, sd = engine.posterior(0, ym, ys_)
        cmu, csd = engine.posterior(1, cm, cs)
        bi, bv, si, sv, ys = engine.acq_argbest(w.acq, w.acq_param, y_max, [-np.inf], [w.constraint_ub], k_seeds=10,
                                                return_values=True)
        outs.append((mu, sd, cmu, csd, bi, bv, si, ys))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    mu, sd, cmu, csd, bi, bv, si, ys = outs[0]
    a = mu - y_max - w.acq_param
    z = a / sd
    ei = a * ndtr(z) + sd * np.exp(-(z**2) / 2.0) / np.sqrt(2.0 * np.pi)
    p = ndtr((w.constraint_ub - cmu) / csd)
    want = -1 * ei * p
    assert np.max(np.abs(ys - want)) <= 1e-12 * np.max(np.abs(want))
    assert bi == int(ys.argmin()) and np.array_equal(si, np.lexsort((np.arange(len(ys)), ys))[:10])
