"""GPU (-m gpu): the fp32 posterior mode (precision = GPBO_F32: fp64 factorisation, k* and W rounded to fp32,
contraction on v_mfma_f32_16x16x4_f32).  The reference has no fp32 path (float64 throughout,
bayes_opt/target_space.py:95-96), so this mode is checked against the SAME fp64 goldens/oracle with an fp32
tolerance stated here: mu keeps fp64 accuracy (its dot product is accumulated in fp64 before rounding); the
VARIANCE carries the absolute error of an fp32 sum of squares, |sigma^2 - sigma_ref^2| <= 2e-5 * y_std^2 (measured 5e-6 at C3/C5)
(so sigma itself is only resolved down to ~2e-3 * y_std: the cancellation 1 - |W k*|^2 cannot be repaired after
the fact); the acquisition within 1e-4 of its range (10x the 2e-6 .. 9e-6 the full C5 shards measure,
profiles/r03_f32_shards.json — the bound tests/test_gpu_sharded.py asserts; until round 4 this file asserted a loose 5e-3
and a conditional arg-best); the arg-best index equals the fp64 reference's, unconditionally: every golden's top-2 gap is
> 6 % of the range, hundreds of times the error bound (asserted, so that the claim cannot silently turn into a coin flip)."""
import numpy as np
import pytest

from bayesianoptimization_amd import workloads as W
from bayesianoptimization_amd.engine import F32
from conftest import load_golden, rel_err
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

F32_ACQ_TOL = 1e-4      # of the acquisition range (tests/test_gpu_sharded.py uses the same bound on all 8 C5 shards)


@pytest.mark.parametrize("N,d,M,kernel,ls", [(200, 3, 4096, O.MATERN25, 0.4), (600, 8, 5000, O.MATERN25, 1.0),
                                              (257, 5, 1000, O.RBF, 0.9), (1100, 16, 3000, O.MATERN25, 1.5)])
def test_f32_posterior_against_oracle(engine, N, d, M, kernel, ls):
    rng = np.random.RandomState(31)
    X = rng.uniform(size=(N, d))
    y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(N)
    gp = O.fit_fixed_theta(kernel, X, y, ls, 1e-6)
    yn, ym, ys = O.normalize_targets(y)
    engine.fit(X, yn, kernel, ls, 1e-6, precision=F32)
    assert rel_err(engine.get_L(N), gp.L) < 1e-10            # the factorisation is still fp64
    Xc = rng.uniform(size=(M, d))
    mu, sd = engine.predict(Xc, y_mean=ym, y_std=ys)
    mu_o, sd_o = O.predict(gp, Xc)
    assert rel_err(mu, mu_o) < 1e-7
    assert np.max(np.abs(sd**2 - sd_o**2)) < 2e-5 * ys**2
    # and the fp64 mode on the same context is untouched
    engine.fit(X, yn, kernel, ls, 1e-6)
    mu64, sd64 = engine.predict(Xc, y_mean=ym, y_std=ys)
    assert rel_err(sd64, sd_o) < 1e-7


@pytest.mark.parametrize("name", ["C2", "C5S", "C5"])
def test_f32_against_reference_goldens(engine, name):
    w = W.ALL[name]
    g = load_golden(name)
    X, y, c = W.make_observations(w)
    yn, ym, ys_ = O.normalize_targets(y)
    engine.fit(X, yn, w.kernel, g["length_scale"], w.noise, slot=0, precision=F32)
    M = int(g["M_evaluated"])
    Xc = W.make_candidates(w.bounds_array(), M, 7)
    engine.set_candidates(Xc)
    mu, sd = engine.posterior(0, ym, ys_)
    S = len(g["mu"])
    assert rel_err(mu[:S], g["mu"]) < 1e-7
    assert np.max(np.abs(sd[:S] ** 2 - g["sd"] ** 2)) < 2e-5 * ys_**2
    lb = ub = None
    if w.constrained:
        cn, cm, cs = O.normalize_targets(c)
        engine.fit(X, cn, W.MATERN25, g["c_length_scale"], w.noise, slot=1, precision=F32)
        cmu, csd = engine.posterior(1, cm, cs)
        assert np.max(np.abs(csd[:S] ** 2 - g["c_sd"] ** 2)) < 2e-5 * cs**2
        lb, ub = [-np.inf], [w.constraint_ub]
    y_max = W.feasible_y_max(w, y, c)
    bi, bv, si, sv, ys = engine.acq_argbest(w.acq, w.acq_param, y_max, lb, ub, k_seeds=4, return_values=True)
    rng_ = np.max(g["ys"]) - np.min(g["ys"])
    e = F32_ACQ_TOL * rng_
    assert np.max(np.abs(ys[:S] - g["ys"])) <= e
    gap = float(g["topk_val"][1] - g["topk_val"][0])
    assert gap > 2 * e                          # C2: 6.7 %, C5S: 8 %, C5: 40 % of |min| against a bound of 0.02 %
    assert bi == int(g["argmin"])
    assert abs(bv - float(g["min"])) <= e
    # the four best in the reference's order wherever its values stand clear of each other by more than twice the bound
    ref_idx, ref_val = g["topk_idx"].astype(np.int64), g["topk_val"]
    for p in range(4):
        if (p == 0 or ref_val[p] - ref_val[p - 1] > 2 * e) and ref_val[p + 1] - ref_val[p] > 2 * e:
            assert si[p] == ref_idx[p]
