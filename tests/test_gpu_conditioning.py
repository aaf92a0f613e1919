"""GPU (-m gpu): parity where K is ILL-CONDITIONED, asserted ELEMENTWISE (north_star: "within 1e-5 relative").

Every other parity test of the suite uses the max-norm `rel_err` (tests/conftest.py) on problems whose kernel matrix has a
condition number <= 5e6.  The device does not do what the reference does: it forms W = L^-1 once per fit and multiplies
(V = W K*^T), the reference back-substitutes per candidate (`solve_triangular`, sklearn _gpr.py:454-456, and
`cho_solve` for alpha, :360-364).  The two agree to kappa(K) * eps, so this file runs them where kappa(K) = 1e8 .. 2e9 — the
regime 2-D problems with a few hundred points, RBF kernels and `allow_duplicate_points=True`
(/root/reference/bayes_opt/target_space.py:424-518) put real users in:

  rbf_d2_N1500      RBF, d = 2, N = 1500, length scale 0.5                                   kappa ~ 9e8
  matern_d2_N2000   Matern-2.5, d = 2, N = 2000, length scale 1.0                            kappa ~ 1.6e9
  c1_space_N300     the README's space (2,4) x (-3,3) with RBF(1.0), N = 300                 kappa ~ 9e7
  dups_matern_N600  20 duplicated rows at N = 600, Matern-2.5(0.3)                           kappa ~ N / alpha
  dups_rbf_N600     the same with RBF(0.3)

each with M = 20 001 candidates of which the first 200 lie within 1e-4 of training points (sigma -> sqrt(alpha): the
cancellation case 1 - sum v^2), through the three product routes:

  gpbo_set_candidates + gpbo_posterior + gpbo_acq_argbest   (large batch: fused / slab MFMA kernels)
  gpbo_predict on 40 points                                 (small batch: batched-GEMV kernels)
  gpbo_predict_grad on 40 points                            (mu, sigma and their input gradients)

Asserted, fp64:  |sigma - sigma_ref| <= 1e-5 sigma_ref for EVERY candidate with sigma_ref > 0;  |mu - mu_ref| <= 1e-5 max(|mu_ref|,
s_y) for every candidate; arg-best and the 16 best indices equal to the oracle's wherever the oracle's neighbouring values
are further apart than twice the error bound those tolerances imply for the acquisition.  The measured errors per case go to
gpurun_out/r05_conditioning.json (-> profiles/).  The oracle is the checker (oracle/gp_oracle.py, pinned to the reference)."""
import json
import os

import numpy as np
import pytest

from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

M = 20001
N_NEAR = 200
K_SEEDS = 16
TOL = 1e-5
_report = {}


def _make(name):
    rng = np.random.RandomState(11 + CASES.index(name))
    if name == "rbf_d2_N1500":
        kernel, ls, N, lo, hi = O.RBF, 0.5, 1500, np.zeros(2), np.ones(2)
    elif name == "matern_d2_N2000":
        kernel, ls, N, lo, hi = O.MATERN25, 1.0, 2000, np.zeros(2), np.ones(2)
    elif name == "c1_space_N300":
        kernel, ls, N, lo, hi = O.RBF, 1.0, 300, np.array([2.0, -3.0]), np.array([4.0, 3.0])
    elif name == "dups_matern_N600":
        kernel, ls, N, lo, hi = O.MATERN25, 0.3, 600, np.zeros(2), np.ones(2)
    else:
        kernel, ls, N, lo, hi = O.RBF, 0.3, 600, np.zeros(2), np.ones(2)
    X = lo + (hi - lo) * rng.uniform(size=(N, 2))
    if name.startswith("dups"):
        X[N - 20:] = X[:20]                                  # allow_duplicate_points=True: exact duplicates
    if name == "c1_space_N300":
        y = -X[:, 0] ** 2 - (X[:, 1] - 1.0) ** 2 + 1.0     # the README's black_box_function
    else:
        y = np.sin(3.0 * X.sum(1)) + 0.1 * rng.standard_normal(N)
        if name.startswith("dups"):
            y[N - 20:] = y[:20]
    Xc = lo + (hi - lo) * rng.uniform(size=(M, 2))
    Xc[:N_NEAR] = X[:N_NEAR] + 1e-4 * (hi - lo) * rng.standard_normal((N_NEAR, 2))
    Xc = np.clip(Xc, lo, hi)
    return kernel, ls, X, y, Xc


def _elementwise(mu, sd, mu_o, sd_o, s_y):
    pos = sd_o > 0
    e_sd = float(np.max(np.abs(sd - sd_o)[pos] / sd_o[pos])) if pos.any() else 0.0
    e_mu = float(np.max(np.abs(mu - mu_o) / np.maximum(np.abs(mu_o), s_y)))
    return e_sd, e_mu


CASES = ["rbf_d2_N1500", "matern_d2_N2000", "c1_space_N300", "dups_matern_N600", "dups_rbf_N600"]


@pytest.mark.parametrize("name", CASES)
def test_ill_conditioned_posterior_elementwise(engine, name):
    kernel, ls, X, y, Xc = _make(name)
    gp = O.fit_fixed_theta(kernel, X, y, ls, 1e-6)
    dl = np.diag(gp.L)
    mu_o, sd_o = O.predict(gp, Xc)
    yn, ym, ys_ = O.normalize_targets(y)
    engine.fit(X, yn, kernel, ls, 1e-6)
    row = {"N": int(X.shape[0]), "kernel": "rbf" if kernel == O.RBF else "matern25", "length_scale": ls,
           "diagL_max_over_min": float(dl.max() / dl.min()), "sd_ref_min": float(sd_o.min()), "sd_ref_max": float(sd_o.max())}
    if X.shape[0] <= 600:
        K = O.kernel_matrix(kernel, X, None, ls)
        K[np.diag_indices_from(K)] += 1e-6
        row["kappa_K"] = float(np.linalg.cond(K))

    # ---- large batch: set_candidates + posterior + acquisition + selection
    engine.set_candidates(Xc)
    mu, sd = engine.posterior(0, ym, ys_)
    e_sd, e_mu = _elementwise(mu, sd, mu_o, sd_o, ys_)
    row["large_batch"] = {"sd_elementwise": e_sd, "mu_elementwise": e_mu,
                          "sd_maxnorm": float(np.max(np.abs(sd - sd_o)) / sd_o.max()),
                          "sd_near_training_points": float(np.max(np.abs(sd - sd_o)[:N_NEAR] / sd_o[:N_NEAR]))}
    assert e_sd <= TOL, (name, "sd", e_sd)
    assert e_mu <= TOL, (name, "mu", e_mu)
    y_max = float(y.max())
    for acq, param in ((O.UCB, 2.576), (O.EI, 0.01)):
        ys_o = -1 * O.base_acq(acq, mu_o, sd_o, param, y_max)
        bi, bv, si, sv, ys = engine.acq_argbest(acq, param, y_max, k_seeds=K_SEEDS, return_values=True)
        # what the posterior tolerances allow the acquisition to move by: UCB is linear in (mu, sd); EI is 1-Lipschitz in mu and
        # <= phi(0)-Lipschitz in sd
        bound = TOL * (np.maximum(np.abs(mu_o), ys_) + param * sd_o) if acq == O.UCB else TOL * (np.maximum(np.abs(mu_o), ys_) + 0.4 * sd_o)
        err = np.abs(ys - ys_o)
        row["large_batch"]["ucb_err_over_bound" if acq == O.UCB else "ei_err_over_bound"] = float(np.max(err / bound))
        assert np.all(err <= bound), (name, acq, float(np.max(err / bound)))
        order = np.argsort(ys_o, kind="stable")
        ref_val = ys_o[order[:K_SEEDS + 1]]
        e = float(np.max(bound[order[:K_SEEDS + 1]]))
        exact = 0
        for p in range(K_SEEDS):          # the oracle's index wherever its value stands clear of both neighbours by > 2 e
            lo_ok = p == 0 or ref_val[p] - ref_val[p - 1] > 2 * e
            hi_ok = ref_val[p + 1] - ref_val[p] > 2 * e
            if lo_ok and hi_ok:
                assert si[p] == order[p], (name, acq, p, si, order[:K_SEEDS])
                exact += 1
            else:
                assert abs(float(ys_o[si[p]]) - ref_val[p]) <= 2 * e
        if ref_val[1] - ref_val[0] > 2 * e:
            assert bi == int(order[0])
        assert abs(bv - float(ys_o[order[0]])) <= e
        row["large_batch"]["ucb_top16_positions_pinned" if acq == O.UCB else "ei_top16_positions_pinned"] = exact

    # ---- small batch (batched-GEMV kernels): 20 of the near-training-point candidates + 20 others
    pick = np.r_[0:20, N_NEAR:N_NEAR + 20]
    mu_s, sd_s = engine.predict(Xc[pick], 0, ym, ys_)
    e_sd, e_mu = _elementwise(mu_s, sd_s, mu_o[pick], sd_o[pick], ys_)
    row["small_batch"] = {"sd_elementwise": e_sd, "mu_elementwise": e_mu}
    assert e_sd <= TOL and e_mu <= TOL, (name, "small batch", e_sd, e_mu)

    # ---- gpbo_predict_grad: values elementwise as above; gradients against the oracle's closed form, relative to the largest
    # gradient component of the batch (a gradient has no elementwise scale of its own: components cross zero)
    mu_g, sd_g, dmu, dsd = engine.predict_grad(Xc[pick], 0, ym, ys_)
    _, _, dmu_o, dsd_o = O.predict_grad(gp, Xc[pick])
    e_sd, e_mu = _elementwise(mu_g, sd_g, mu_o[pick], sd_o[pick], ys_)
    g_mu = float(np.max(np.abs(dmu - dmu_o)) / np.max(np.abs(dmu_o)))
    g_sd = float(np.max(np.abs(dsd - dsd_o)) / np.max(np.abs(dsd_o)))
    row["predict_grad"] = {"sd_elementwise": e_sd, "mu_elementwise": e_mu, "dmu_maxnorm": g_mu, "dsd_maxnorm": g_sd}
    assert e_sd <= TOL and e_mu <= TOL, (name, "predict_grad values", e_sd, e_mu)
    assert g_mu <= TOL and g_sd <= 1e-4, (name, "predict_grad gradients", g_mu, g_sd)

    _report[name] = row
    if os.path.isdir("gpurun_out"):
        json.dump({"tolerance": TOL, "M": M, "near_training_points": N_NEAR, "cases": _report},
                  open("gpurun_out/r05_conditioning.json", "w"), indent=1)
