"""TEST INFRASTRUCTURE — the SciPy side of the local-search sweep, computed once on the CPU and committed as a fixture.

    python -m oracle.gen_polish_sweep          ->  tests/golden/polish_sweep.npz  (+ an entry in MANIFEST.json)

What the reference does in its local-search stage (bayes_opt/acquisition.py:322-420, all-float space): from each of the
n_smart best random candidates, `scipy.optimize.minimize(acq, x_seed, bounds=..., method="L-BFGS-B")` with SciPy's forward
differences, on  acq(x) = -base_acq(mu(x), sd(x)) [* P(lb <= c(x) <= ub)]  (acquisition.py:198-217, constraint.py:199-221).
`gpbo_polish_seeds` replaces that stage with another optimiser, so its parity is statistical and needs a
sample worth the name: 66 problems — UCB / EI / POI x unconstrained / constrained (UCB takes no constraints) x
d in {2, 8, 16, 32} x N in {60, 512, 2048}, plus six RBF ones — x 10 seeds each.  SciPy + the NumPy oracle are deterministic
CPU code, identical here and on the GPU box, so their half of the comparison is a fixture: per problem the 10 seeds (the best
of 4000 random candidates under the oracle's acquisition, as `_random_sample_minimize` picks them, acquisition.py:311-317)
and SciPy's end point / value / success / evaluation count from each.  tests/test_gpu_polish.py runs the device stage from
the same seeds and evaluates its end points with the same oracle.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
from scipy.optimize import minimize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import gp_oracle as O  # noqa: E402

N_SEEDS = 10
N_CAND = 4000
OUT = os.path.join(ROOT, "tests", "golden", "polish_sweep.npz")


def problems():
    """[(key, acq, param, constrained, two_sided, kernel, d, N)] — the sweep; the key names the fixture's arrays."""
    out = []
    for acq, name, param in ((O.UCB, "ucb", 2.576), (O.EI, "ei", 0.01), (O.POI, "poi", 0.01)):
        for constrained in (False, True):
            if constrained and acq == O.UCB:
                continue            # ConstraintNotSupportedError in the reference (acquisition.py:524-529)
            for d in (2, 8, 16, 32):
                for N in (60, 512, 2048):
                    out.append((f"{name}_{'con' if constrained else 'unc'}_d{d}_N{N}", acq, param, constrained,
                                constrained and acq == O.POI, O.MATERN25, d, N))
    for acq, name, param in ((O.UCB, "ucb", 2.576), (O.EI, "ei", 0.01)):
        for d, N in ((2, 60), (8, 512), (16, 512)):
            out.append((f"{name}_unc_rbf_d{d}_N{N}", acq, param, False, False, O.RBF, d, N))
    return out


def build(key, acq, param, constrained, two_sided, kernel, d, N):
    """Inputs, the oracle's models and the objective of one problem (shared by the generator and the GPU test)."""
    seed = sum(ord(ch) * (i + 1) for i, ch in enumerate(key)) % (2**31 - 1)
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(N, d))
    s = X[:, : min(d, 4)].sum(1)
    y = np.sin(3.0 * s) + 0.05 * rng.standard_normal(N)
    ls = 0.3 * np.sqrt(d) if kernel == O.MATERN25 else 0.22 * np.sqrt(d)
    gp = O.fit_fixed_theta(kernel, X, y, ls, 1e-6)
    cons, c, lb, ub, cls = None, None, None, None, 0.35 * np.sqrt(d)
    y_max = float(np.max(y))
    if constrained:
        c = np.cos(2.0 * s) + 0.02 * rng.standard_normal(N)
        lb, ub = ([-0.6], [0.5]) if two_sided else ([-np.inf], [0.5])
        cgp = O.fit_fixed_theta(O.MATERN25, X, c, cls, 1e-6)
        cons = ([cgp], lb, ub)
        ok = (c >= lb[0]) & (c <= ub[0])
        y_max = float(np.max(y[ok]))
    cand = rng.uniform(size=(N_CAND, d))
    vals = O.neg_acquisition(gp, cand, acq, param, y_max, cons)
    seeds = cand[np.argsort(vals, kind="stable")[:N_SEEDS]]
    box = np.array([[0.0, 1.0]] * d)

    def f_batch(P):
        return O.neg_acquisition(gp, np.atleast_2d(P), acq, param, y_max, cons)

    return {"X": X, "y": y, "c": c, "ls": ls, "cls": cls, "lb": lb, "ub": ub, "y_max": y_max, "seeds": seeds, "box": box,
            "f_batch": f_batch, "kernel": kernel}


def _fd(f_batch, box):
    """value-and-gradient by forward differences (h = 1e-8, one-sided at the upper bound) — the scheme SciPy's L-BFGS-B uses
    when no `jac` is given (scipy/optimize/_numdiff.py, '2-point' with bounds), in ONE batched oracle call per evaluation."""
    lo, hi = box[:, 0], box[:, 1]

    def fun(x):
        d = x.shape[0]
        h = np.full(d, 1e-8)
        h = np.where(x + h > hi, -h, h)
        P = np.repeat(x[None, :], d + 1, axis=0)
        P[np.arange(1, d + 1), np.arange(d)] += h
        v = f_batch(P)
        steps = P[np.arange(1, d + 1), np.arange(d)] - x
        return float(v[0]), (v[1:] - v[0]) / steps

    return fun


def generate(verbose=True):
    arrays = {}
    t_all = time.time()
    for spec in problems():
        key = spec[0]
        t0 = time.time()
        p = build(*spec)
        fun = _fd(p["f_batch"], p["box"])
        xs, fs, ok, nfev, nit = [], [], [], [], []
        for s0 in p["seeds"]:
            r = minimize(fun, s0, jac=True, bounds=p["box"], method="L-BFGS-B")
            xs.append(np.clip(r.x, 0.0, 1.0)); fs.append(float(r.fun)); ok.append(bool(r.success)); nfev.append(int(r.nfev)); nit.append(int(r.nit))
        arrays[f"{key}__seeds"] = p["seeds"]
        arrays[f"{key}__scipy_x"] = np.array(xs)
        arrays[f"{key}__scipy_f"] = np.array(fs)
        arrays[f"{key}__scipy_ok"] = np.array(ok)
        arrays[f"{key}__scipy_nfev"] = np.array(nfev)
        arrays[f"{key}__scipy_nit"] = np.array(nit)
        arrays[f"{key}__f_seeds"] = p["f_batch"](p["seeds"])
        if verbose:
            print(f"{key:28s} best {min(fs):+.6e}  ok {sum(ok)}/10  nfev {int(np.mean(nfev))}  {time.time() - t0:.1f}s", flush=True)
    np.savez_compressed(OUT, **arrays)
    man_path = os.path.join(ROOT, "tests", "golden", "MANIFEST.json")
    man = json.load(open(man_path))
    import scipy
    man["polish_sweep"] = {"file": "polish_sweep.npz", "generator": "python -m oracle.gen_polish_sweep",
                           "what": "SciPy L-BFGS-B (forward differences) from the 10 best of 4000 random candidates on the oracle's "
                                   "acquisition, 66 problems", "scipy": scipy.__version__, "numpy": np.__version__,
                           "problems": len(problems()), "seconds": round(time.time() - t_all, 1)}
    json.dump(man, open(man_path, "w"), indent=1)
    return arrays


if __name__ == "__main__":
    generate()
