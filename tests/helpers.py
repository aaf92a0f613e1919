"""Test infrastructure: oracle-backed stand-ins and shared builders (never imported by the product)."""
import numpy as np

from bayesianoptimization_amd import workloads as W
from oracle import gp_oracle as O


def oracle_case(w: W.Workload, length_scale, M=None, c_length_scale=None):
    """Fit the oracle for workload `w` and evaluate the negated acquisition on the first M candidates."""
    X, y, c = W.make_observations(w)
    M = w.M if M is None else M
    Xc = W.make_candidates(w.bounds_array(), M, 7)
    gp = O.fit_fixed_theta(w.kernel, X, y, length_scale, w.noise)
    cons = None
    if w.constrained:
        cgp = O.fit_fixed_theta(W.MATERN25, X, c, c_length_scale, w.noise)
        cons = ([cgp], [-np.inf], [w.constraint_ub])
    y_max = W.feasible_y_max(w, y, c)
    return {"X": X, "y": y, "c": c, "Xc": Xc, "gp": gp, "cons": cons, "y_max": y_max}


class OracleEngine:
    """Implements the slice of GpEngine that ShardedAcquisition uses, on the CPU oracle (tests only)."""

    def __init__(self, gp, cons=None):
        self.gp, self.cons = gp, cons
        self.world_size, self.rank = 1, 0

    def set_candidates(self, Xc):
        self.Xc = np.asarray(Xc)
        self.n_candidates = self.Xc.shape[0]

    def acq_argbest(self, acq, param, y_max=0.0, lb=None, ub=None, k_seeds=0, index_offset=0, return_values=False):
        ys = O.neg_acquisition(self.gp, self.Xc, acq, param, y_max, self.cons)
        if np.isnan(ys).any():
            bi, bv = int(np.flatnonzero(np.isnan(ys))[0]), float("nan")
        else:
            bi, bv = int(ys.argmin()), float(ys.min())
        nan = np.isnan(ys)
        order = np.lexsort((np.arange(len(ys)), np.where(nan, np.inf, ys) + 0.0, nan))[:k_seeds]
        return bi + index_offset, bv, order + index_offset, ys[order], (ys if return_values else None)
