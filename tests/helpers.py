"""Test infrastructure: oracle-backed stand-ins and shared builders (never imported by the product)."""
import contextlib
import os

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


class FakeEngine:
    """The full GpEngine surface HipGPR/acquisition use, computed by the CPU oracle.  TEST DOUBLE ONLY:
    lets the host glue (HipGPR, fused acquisition classes, accelerate) be exercised against the real
    bayes_opt in the CPU container, where no GPU exists."""

    def __init__(self):
        self.models = {}
        self.post = {}
        self.Xc = None
        self.n_candidates = 0
        self.calls = []

    def _touch(self, slot):
        self.serial = getattr(self, "serial", {})
        self.serial[slot] = self.serial.get(slot, 0) + 1
        return self.serial[slot]

    def fit_serial(self, slot=0):
        return getattr(self, "serial", {}).get(slot, 0)

    def fit(self, X, y_norm, kernel, length_scale, noise, slot=0, precision=0):
        self.calls.append(("fit", slot, X.shape))
        gp = O.fit_fixed_theta(kernel, X, y_norm, length_scale, noise, normalize_y=False)
        self.models[slot] = gp
        self.inputs = getattr(self, "inputs", {})
        self.inputs[slot] = (np.array(X), kernel, length_scale, noise)
        return self._touch(slot)

    def fit_append(self, x_new, y_norm, slot=0):
        self.calls.append(("fit_append", slot, x_new.shape))
        if slot not in self.models:
            raise RuntimeError("gpbo_fit_append: slot has no fitted model (call gpbo_fit first)")
        X0, kernel, length_scale, noise = self.inputs[slot]
        X = np.vstack([X0, x_new]) if x_new.shape[0] else X0
        self.models[slot] = O.fit_fixed_theta(kernel, X, y_norm, length_scale, noise, normalize_y=False)
        self.inputs[slot] = (X, kernel, length_scale, noise)
        return self._touch(slot)

    def lml(self, X, y_norm, kernel, length_scale, noise, eval_gradient=True, slot=0):
        self.calls.append(("lml", slot))
        self._touch(slot)
        self.models.pop(slot, None)   # like the device: the slot's fit is clobbered
        return O.log_marginal_likelihood(kernel, X, y_norm, length_scale, noise, eval_gradient)

    def lml_batch(self, X, y_norm, kernel, length_scales, noise, eval_gradient=True, reuse_inputs=False):
        self.calls.append(("lml_batch", len(length_scales)))
        return [O.log_marginal_likelihood(kernel, X, y_norm, ls, noise, eval_gradient) for ls in np.atleast_2d(length_scales)]

    def get_L(self, n, slot=0):
        return self.models[slot].L.copy()

    def get_alpha(self, n, slot=0):
        return self.models[slot].alpha.copy()

    def set_candidates(self, Xc):
        self.calls.append(("set_candidates", Xc.shape))
        self.Xc = np.asarray(Xc, dtype=np.float64)
        self.n_candidates = self.Xc.shape[0]
        self.post = {}

    def generate_candidates_like(self, M, lo, hi, random_state):
        """What the device's index-parity generator returns: the reference's per-column draws (the kernel's block
        walk itself is checked against NumPy in test_mt19937_block_walk_reproduces_randomstate_uniform)."""
        self.calls.append(("generate_candidates_like", M))
        self.Xc = np.column_stack([random_state.uniform(lo[t], hi[t], M) for t in range(len(lo))])
        self.n_candidates = M
        self.post = {}

    def get_candidate_rows(self, idx, d):
        idx = np.atleast_1d(np.asarray(idx, dtype=np.int64))
        return self.Xc[idx]

    def posterior(self, slot=0, y_mean=0.0, y_std=1.0, fetch=True):
        self.calls.append(("posterior", slot))
        mu, sd = O.predict(self.models[slot], self.Xc)
        mu, sd = y_std * mu + y_mean, sd * y_std
        self.negvar = getattr(self, "negvar", False) or bool(O.negative_variances(self.models[slot], self.Xc))
        self.post[slot] = (mu, sd)
        return (mu, sd) if fetch else (None, None)

    def predict(self, Xc, slot=0, y_mean=0.0, y_std=1.0):
        self.set_candidates(Xc)
        return self.posterior(slot, y_mean, y_std, True)

    def take_negative_variance_flag(self):
        seen, self.negvar = getattr(self, "negvar", False), False
        return seen

    def predict_cov(self, Xc, slot=0, y_mean=0.0, y_std=1.0):
        self.calls.append(("predict_cov", slot, np.shape(Xc)))
        mu, cov = O.predict_cov(self.models[slot], np.asarray(Xc, dtype=np.float64))
        return y_std * mu + y_mean, cov * y_std**2

    def predict_grad(self, Xc, slot=0, y_mean=0.0, y_std=1.0):
        self.calls.append(("predict_grad", slot, np.shape(Xc)))
        mu, sd, dmu, dsd = O.predict_grad(self.models[slot], np.asarray(Xc, dtype=np.float64))
        return y_std * mu + y_mean, sd * y_std, y_std * dmu, y_std * dsd

    def polish_seeds(self, acq, param, y_max, lb, ub, y_means, y_stds, seeds, box, max_iter=0):
        """The local-search stage as ONE engine call (GpEngine.polish_seeds / gpbo_polish_seeds), on the oracle: every seed's
        L-BFGS-B run by SciPy over the oracle's -acquisition [x constraint probability] — analytic gradient (O.predict_grad + the
        chain rule) without constraints, SciPy's finite differences with them.  Returns (x, f, status 0 | 2, rounds)."""
        import copy

        from scipy.optimize import minimize

        self.calls.append(("polish_seeds", acq, len(seeds)))
        gps = []
        for j in range(len(y_means)):
            g = copy.copy(self.models[j])
            g.y_mean, g.y_std = float(y_means[j]), float(y_stds[j])
            gps.append(g)
        ym = 0.0 if y_max is None else float(y_max)
        cons = (gps[1:], lb, ub) if len(gps) > 1 else None

        def f_only(x):
            return float(O.neg_acquisition(gps[0], x[None], acq, param, ym, cons)[0])

        def f_grad(x):
            mu, sd, dmu, dsd = (v[0] for v in O.predict_grad(gps[0], x[None]))
            if acq == O.UCB:
                a, ca, cs = mu + param * sd, 1.0, param
            else:
                aa = mu - ym - param
                z = aa / sd
                cdf, pdf = float(O.norm_cdf(z)), float(O.norm_pdf(z))
                a, ca, cs = (aa * cdf + sd * pdf, cdf, pdf) if acq == O.EI else (cdf, pdf / sd, -pdf * z / sd)
            return -a, -(ca * dmu + cs * dsd)

        box = np.asarray(box, dtype=np.float64)
        xs, fs, status = [], [], []
        for s0 in np.asarray(seeds, dtype=np.float64):
            res = minimize(f_grad, s0, jac=True, bounds=box, method="L-BFGS-B") if cons is None else \
                minimize(f_only, s0, bounds=box, method="L-BFGS-B")
            xs.append(res.x)
            fs.append(float(np.squeeze(res.fun)))
            status.append(0 if res.success else 2)
        self._resident = False
        return np.array(xs), np.array(fs), np.array(status, dtype=np.int32), 0

    def acq_argbest(self, acq, param, y_max=0.0, lb=None, ub=None, k_seeds=0, index_offset=0, return_values=False):
        self.calls.append(("acq_argbest", acq, k_seeds))
        mu, sd = self.post[0]
        ys = -1 * O.base_acq(acq, mu, sd, param, y_max)
        if lb is not None:
            p = None
            for j in range(len(lb)):
                cm, cs = self.post[j + 1]
                pl = O._cdf_loc_scale(lb[j], cm, cs) if lb[j] != -np.inf else np.array([0.0])
                pu = O._cdf_loc_scale(ub[j], cm, cs) if ub[j] != np.inf else np.array([1.0])
                p = (pu - pl) if p is None else p * (pu - pl)
            ys = ys * p
        if np.isnan(ys).any():
            bi, bv = int(np.flatnonzero(np.isnan(ys))[0]), float("nan")
        else:
            bi, bv = int(ys.argmin()), float(ys.min())
        nan = np.isnan(ys)
        order = np.lexsort((np.arange(len(ys)), np.where(nan, np.inf, ys) + 0.0, nan))[:k_seeds]
        return bi + index_offset, bv, order + index_offset, ys[order], (ys if return_values else None)


def philox4x32_10_uniform(M, d, lo, hi, seed):
    """NumPy restatement of candidates.hip (Philox4x32-10, counter = pair index, key = 64-bit seed) -> (M, d)."""
    total = M * d
    nq = (total + 1) // 2
    q = np.arange(nq, dtype=np.uint64)
    c = [(q & np.uint64(0xFFFFFFFF)).astype(np.uint64), (q >> np.uint64(32)).astype(np.uint64),
         np.zeros(nq, np.uint64), np.zeros(nq, np.uint64)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32
        n1 = p1 & m32
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32
        n3 = p0 & m32
        c = [n0, n1, n2, n3]
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    u = np.empty(2 * nq)
    for h in range(2):
        a = (c[2 * h] >> np.uint64(5)).astype(np.float64)
        b = (c[2 * h + 1] >> np.uint64(6)).astype(np.float64)
        u[h::2] = (a * 67108864.0 + b) / 9007199254740992.0
    u = u[:total].reshape(M, d)
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    return lo + (hi - lo) * u


def mt19937_device_mirror(key, pos0, M, d, lo, hi):
    """NumPy mirror of mt19937_uniform_kernel (csrc/mt19937.hip): same three-phase block walk, same pairing of
    words across block boundaries, same (row, col) bookkeeping.  Returns (Xc, key', pos').  Tests only."""
    N_, M_ = 624, 397
    key = np.array(key, dtype=np.uint32)
    T = M * d
    words, avail = 2 * T, N_ - pos0
    n_blocks = (words - avail + N_ - 1) // N_ if words > avail else 0
    Xc = np.full((M, d), np.nan)
    rng_ = np.asarray(hi, dtype=np.float64) - np.asarray(lo, dtype=np.float64)

    def twist(cur, nxt, far):
        y = (cur & np.uint32(0x80000000)) | (nxt & np.uint32(0x7FFFFFFF))
        return far ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), np.uint32(0x9908B0DF), np.uint32(0))

    def temper(y):
        y = y ^ (y >> np.uint32(11))
        y = y ^ ((y << np.uint32(7)) & np.uint32(0x9D2C5680))
        y = y ^ ((y << np.uint32(15)) & np.uint32(0xEFC60000))
        return y ^ (y >> np.uint32(18))

    buf = [key.copy(), np.zeros(N_, dtype=np.uint32)]
    row = col = 0
    for b in range(n_blocks + 1):
        cur, prev = buf[b & 1], buf[(b & 1) ^ 1]
        if b > 0:
            k = np.arange(0, 227)
            cur[k] = twist(prev[k], prev[k + 1], prev[k + M_])
            k = np.arange(227, 454)
            cur[k] = twist(prev[k], prev[k + 1], cur[k - 227])
            k = np.arange(454, 624)
            nxt = np.where(k == 623, cur[0], prev[np.minimum(k + 1, 623)])
            cur[k] = twist(prev[k], nxt, cur[k - 227])
        v_lo, v_hi = N_ * b, N_ * b + N_
        t_lo = (v_lo - pos0) // 2 if v_lo - pos0 - 1 >= 0 else 0
        t_hi = min(T, (v_hi - pos0 - 2) // 2 + 1 if v_hi - pos0 - 2 >= 0 else 0)
        cnt = max(0, t_hi - t_lo)
        if cnt:
            o = np.arange(cnt)
            t = t_lo + o
            i2 = pos0 + 2 * t + 1 - v_lo
            w2 = cur[i2]
            w1 = np.where(i2 > 0, cur[np.maximum(i2 - 1, 0)], prev[623])
            a = (temper(w1) >> np.uint32(5)).astype(np.float64)
            bb = (temper(w2) >> np.uint32(6)).astype(np.float64)
            u = (a * 67108864.0 + bb) / 9007199254740992.0
            r = row + o
            c = col + r // M
            r = r % M
            Xc[r, c] = np.asarray(lo)[c] + rng_[c] * u
        row += cnt
        col += row // M
        row %= M
    pos = pos0 + words if words <= avail else (words - avail - 1) % N_ + 1
    return Xc, buf[n_blocks & 1].copy(), pos


@contextlib.contextmanager
def fit_paths(fused=None, mid=None):
    """Pin the debug build's fit dispatch for the duration: fused = largest NP of the one-workgroup kernel (csrc/fused_small.hip),
    mid = largest NP of the strip path (csrc/mid_fit.hip); 0 turns a path off, None leaves the product's rule.  fit_paths(0, 0) is
    the multi-launch sequence at every size.  (GPBO_FUSED_MAX_NP / GPBO_MID_MAX_NP are read per call by libgpbo_dbg.so only.)"""
    names = {"GPBO_FUSED_MAX_NP": fused, "GPBO_MID_MAX_NP": mid}
    old = {k: os.environ.get(k) for k in names}
    for k, v in names.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
