"""GpEngine — thin Python owner of one libgpbo context (one GPU, one HIP stream).

Slot 0 holds the target GP, slots 1.. the constraint GPs; the candidate matrix is uploaded once and
stays resident in HBM across posterior / acquisition calls.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import json

import numpy as np

from . import _lib
from ._lib import dptr, iptr

RBF, MATERN25 = 0, 1
UCB, EI, POI = 0, 1, 2
F64, F32 = 0, 1

TIMING_NAMES = ("fit", "posterior_main", "posterior_finalize", "acq_argbest", "kmat", "cholesky", "trtri")


def advance_mt19937(random_state, n_words: int, lib=None):
    """Advance a legacy MT19937 `RandomState` by `n_words` 32-bit outputs WITHOUT generating them: whole 624-word blocks by
    the polynomial jump the device generator uses (gpbo_mt19937_jump_blocks, host code: x^J mod phi applied to the state),
    the rest by position.  What `n_words // 2` calls of `random_sample()` would leave behind, bit for bit — a rank that
    generated only ITS rows of a candidate matrix on its GPU uses it to leave its RandomState where the reference's
    `space.random_sample(M, random_state)` would (tests/test_mt_jump_host.py checks it against NumPy, no GPU needed)."""
    lib = lib or _lib.load_library()
    name, key, pos, has_gauss, cached = random_state.get_state(legacy=True)
    if name != "MT19937":
        raise TypeError("advance_mt19937 needs an MT19937 RandomState")
    n_words = int(n_words)
    avail = 624 - int(pos)
    if n_words <= avail:
        random_state.set_state((name, key, int(pos) + n_words, has_gauss, cached))
        return
    n_blocks = (n_words - avail + 623) // 624
    key_in = np.ascontiguousarray(key, dtype=np.uint32)
    key_out = np.empty(624, dtype=np.uint32)
    rc = lib.gpbo_mt19937_jump_blocks(key_in.ctypes.data_as(C.POINTER(C.c_uint32)), n_blocks,
                                      key_out.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc != _lib.GPBO_OK:
        _lib.raise_for_status(lib, None, rc)
    random_state.set_state((name, key_out, (n_words - avail - 1) % 624 + 1, has_gauss, cached))


_INT_ARRAYS: dict = {}      # ctypes array types by length (creating one costs ~2 us per call)


class GpEngine:
    """One engine context = one GPU, one HIP stream, 8 model slots (0 = target GP, 1.. = constraint GPs) and one resident
    candidate matrix.  Calls are synchronous from the host's point of view unless noted (`posterior(fetch=False)` only
    enqueues); a context is not thread-safe — the lockstep helpers keep every device call on the serving thread."""

    def __init__(self, device: int = 0, debug: bool = False):
        # debug=True: the context lives in libgpbo_dbg.so (same sources + gpbo_debug_* entry points and the kernel A/B
        # environment switches) — tests and scripts only
        self._lib = _lib.load_debug_library() if debug else _lib.load_library()
        self.debug = bool(debug)
        h = C.c_void_p()
        rc = self._lib.gpbo_create(int(device), C.byref(h))
        if rc != _lib.GPBO_OK:
            _lib.raise_for_status(self._lib, None, rc)
        self._h = h
        self.device = int(device)
        self.n_candidates = 0
        self.world_size = 1
        self.rank = 0
        self._serial: dict[int, int] = {}    # slot -> number of times its factorisation was rewritten
        self._overlap_depth = 0              # > 0 inside overlapped_fits()
        self._pending_fits: set[int] = set()  # slots with a gpbo_fit_begin not yet waited for
        self.timing = True                   # the calls record their HIP event pairs (set_timing)

    # -- lifecycle ---------------------------------------------------------------------------
    def __deepcopy__(self, memo):
        # a context is a device resource, not data: estimators that get deep-copied (sklearn.base.clone, bayes_opt's
        # ConstantLiar copying a constrained target space) keep pointing at the same engine
        return self

    __copy__ = lambda self: self  # noqa: E731

    def __reduce__(self):
        raise TypeError("GpEngine holds a GPU context and cannot be pickled; create one per process (GpEngine(device))")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpbo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, info=0):
        if rc != _lib.GPBO_OK:
            _lib.raise_for_status(self._lib, self._h, rc, info)

    def synchronize(self):
        self._settle()
        self._check(self._lib.gpbo_synchronize(self._h))

    def device_info(self) -> dict:
        buf = C.create_string_buffer(1024)
        self._check(self._lib.gpbo_device_info(self._h, buf, 1024))
        return json.loads(buf.value.decode())

    # -- fit ---------------------------------------------------------------------------------
    def fit(self, X, y_norm, kernel: int, length_scale, noise: float, slot: int = 0, precision: int = F64):
        """K + noise*I -> L, W = L^-1, alpha, at fixed theta (sklearn _gpr.py:346-364)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError("X must be 2-D (n_samples, n_features)")
        y_norm = np.ascontiguousarray(y_norm, dtype=np.float64).ravel()
        if y_norm.shape[0] != X.shape[0]:
            raise ValueError("X and y have inconsistent numbers of samples")
        ls = np.ascontiguousarray(np.atleast_1d(np.asarray(length_scale, dtype=np.float64)))
        info = C.c_int(0)
        self._settle(slot)
        self._touch(slot)
        if self._overlap_depth > 0:
            # inside overlapped_fits(): enqueue on the slot's own stream and return; waited for (and checked) at the end
            # of the block or by the first call that reads the slot
            rc = self._lib.gpbo_fit_begin(self._h, int(slot), dptr(X), dptr(y_norm), X.shape[0], X.shape[1], int(kernel),
                                          dptr(ls), int(ls.shape[0]), float(noise), int(precision))
            self._check(rc)
            self._pending_fits.add(int(slot))
            # gpbo_fit_begin copies X / y asynchronously: the arrays it was given stay referenced until the fit is waited for
            self._pending_inputs = getattr(self, "_pending_inputs", {})
            self._pending_inputs[int(slot)] = (X, y_norm, ls)
            return self._touch(slot)
        rc = self._lib.gpbo_fit(self._h, int(slot), dptr(X), dptr(y_norm), X.shape[0], X.shape[1], int(kernel),
                                dptr(ls), int(ls.shape[0]), float(noise), int(precision), C.byref(info))
        self._check(rc, info.value)
        return self._touch(slot)

    @contextlib.contextmanager
    def overlapped_fits(self):
        """`fit()` calls inside the block are enqueued on their slots' own streams (gpbo_fit_begin) and overlap on the
        device — the target GP and the constraint GPs of one suggest() (bayes_opt/acquisition.py:84-86).  Leaving the
        block waits for all of them; a kernel matrix that is not positive definite raises there (np.linalg.LinAlgError
        with sklearn's hint, as fit() does).  Any call that reads a slot in between waits for that slot first."""
        self._overlap_depth += 1
        try:
            yield self
        except BaseException:
            # something else failed inside the block: still leave no fit in flight, but let THAT error propagate
            self._overlap_depth -= 1
            if self._overlap_depth == 0:
                try:
                    self.wait_fits()
                except Exception:  # noqa: BLE001
                    pass
            raise
        else:
            self._overlap_depth -= 1
            if self._overlap_depth == 0:
                self.wait_fits()

    def wait_fits(self):
        """Wait for every pending gpbo_fit_begin; all are waited for before the first failure is raised."""
        first = None
        for slot in sorted(self._pending_fits):
            try:
                self._wait_fit(slot)
            except Exception as e:  # noqa: BLE001
                first = first or e
        if first is not None:
            raise first

    def _wait_fit(self, slot: int):
        self._pending_fits.discard(int(slot))
        info = C.c_int(0)
        rc = self._lib.gpbo_fit_wait(self._h, int(slot), C.byref(info))
        getattr(self, "_pending_inputs", {}).pop(int(slot), None)
        self._check(rc, info.value)

    def _settle(self, slot=None):
        """Before a call reads or rewrites a slot (None: any slot): its pending fit, if any, has to be complete."""
        if not self._pending_fits:
            return
        if slot is None:
            self.wait_fits()
        elif int(slot) in self._pending_fits:
            self._wait_fit(int(slot))

    def fit_append(self, x_new, y_norm, slot: int = 0):
        """Grow the slot's fitted model by the rows `x_new` at unchanged kernel/length scale/noise (gpbo_fit_append);
        `y_norm` = ALL normalised targets, old and new.  `x_new` may be empty (new targets for the same inputs)."""
        self._settle(slot)
        y_norm = np.ascontiguousarray(y_norm, dtype=np.float64).ravel()
        x_new = np.ascontiguousarray(x_new, dtype=np.float64)
        if x_new.ndim != 2:
            raise ValueError("x_new must be 2-D (n_new, n_features)")
        info = C.c_int(0)
        self._touch(slot)
        rc = self._lib.gpbo_fit_append(self._h, int(slot), dptr(x_new) if x_new.shape[0] else None, x_new.shape[0],
                                       x_new.shape[1], dptr(y_norm), y_norm.shape[0], C.byref(info))
        self._check(rc, info.value)
        return self._touch(slot)

    def lml_batch(self, X, y_norm, kernel: int, length_scales, noise: float, eval_gradient=True, reuse_inputs=False):
        """[(lml, grad)] for every row of `length_scales` (n_theta x n_ls), evaluated side by side on the device
        (gpbo_lml_batch); each entry is bitwise what `lml()` returns for that row.  Model slots are not touched.
        reuse_inputs=True: (X, y_norm) are the arrays of the previous call and are not uploaded again."""
        vals, grads = self.lml_batch_arrays(X, y_norm, kernel, length_scales, noise, eval_gradient, reuse_inputs)
        return [(float(vals[i]), grads[i].copy()) for i in range(vals.shape[0])]

    def lml_batch_arrays(self, X, y_norm, kernel: int, length_scales, noise: float, eval_gradient=True, reuse_inputs=False):
        """`lml_batch` as two arrays — values (n_theta,) and gradients (n_theta, n_ls): what a theta-search round needs, without
        the per-lane tuples (a round is ~50-100 us of device time at small N: every microsecond of Python around it shows)."""
        self._settle()
        ls = np.ascontiguousarray(np.atleast_2d(np.asarray(length_scales, dtype=np.float64)))
        n, n_ls = ls.shape
        vals = np.zeros(n)
        grads = np.zeros((n, n_ls))
        infos = _INT_ARRAYS.get(n)
        if infos is None:
            infos = _INT_ARRAYS[n] = C.c_int * n
        if reuse_inputs:
            xp = yp = None
        else:
            X = np.ascontiguousarray(X, dtype=np.float64)
            y_norm = np.ascontiguousarray(y_norm, dtype=np.float64).ravel()
            xp, yp = dptr(X), dptr(y_norm)
        rc = self._lib.gpbo_lml_batch(self._h, n, xp, yp, X.shape[0], X.shape[1], int(kernel), dptr(ls), n_ls, float(noise),
                                      int(bool(eval_gradient)), dptr(vals), dptr(grads), infos())
        if rc:
            self._check(rc)
        return vals, grads

    def lml_search_rounds(self, X, y_norm, kernel: int, n_ls: int, noise: float):
        """`round(scales (n <= 8, n_ls)) -> (values (n,), gradients (n, n_ls))` for the rounds of ONE theta search: the inputs are
        uploaded by the first round and stay resident (gpbo_lml_batch with X = y = NULL afterwards), and everything a round needs
        around the library call — the length-scale block, the result arrays, their ctypes pointers — is made once here instead of
        per round (a round at N <= 64 is a 41 us kernel: the ~10 us of argument handling around it showed).  The arrays returned are
        the frame's own: valid until the next round."""
        self._settle()
        X = np.ascontiguousarray(X, dtype=np.float64)
        y_norm = np.ascontiguousarray(y_norm, dtype=np.float64).ravel()
        N, d = int(X.shape[0]), int(X.shape[1])
        ls = np.ones((8, n_ls))
        vals = np.zeros(8)
        grads = np.zeros((8, n_ls))
        infos = (C.c_int * 8)()
        p_ls, p_vals, p_grads = dptr(ls), dptr(vals), dptr(grads)
        xp, yp = [dptr(X)], [dptr(y_norm)]
        call, h, kind, nz, ck = self._lib.gpbo_lml_batch, self._h, int(kernel), float(noise), self._check

        def one_round(scales):
            n = scales.shape[0]
            ls[:n] = scales
            rc = call(h, n, xp[0], yp[0], N, d, kind, p_ls, n_ls, nz, 1, p_vals, p_grads, infos)
            xp[0] = yp[0] = None              # (X, y_norm stay referenced by this closure for the search's lifetime)
            if rc:
                ck(rc)
            return vals[:n], grads[:n]

        return one_round

    def _touch(self, slot: int) -> int:
        """Every call that rewrites a slot's factorisation bumps its serial; an estimator compares the serial it got
        from its last fit with `fit_serial(slot)` to know whether the slot still holds ITS model."""
        self._serial[int(slot)] = self._serial.get(int(slot), 0) + 1
        return self._serial[int(slot)]

    def fit_serial(self, slot: int = 0) -> int:
        return self._serial.get(int(slot), 0)

    def lml(self, X, y_norm, kernel: int, length_scale, noise: float, eval_gradient=True, slot: int = 0):
        """(log marginal likelihood, d/dlog(length_scale)) at theta (sklearn _gpr.py:575-652). Clobbers the slot's fit."""
        self._settle(slot)
        X = np.ascontiguousarray(X, dtype=np.float64)
        y_norm = np.ascontiguousarray(y_norm, dtype=np.float64).ravel()
        ls = np.ascontiguousarray(np.atleast_1d(np.asarray(length_scale, dtype=np.float64)))
        val = C.c_double(0.0)
        grad = np.zeros(ls.shape[0])
        info = C.c_int(0)
        self._touch(slot)
        rc = self._lib.gpbo_lml(self._h, int(slot), dptr(X), dptr(y_norm), X.shape[0], X.shape[1], int(kernel),
                                dptr(ls), int(ls.shape[0]), float(noise), int(bool(eval_gradient)), C.byref(val),
                                dptr(grad), C.byref(info))
        self._check(rc, info.value)
        return (val.value, grad) if eval_gradient else val.value

    def _square(self, fn, slot, n):
        out = np.empty((n, n), dtype=np.float64)
        self._check(fn(self._h, int(slot), dptr(out)))
        return out

    def get_K(self, n, slot=0):
        self._settle(slot)
        return self._square(self._lib.gpbo_get_K, slot, n)

    def get_L(self, n, slot=0):
        self._settle(slot)
        return self._square(self._lib.gpbo_get_L, slot, n)

    def get_Linv(self, n, slot=0):
        self._settle(slot)
        return self._square(self._lib.gpbo_get_Linv, slot, n)

    def get_alpha(self, n, slot=0):
        self._settle(slot)
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.gpbo_get_alpha(self._h, int(slot), dptr(out)))
        return out

    # -- candidates / posterior ------------------------------------------------------------------
    def set_candidates(self, Xc):
        Xc = np.ascontiguousarray(Xc, dtype=np.float64)
        if Xc.ndim != 2:
            raise ValueError("candidates must be 2-D (M, d)")
        self._check(self._lib.gpbo_set_candidates(self._h, dptr(Xc), Xc.shape[0], Xc.shape[1]))
        self.n_candidates = Xc.shape[0]

    def generate_candidates(self, M: int, lo, hi, seed: int):
        """Throughput mode: M x d uniforms in [lo, hi) generated on the device (Philox); NOT the reference's stream."""
        lo = np.ascontiguousarray(lo, dtype=np.float64).ravel()
        hi = np.ascontiguousarray(hi, dtype=np.float64).ravel()
        if lo.shape != hi.shape:
            raise ValueError("lo and hi must have the same length")
        self._check(self._lib.gpbo_generate_candidates(self._h, int(M), lo.shape[0], dptr(lo), dptr(hi),
                                                       int(seed) & 0xFFFFFFFFFFFFFFFF))
        self.n_candidates = int(M)
        self._cand_dim = lo.shape[0]

    def generate_candidates_like(self, M: int, lo, hi, random_state):
        """Index-parity mode: what `np.column_stack([random_state.uniform(lo[t], hi[t], M) for t in range(d)])`
        would be, generated on the device from `random_state`'s MT19937 state; `random_state` (a legacy
        np.random.RandomState) is advanced exactly as those draws would advance it."""
        lo = np.ascontiguousarray(lo, dtype=np.float64).ravel()
        hi = np.ascontiguousarray(hi, dtype=np.float64).ravel()
        if lo.shape != hi.shape:
            raise ValueError("lo and hi must have the same length")
        if not np.all(np.isfinite(hi - lo)):
            raise OverflowError("Range exceeds valid bounds")        # as RandomState.uniform
        name, key, pos, has_gauss, cached = random_state.get_state(legacy=True)
        if name != "MT19937":
            raise TypeError("generate_candidates_like needs an MT19937 RandomState")
        key = np.ascontiguousarray(key, dtype=np.uint32).copy()
        cpos = C.c_int(int(pos))
        self._check(self._lib.gpbo_generate_candidates_mt19937(
            self._h, int(M), lo.shape[0], dptr(lo), dptr(hi), key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(cpos)))
        random_state.set_state((name, key, cpos.value, has_gauss, cached))
        self.n_candidates = int(M)
        self._cand_dim = lo.shape[0]

    def generate_candidate_rows_like(self, M: int, lo, hi, random_state, row_begin: int, row_end: int):
        """Rows [row_begin, row_end) of the matrix `generate_candidates_like(M, lo, hi, random_state)` would leave resident
        — this rank's block of ONE reference stream in the one-process-per-GPU mode (gpbo_generate_candidate_rows_mt19937:
        every column's run of rows starts at a jump-ahead state, nothing is generated twice, nothing crosses PCIe).  Every
        rank passes a RandomState in the same state; every rank's is advanced past the WHOLE matrix — on the host, by the
        same polynomial jump (advance_mt19937) — so the ranks' streams stay in step without exchanging anything."""
        lo = np.ascontiguousarray(lo, dtype=np.float64).ravel()
        hi = np.ascontiguousarray(hi, dtype=np.float64).ravel()
        if lo.shape != hi.shape:
            raise ValueError("lo and hi must have the same length")
        if not np.all(np.isfinite(hi - lo)):
            raise OverflowError("Range exceeds valid bounds")
        name, key, pos, has_gauss, cached = random_state.get_state(legacy=True)
        if name != "MT19937":
            raise TypeError("generate_candidate_rows_like needs an MT19937 RandomState")
        key = np.ascontiguousarray(key, dtype=np.uint32).copy()
        self._check(self._lib.gpbo_generate_candidate_rows_mt19937(
            self._h, int(M), lo.shape[0], int(row_begin), int(row_end), dptr(lo), dptr(hi),
            key.ctypes.data_as(C.POINTER(C.c_uint32)), int(pos), None, None))
        advance_mt19937(random_state, 2 * int(M) * lo.shape[0], self._lib)
        self.n_candidates = int(row_end) - int(row_begin)
        self._cand_dim = lo.shape[0]

    def get_candidate_rows(self, idx, d: int):
        idx = np.ascontiguousarray(np.atleast_1d(idx), dtype=np.int64)
        out = np.empty((idx.shape[0], d))
        self._check(self._lib.gpbo_get_candidate_rows(self._h, iptr(idx), idx.shape[0], dptr(out)))
        return out

    #: the column-group assembly below exists on a single device only (a device group samples mixed spaces on the host)
    mixed_device_sampling = True

    def generate_candidates_mixed(self, M: int, groups, random_state):
        """`space.random_sample(M, random_state)` of a space with float AND integer / categorical parameters, resident on
        the device: `groups` = [(kind, col0, ncols, lo, hi, param)] in key order (kind 0 float, 1 int, 2 categorical).
        Runs of float groups are generated on the device from the caller's MT19937 state (gpbo_generate_candidate_columns_
        mt19937; the state comes back advanced); an int / categorical parameter is drawn by ITS OWN `random_sample` on the
        host from exactly that position (RandomState.randint: masked rejection sampling, its word consumption depends on
        the values) and uploaded as its columns (gpbo_set_candidate_columns).  Matrix and RandomState end up bit for bit
        where the reference's loop (target_space.py:593-600) leaves them."""
        name, key, pos, has_gauss, cached = random_state.get_state(legacy=True)
        if name != "MT19937":
            raise TypeError("generate_candidates_mixed needs an MT19937 RandomState")
        M = max(1, int(M))
        d_total = sum(g[2] for g in groups)
        key = np.ascontiguousarray(key, dtype=np.uint32).copy()
        cpos = C.c_int(int(pos))
        i = 0
        while i < len(groups):
            kind, col0, ncols, lo, hi, param = groups[i]
            if kind == 0:
                j = i
                while j + 1 < len(groups) and groups[j + 1][0] == 0:      # a run of float parameters: one device call
                    j += 1
                lo_v = np.ascontiguousarray(np.concatenate([np.atleast_1d(g[3]) for g in groups[i:j + 1]]), dtype=np.float64)
                hi_v = np.ascontiguousarray(np.concatenate([np.atleast_1d(g[4]) for g in groups[i:j + 1]]), dtype=np.float64)
                if not np.all(np.isfinite(hi_v - lo_v)):
                    raise OverflowError("Range exceeds valid bounds")
                self._check(self._lib.gpbo_generate_candidate_columns_mt19937(
                    self._h, M, d_total, int(col0), int(lo_v.shape[0]), dptr(lo_v), dptr(hi_v),
                    key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(cpos)))
                i = j + 1
                continue
            random_state.set_state((name, key, cpos.value, has_gauss, cached))
            vals = np.ascontiguousarray(np.asarray(param.random_sample(M, random_state), dtype=np.float64).reshape(M, ncols))
            self._check(self._lib.gpbo_set_candidate_columns(self._h, dptr(vals), M, d_total, int(col0), int(ncols)))
            _, key, p2, has_gauss, cached = random_state.get_state(legacy=True)
            key = np.ascontiguousarray(key, dtype=np.uint32).copy()
            cpos = C.c_int(int(p2))
            i += 1
        random_state.set_state((name, key, cpos.value, has_gauss, cached))
        self.n_candidates = M
        self._cand_dim = d_total

    def transform_candidates(self, groups):
        """TargetSpace.kernel_transform over the resident candidates, on the device (gpbo_transform_candidates): the
        posterior sees round()ed integer columns and the reference's one-hot categorical columns; get_candidate_rows keeps
        returning the rows as drawn."""
        n = len(groups)
        kinds = (C.c_int * n)(*[int(g[0]) for g in groups])
        col0 = (C.c_int * n)(*[int(g[1]) for g in groups])
        ncols = (C.c_int * n)(*[int(g[2]) for g in groups])
        self._check(self._lib.gpbo_transform_candidates(self._h, n, kinds, col0, ncols))

    def posterior(self, slot=0, y_mean=0.0, y_std=1.0, fetch=True):
        """mu, sd for the resident candidates (sklearn _gpr.py:443-494). fetch=False keeps them on device."""
        self._settle(slot)
        M = self.n_candidates
        mu = np.empty(M) if fetch else None
        sd = np.empty(M) if fetch else None
        self._check(self._lib.gpbo_posterior(self._h, int(slot), float(y_mean), float(y_std), dptr(mu), dptr(sd)))
        return mu, sd

    def predict(self, Xc, slot=0, y_mean=0.0, y_std=1.0):
        self.set_candidates(Xc)
        return self.posterior(slot, y_mean, y_std, fetch=True)

    def take_negative_variance_flag(self):
        """True when a posterior / predict since the last call clipped a NEGATIVE variance — sklearn's warning
        condition (_gpr.py:479-485); clears the flag (gpbo_take_negative_variance_flag)."""
        seen = C.c_int(0)
        self._check(self._lib.gpbo_take_negative_variance_flag(self._h, C.byref(seen)))
        return bool(seen.value)

    def predict_cov(self, Xc, slot=0, y_mean=0.0, y_std=1.0):
        """(mu (M,), cov (M,M)) as GaussianProcessRegressor.predict(return_cov=True) (gpbo_predict_cov)."""
        self._settle(slot)
        Xc = np.ascontiguousarray(Xc, dtype=np.float64)
        M, d = Xc.shape
        mu, cov = np.empty(M), np.empty((M, M))
        self._check(self._lib.gpbo_predict_cov(self._h, int(slot), dptr(Xc), M, d, float(y_mean), float(y_std),
                                               dptr(mu), dptr(cov)))
        self.n_candidates = M
        self._resident = False
        return mu, cov

    def predict_grad(self, Xc, slot=0, y_mean=0.0, y_std=1.0):
        """(mu (M,), sd (M,), dmu (M,d), dsd (M,d)) for a small host batch (M <= 256): gpbo_predict_grad."""
        self._settle(slot)
        Xc = np.ascontiguousarray(Xc, dtype=np.float64)
        M, d = Xc.shape
        mu, sd = np.empty(M), np.empty(M)
        dmu, dsd = np.empty((M, d)), np.empty((M, d))
        self._check(self._lib.gpbo_predict_grad(self._h, int(slot), dptr(Xc), M, d, float(y_mean), float(y_std),
                                                dptr(mu), dptr(sd), dptr(dmu), dptr(dsd)))
        self.n_candidates = M
        self._resident = False       # (device groups: the first device's candidate buffer was re-used)
        return mu, sd, dmu, dsd

    def polish_seeds(self, acq: int, param: float, y_max, lb, ub, y_means, y_stds, seeds, box, max_iter: int = 0):
        """gpbo_polish_seeds: the local searches of one suggest() in one call.  Returns (x (S,d), f (S,), status (S,) — 0/1
        converged, 2 SciPy's success = False —, rounds)."""
        self._settle()
        seeds = np.ascontiguousarray(seeds, dtype=np.float64)
        S, d = seeds.shape
        lb = np.ascontiguousarray(np.atleast_1d(np.asarray(lb, dtype=np.float64))) if lb is not None else None
        ub = np.ascontiguousarray(np.atleast_1d(np.asarray(ub, dtype=np.float64))) if ub is not None else None
        n_c = 0 if lb is None else lb.shape[0]
        ym = np.ascontiguousarray(np.atleast_1d(np.asarray(y_means, dtype=np.float64)))
        ys = np.ascontiguousarray(np.atleast_1d(np.asarray(y_stds, dtype=np.float64)))
        if ym.shape[0] != 1 + n_c or ys.shape[0] != 1 + n_c:
            raise ValueError("one (y_mean, y_std) per model slot")
        lo = np.ascontiguousarray(np.asarray(box, dtype=np.float64)[:, 0])
        hi = np.ascontiguousarray(np.asarray(box, dtype=np.float64)[:, 1])
        x = np.empty((S, d))
        f = np.empty(S)
        status = np.zeros(S, dtype=np.int32)
        self.last_polish = {"nit": np.zeros(S, dtype=np.int32), "nfev": np.zeros(S, dtype=np.int32)}
        rounds = C.c_int(0)
        self._check(self._lib.gpbo_polish_seeds(self._h, int(acq), float(param), float(0.0 if y_max is None else y_max), n_c,
                                                dptr(lb), dptr(ub), dptr(ym), dptr(ys), dptr(seeds), S, d, dptr(lo), dptr(hi),
                                                int(max_iter), dptr(x), dptr(f), status.ctypes.data_as(C.POINTER(C.c_int)),
                                                C.byref(rounds), self.last_polish["nit"].ctypes.data_as(C.POINTER(C.c_int)),
                                                self.last_polish["nfev"].ctypes.data_as(C.POINTER(C.c_int))))
        self.last_polish["rounds"] = rounds.value
        self.n_candidates = S        # the candidate buffer held the rounds' trial points (at most S of them)
        self._resident = False
        return x, f, status, rounds.value

    # -- acquisition -----------------------------------------------------------------------------
    def acq_argbest(self, acq: int, param: float, y_max: float = 0.0, lb=None, ub=None, k_seeds: int = 0,
                    index_offset: int = 0, return_values: bool = False):
        """ys = -acq [* p_c]; returns (best_idx, best_val, seed_idx, seed_val, ys|None)."""
        lb = np.ascontiguousarray(np.atleast_1d(np.asarray(lb, dtype=np.float64))) if lb is not None else None
        ub = np.ascontiguousarray(np.atleast_1d(np.asarray(ub, dtype=np.float64))) if ub is not None else None
        n_c = 0 if lb is None else lb.shape[0]
        if n_c and (ub is None or ub.shape[0] != n_c):
            raise ValueError("lb and ub must have the same length")
        best_idx, best_val = C.c_int64(0), C.c_double(0.0)
        seed_idx = np.full(max(k_seeds, 1), -1, dtype=np.int64)
        seed_val = np.full(max(k_seeds, 1), np.nan)
        ys = np.empty(self.n_candidates) if return_values else None
        rc = self._lib.gpbo_acq_argbest(self._h, int(acq), float(param), float(y_max if y_max is not None else 0.0),
                                        n_c, dptr(lb), dptr(ub), int(k_seeds), int(index_offset),
                                        C.byref(best_idx), C.byref(best_val), iptr(seed_idx), dptr(seed_val),
                                        dptr(ys))
        self._check(rc)
        return best_idx.value, best_val.value, seed_idx[:k_seeds], seed_val[:k_seeds], ys

    # -- timing / probes (debug_* and the latency / hybrid probes need GpEngine(debug=True)) ----------------------------
    def _need_debug(self, what: str):
        if not getattr(self, "debug", False):
            raise _lib.GpboError(f"{what} is a debug entry point: create the engine with GpEngine(device, debug=True) (libgpbo_dbg.so)")

    def debug_fail_next_acq(self):
        """Fault injection (debug build): the next comm_acq_argbest behaves as if this rank's local pass had failed."""
        self._need_debug("gpbo_debug_fail_next_acq")
        self._check(self._lib.gpbo_debug_fail_next_acq(self._h))

    def set_timing(self, on: bool):
        """gpbo_set_timing: whether the calls record their HIP event pairs (what `last_timings()` reads; a new engine does).
        A record is a marker packet on the stream: a 0.1 ms step notices its eight."""
        self._check(self._lib.gpbo_set_timing(self._h, 1 if on else 0))
        self.timing = bool(on)

    def last_timings(self) -> dict:
        ms = (C.c_float * 8)()
        self._check(self._lib.gpbo_last_timings(self._h, ms, 8))
        return {n: float(ms[i]) for i, n in enumerate(TIMING_NAMES)}

    def debug_cholesky(self, A, variant=3, iters=1):
        """The device Cholesky alone (gpbo_debug_cholesky): returns (L lower n x n, dinv [n/64][64][64], stamps[16], ms, info)."""
        self._need_debug("gpbo_debug_cholesky")
        A = np.ascontiguousarray(A, dtype=np.float64)
        n = A.shape[0]
        Lo = np.empty((n, n))
        dinv = np.empty((n // 64, 64, 64))
        stamps = np.zeros(16, dtype=np.int64)
        ms, info = C.c_double(0.0), C.c_int(0)
        self._check(self._lib.gpbo_debug_cholesky(self._h, dptr(A), n, int(variant), int(iters), dptr(Lo), dptr(dinv),
                                                  stamps.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(ms), C.byref(info)))
        return np.tril(Lo), dinv, stamps, ms.value, info.value

    def debug_select(self, ys, k, variant=1, iters=0):
        """The selection launches alone over `ys` (gpbo_debug_select): (idx (k,), vals (k,), first_nan, ms per selection)."""
        self._need_debug("gpbo_debug_select")
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        idx = np.empty(int(k), dtype=np.int64)
        vals = np.empty(int(k))
        first_nan, ms = C.c_int64(-1), C.c_float(0.0)
        self._check(self._lib.gpbo_debug_select(self._h, dptr(ys), ys.shape[0], int(k), int(variant), int(iters),
                                                idx.ctypes.data_as(C.POINTER(C.c_int64)), dptr(vals), C.byref(first_nan), C.byref(ms)))
        return idx, vals, int(first_nan.value), float(ms.value)

    def latency_probe(self, n=16):
        self._need_debug("gpbo_debug_latency_probe")
        out = np.zeros(32, dtype=np.int64)
        self._check(self._lib.gpbo_debug_latency_probe(self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), int(n)))
        return out[:n]

    def debug_gemm(self, A, B, C_in=None, alpha=1.0, beta=0.0, b_trans=False):
        self._need_debug("gpbo_debug_gemm")
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        m, k = A.shape
        n = B.shape[0] if b_trans else B.shape[1]
        Cm = np.zeros((m, n)) if C_in is None else np.ascontiguousarray(C_in, dtype=np.float64).copy()
        self._check(self._lib.gpbo_debug_gemm(self._h, m, n, k, float(alpha), dptr(A), dptr(B), int(b_trans),
                                              float(beta), dptr(Cm)))
        return Cm

    def gemm_bench(self, m, n, k, b_trans=True, a_trans=False, lower_only=False, iters=10) -> dict:
        self._need_debug("gpbo_debug_gemm_bench")
        out = np.zeros(2)
        self._check(self._lib.gpbo_debug_gemm_bench(self._h, int(m), int(n), int(k), int(b_trans), int(a_trans),
                                                    int(lower_only), int(iters), dptr(out)))
        return {"ms": float(out[0]), "tflops": float(out[1])}

    def mfma_f64_peak(self, iters=20000) -> float:
        out = C.c_double(0.0)
        self._check(self._lib.gpbo_mfma_f64_peak(self._h, int(iters), C.byref(out)))
        return out.value

    def mfma_f64_probe(self, iters=20000, waves_per_simd=1, mode=0) -> dict:
        out = np.zeros(4)
        self._check(self._lib.gpbo_mfma_f64_probe(self._h, int(iters), int(waves_per_simd), int(mode), dptr(out)))
        return {"tflops": out[0], "cycles_per_mfma": out[1], "shader_mhz": out[2], "ms": out[3]}

    def hybrid_probe(self, iters=2000, cfg=2) -> dict:
        self._need_debug("gpbo_hybrid_probe")
        out = np.zeros(3)
        self._check(self._lib.gpbo_hybrid_probe(self._h, int(iters), int(cfg), dptr(out)))
        return {"ms": out[0], "mfma_tflops": out[1], "valu_tflops": out[2]}

    def hbm_copy_peak(self, nbytes=1 << 30) -> float:
        out = C.c_double(0.0)
        self._check(self._lib.gpbo_hbm_copy_peak(self._h, int(nbytes), C.byref(out)))
        return out.value

    # -- RCCL ----------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        lib = _lib.load_library()
        buf = C.create_string_buffer(128)
        rc = lib.gpbo_comm_unique_id(buf)
        if rc != _lib.GPBO_OK:
            _lib.raise_for_status(lib, None, rc)
        return buf.raw

    def comm_init(self, unique_id: bytes, world_size: int, rank: int):
        assert len(unique_id) == 128
        self._check(self._lib.gpbo_comm_init(self._h, unique_id, int(world_size), int(rank)))
        self.world_size, self.rank = int(world_size), int(rank)

    def comm_acq_argbest(self, acq: int, param: float, y_max: float = 0.0, lb=None, ub=None, k_seeds: int = 0,
                         index_offset: int = 0, return_values: bool = False):
        """`acq_argbest` over the union of every rank's shard (gpbo_comm_acq_argbest): the shard's records are packed on
        the device, all-gathered over RCCL and merged identically on every rank.  Same return tuple; `ys` is this
        rank's shard."""
        lb = np.ascontiguousarray(np.atleast_1d(np.asarray(lb, dtype=np.float64))) if lb is not None else None
        ub = np.ascontiguousarray(np.atleast_1d(np.asarray(ub, dtype=np.float64))) if ub is not None else None
        n_c = 0 if lb is None else lb.shape[0]
        if n_c and (ub is None or ub.shape[0] != n_c):
            raise ValueError("lb and ub must have the same length")
        best_idx, best_val = C.c_int64(0), C.c_double(0.0)
        seed_idx = np.full(max(k_seeds, 1), -1, dtype=np.int64)
        seed_val = np.full(max(k_seeds, 1), np.nan)
        ys = np.empty(self.n_candidates) if return_values else None
        rc = self._lib.gpbo_comm_acq_argbest(self._h, int(acq), float(param), float(y_max if y_max is not None else 0.0),
                                             n_c, dptr(lb), dptr(ub), int(k_seeds), int(index_offset),
                                             C.byref(best_idx), C.byref(best_val), iptr(seed_idx), dptr(seed_val), dptr(ys))
        self._check(rc)
        return best_idx.value, best_val.value, seed_idx[:k_seeds], seed_val[:k_seeds], ys

    def comm_allreduce_max(self, value: float) -> float:
        """Drain this rank's stream, then the maximum of `value` over all ranks (barrier + max-over-ranks timing)."""
        v = np.array([float(value)])
        self._check(self._lib.gpbo_comm_allreduce_max(self._h, dptr(v)))
        return float(v[0])

    def comm_allgather_best(self, vals, idxs):
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        idxs = np.ascontiguousarray(idxs, dtype=np.int64)
        n = vals.shape[0]
        all_vals = np.empty(n * self.world_size)
        all_idxs = np.empty(n * self.world_size, dtype=np.int64)
        self._check(self._lib.gpbo_comm_allgather_best(self._h, dptr(vals), iptr(idxs), n, dptr(all_vals),
                                                       iptr(all_idxs)))
        return all_vals, all_idxs


class GroupEngine(GpEngine):
    """G GPUs of one node behind the GpEngine interface, in ONE process (gpbo_group_*).

    The model slots are replicated (every device factorises the same GP — deterministic, so the factors are identical),
    the resident candidate matrix is block-partitioned in index order (device r keeps rows [r M / G, (r + 1) M / G)),
    and `acq_argbest` ends in ONE exchange: each shard's 1 + k (value, global index) records, packed on the device,
    all-gathered over RCCL and merged identically on every rank.  Everything that is not M-scaled — single-point and
    finite-difference predicts of the host optimisers, the LML evaluations of the theta search, the parity accessors —
    runs on the first device's context through the inherited methods.

    `devices` listing a GPU more than once gives virtual ranks (several shards on one GPU, merged on the host): the
    single-GPU rehearsal of the sharded path.  `collective` says which exchange the group uses.
    """

    def __init__(self, devices):
        devices = [int(x) for x in devices]
        if not devices:
            raise ValueError("GroupEngine needs at least one device")
        self._lib = _lib.load_library()
        arr = (C.c_int * len(devices))(*devices)
        g = C.c_void_p()
        rc = self._lib.gpbo_group_create(len(devices), arr, C.byref(g))
        if rc != _lib.GPBO_OK:
            _lib.raise_for_status(self._lib, None, rc)
        self._g = g
        self._h = C.c_void_p(self._lib.gpbo_group_ctx(g, 0))     # borrowed: the first device's context
        self.devices = devices
        self.device = devices[0]
        self.n_candidates = 0
        self.world_size = len(devices)
        self.rank = 0
        self._serial = {}
        self._overlap_depth = 0     # group fits are synchronous on every device: overlapped_fits() is a plain block
        self._pending_fits = set()  # (never filled: the inherited accessors only look at it)
        self.timing = True
        self._resident = False      # the group's candidate shards are in place (a small predict on device 0 clobbers them)
        self.collective = self._lib.gpbo_group_collective(g).decode()
        self._orphans = []          # arrays of failed calls (see _gcheck)

    def close(self):
        if getattr(self, "_g", None):
            self._lib.gpbo_group_destroy(self._g)
            self._g = None
            self._h = None

    def _gcheck(self, rc, info=0, borrowed=()):
        """`borrowed`: the host arrays the call handed to the group's worker threads.  After a failed call — above all one that
        missed its deadline, where a worker may still be inside the job (include/gpbo.h: the caller's arrays must outlive the
        group then) — they stay referenced until close()."""
        if rc != _lib.GPBO_OK:
            if borrowed:
                self._orphans.append(borrowed)
            _lib.raise_for_status(self._lib, None, rc, info, group=self._g)

    def per_device_info(self) -> list:
        """`device_info()` of every device of the group: PCI bus id, rank / world and what RCCL says its communicator has
        (ncclCommCount) — a multi-GPU bench line quotes these, not what the launcher asked for."""
        out = []
        for r in range(self.world_size):
            buf = C.create_string_buffer(1024)
            h = C.c_void_p(self._lib.gpbo_group_ctx(self._g, r))
            self._check(self._lib.gpbo_device_info(h, buf, 1024))
            out.append(json.loads(buf.value.decode()))
        return out

    def set_timing(self, on: bool):
        for r in range(self.world_size):
            h = C.c_void_p(self._lib.gpbo_group_ctx(self._g, r))
            self._check(self._lib.gpbo_set_timing(h, 1 if on else 0))
        self.timing = bool(on)

    def per_device_timings(self) -> list:
        """`last_timings()` of every device of the group (HIP events on each device's own stream): a straggler shows here."""
        out = []
        for r in range(self.world_size):
            ms = (C.c_float * 8)()
            h = C.c_void_p(self._lib.gpbo_group_ctx(self._g, r))
            self._check(self._lib.gpbo_last_timings(h, ms, 8))
            out.append({n: float(ms[i]) for i, n in enumerate(TIMING_NAMES)})
        return out

    def synchronize(self):
        self._gcheck(self._lib.gpbo_group_synchronize(self._g))

    @contextlib.contextmanager
    def overlapped_fits(self):
        """gpbo_group_fit factorises on every device and returns when all are done: nothing is left in flight, so the
        block is a plain block (the single-device engine overlaps the slots' fits here)."""
        yield self

    def take_negative_variance_flag(self):
        """OR of the clipped-variance flags of ALL member devices (a sharded posterior runs on every one of them);
        clears each.  Same warning condition as the single-device path (sklearn _gpr.py:479-485)."""
        any_seen = False
        for r in range(self.world_size):
            seen = C.c_int(0)
            h = C.c_void_p(self._lib.gpbo_group_ctx(self._g, r))
            self._check(self._lib.gpbo_take_negative_variance_flag(h, C.byref(seen)))
            any_seen = any_seen or bool(seen.value)
        return any_seen

    def member_timings(self, rank: int) -> dict:
        ms = (C.c_float * 8)()
        h = C.c_void_p(self._lib.gpbo_group_ctx(self._g, int(rank)))
        self._check(self._lib.gpbo_last_timings(h, ms, 8))
        return {n: float(ms[i]) for i, n in enumerate(TIMING_NAMES)}

    def shard(self, rank: int) -> tuple[int, int]:
        a, b = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64)
        self._gcheck(self._lib.gpbo_group_shard(self._g, int(rank), iptr(a), iptr(b)))
        return int(a[0]), int(b[0])

    # -- replicated model ----------------------------------------------------------------------
    def fit(self, X, y_norm, kernel: int, length_scale, noise: float, slot: int = 0, precision: int = F64):
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError("X must be 2-D (n_samples, n_features)")
        y_norm = np.ascontiguousarray(y_norm, dtype=np.float64).ravel()
        if y_norm.shape[0] != X.shape[0]:
            raise ValueError("X and y have inconsistent numbers of samples")
        ls = np.ascontiguousarray(np.atleast_1d(np.asarray(length_scale, dtype=np.float64)))
        info = C.c_int(0)
        self._touch(slot)
        rc = self._lib.gpbo_group_fit(self._g, int(slot), dptr(X), dptr(y_norm), X.shape[0], X.shape[1], int(kernel),
                                      dptr(ls), int(ls.shape[0]), float(noise), int(precision), C.byref(info))
        self._gcheck(rc, info.value, borrowed=(X, y_norm, ls, info))
        return self._touch(slot)

    def fit_append(self, x_new, y_norm, slot: int = 0):
        y_norm = np.ascontiguousarray(y_norm, dtype=np.float64).ravel()
        x_new = np.ascontiguousarray(x_new, dtype=np.float64)
        if x_new.ndim != 2:
            raise ValueError("x_new must be 2-D (n_new, n_features)")
        info = C.c_int(0)
        self._touch(slot)
        rc = self._lib.gpbo_group_fit_append(self._g, int(slot), dptr(x_new) if x_new.shape[0] else None, x_new.shape[0],
                                             x_new.shape[1], dptr(y_norm), y_norm.shape[0], C.byref(info))
        self._gcheck(rc, info.value, borrowed=(x_new, y_norm, info))
        return self._touch(slot)

    def lml_batch_arrays(self, X, y_norm, kernel: int, length_scales, noise: float, eval_gradient=True, reuse_inputs=False):
        """The lanes of one lockstep round of the theta search spread over the group's devices (gpbo_group_lml_batch: lane
        i on device i mod G; inputs made resident on every device by the first call of a search).  Same return value as
        GpEngine.lml_batch_arrays, every lane bitwise what `lml()` returns on one device; `last_lane_devices` says where
        each lane ran.  (`lml_batch`, inherited, is built on this: both forms of a round go through the group.)"""
        X = np.ascontiguousarray(X, dtype=np.float64)
        y_norm = np.ascontiguousarray(y_norm, dtype=np.float64).ravel()
        ls = np.ascontiguousarray(np.atleast_2d(np.asarray(length_scales, dtype=np.float64)))
        n, n_ls = ls.shape
        vals = np.zeros(n)
        grads = np.zeros((n, n_ls))
        infos = (C.c_int * n)()
        where = (C.c_int * n)()
        rc = self._lib.gpbo_group_lml_batch(self._g, n, None if reuse_inputs else dptr(X), None if reuse_inputs else dptr(y_norm),
                                            X.shape[0], X.shape[1], int(kernel), dptr(ls), n_ls, float(noise),
                                            int(bool(eval_gradient)), dptr(vals), dptr(grads), infos, where)
        self._gcheck(rc, borrowed=(X, y_norm, ls, vals, grads, infos, where))
        self.last_lane_devices = list(where)
        return vals, grads

    # -- sharded candidates ----------------------------------------------------------------------
    def set_candidates(self, Xc):
        Xc = np.ascontiguousarray(Xc, dtype=np.float64)
        if Xc.ndim != 2:
            raise ValueError("candidates must be 2-D (M, d)")
        n_real = Xc.shape[0]
        if n_real < self.world_size:
            # fewer rows than devices: pad by repeating the last row (an equal value never beats the lower index, and
            # the copies are dropped from every result)
            pad = np.repeat(Xc[-1:], self.world_size - n_real, axis=0)
            Xc = np.ascontiguousarray(np.vstack([Xc, pad]))
        self._gcheck(self._lib.gpbo_group_set_candidates(self._g, dptr(Xc), Xc.shape[0], Xc.shape[1]), borrowed=(Xc,))
        self.n_candidates = n_real
        self._M_pad = Xc.shape[0]
        self._cand_dim = Xc.shape[1]
        self._resident = True

    mixed_device_sampling = False      # a device group samples spaces with int / categorical parameters on the host

    def generate_candidates(self, M: int, lo, hi, seed: int):
        raise NotImplementedError("the Philox throughput generator is per device; a device group draws the reference's "
                                  "stream (device_sampling='auto' / False)")

    def generate_candidates_like(self, M: int, lo, hi, random_state):
        """The reference's candidate matrix from `random_state` (one uniform(lo_j, hi_j, M) per column, in order —
        target_space.py:593-600, parameter.py:86-87): every device generates ITS row block from the caller's MT19937
        state by jump-ahead (gpbo_group_generate_candidates_mt19937); fewer rows than devices: host draw + upload."""
        lo = np.ascontiguousarray(lo, dtype=np.float64).ravel()
        hi = np.ascontiguousarray(hi, dtype=np.float64).ravel()
        if lo.shape != hi.shape:
            raise ValueError("lo and hi must have the same length")
        if not np.all(np.isfinite(hi - lo)):
            raise OverflowError("Range exceeds valid bounds")        # as RandomState.uniform
        n = max(1, int(M))
        if n < self.world_size:
            Xc = np.empty((n, lo.shape[0]))
            for j in range(lo.shape[0]):
                Xc[:, j] = random_state.uniform(lo[j], hi[j], n)
            self.set_candidates(Xc)
            return
        name, key, pos, has_gauss, cached = random_state.get_state(legacy=True)
        if name != "MT19937":
            raise TypeError("generate_candidates_like needs an MT19937 RandomState")
        key = np.ascontiguousarray(key, dtype=np.uint32).copy()
        cpos = C.c_int(int(pos))
        self._gcheck(self._lib.gpbo_group_generate_candidates_mt19937(
            self._g, n, lo.shape[0], dptr(lo), dptr(hi), key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(cpos)),
            borrowed=(lo, hi, key, cpos))
        random_state.set_state((name, key, cpos.value, has_gauss, cached))
        self.n_candidates = n
        self._M_pad = n
        self._cand_dim = lo.shape[0]
        self._resident = True

    def get_candidate_rows(self, idx, d: int):
        idx = np.ascontiguousarray(np.atleast_1d(idx), dtype=np.int64)
        out = np.empty((idx.shape[0], d))
        self._need_resident()
        self._gcheck(self._lib.gpbo_group_get_candidate_rows(self._g, iptr(idx), idx.shape[0], dptr(out)))
        return out

    def _need_resident(self):
        if not self._resident:
            raise _lib.GpboError("the group's candidate shards are not resident (call set_candidates first; a small-batch "
                                 "predict re-uses the first device's candidate buffer)")

    def posterior(self, slot=0, y_mean=0.0, y_std=1.0, fetch=True):
        self._need_resident()
        M = self._M_pad
        mu = np.empty(M) if fetch else None
        sd = np.empty(M) if fetch else None
        self._gcheck(self._lib.gpbo_group_posterior(self._g, int(slot), float(y_mean), float(y_std), dptr(mu), dptr(sd)))
        return (mu[:self.n_candidates], sd[:self.n_candidates]) if fetch else (None, None)

    def predict(self, Xc, slot=0, y_mean=0.0, y_std=1.0, sharded=None):
        """Small batches (the host optimisers' points) run on the first device; from `world_size * 4096` rows on the
        batch is sharded like the random stage."""
        Xc = np.ascontiguousarray(Xc, dtype=np.float64)
        if sharded is None:
            sharded = Xc.shape[0] >= 4096 * self.world_size
        if sharded:
            self.set_candidates(Xc)
            return self.posterior(slot, y_mean, y_std, fetch=True)
        self._resident = False
        M = Xc.shape[0]
        mu, sd = np.empty(M), np.empty(M)
        self._check(self._lib.gpbo_predict(self._h, int(slot), dptr(Xc), M, Xc.shape[1], float(y_mean), float(y_std),
                                           dptr(mu), dptr(sd)))
        return mu, sd

    def acq_argbest(self, acq: int, param: float, y_max: float = 0.0, lb=None, ub=None, k_seeds: int = 0,
                    index_offset: int = 0, return_values: bool = False):
        if index_offset:
            raise ValueError("a device group numbers its candidates globally; index_offset must be 0")
        self._need_resident()
        lb = np.ascontiguousarray(np.atleast_1d(np.asarray(lb, dtype=np.float64))) if lb is not None else None
        ub = np.ascontiguousarray(np.atleast_1d(np.asarray(ub, dtype=np.float64))) if ub is not None else None
        n_c = 0 if lb is None else lb.shape[0]
        if n_c and (ub is None or ub.shape[0] != n_c):
            raise ValueError("lb and ub must have the same length")
        best_idx, best_val = np.zeros(1, dtype=np.int64), np.zeros(1)
        seed_idx = np.full(max(k_seeds, 1), -1, dtype=np.int64)
        seed_val = np.full(max(k_seeds, 1), np.nan)
        ys = np.empty(self._M_pad) if return_values else None
        rc = self._lib.gpbo_group_acq_argbest(self._g, int(acq), float(param), float(y_max if y_max is not None else 0.0),
                                              n_c, dptr(lb), dptr(ub), int(k_seeds), iptr(best_idx), dptr(best_val),
                                              iptr(seed_idx), dptr(seed_val), dptr(ys))
        self._gcheck(rc, borrowed=(lb, ub, best_idx, best_val, seed_idx, seed_val, ys))
        seed_idx, seed_val = seed_idx[:k_seeds], seed_val[:k_seeds]
        if self._M_pad != self.n_candidates:      # drop the padding copies of the last row
            keep = seed_idx < self.n_candidates
            n_keep = int(keep.sum())
            seed_idx = np.concatenate([seed_idx[keep], np.full(k_seeds - n_keep, -1, dtype=np.int64)])
            seed_val = np.concatenate([seed_val[keep], np.full(k_seeds - n_keep, np.nan)])
            ys = ys[:self.n_candidates] if ys is not None else None
        return int(best_idx[0]), float(best_val[0]), seed_idx, seed_val, ys
