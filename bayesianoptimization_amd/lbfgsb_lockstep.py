"""Several SciPy L-BFGS-B minimisations advanced together over ONE batched objective, on one thread.

The reference polishes the best random candidates one after another with
`scipy.optimize.minimize(acq, x_try, bounds=..., method="L-BFGS-B")` (bayes_opt/acquisition.py:364-374); every function
value there is one `gp.predict` call.  SciPy's L-BFGS-B is a reverse-communication routine: `setulb(...)` returns whenever
it needs f and g at the current x.  This module drives that routine itself for all seeds: each round it steps every live
run up to its next request, evaluates ALL requested points (and their finite-difference neighbours) with one call of the
batched objective — one device launch — and hands the values back.  The optimiser arithmetic is SciPy's own compiled
`setulb`, called with exactly the arguments `scipy.optimize._lbfgsb_py._minimize_lbfgsb` passes (same work arrays, `factr`,
`pgtol`, `maxcor`, `maxls`, iteration/evaluation limits), and the forward differences are the ones `approx_derivative`
forms (see `forward_difference_points`), so every run visits the iterates `minimize` would visit and ends in the same
`OptimizeResult` fields.  What disappears is Python: `ScalarFunction`, `MemoizeJac`, bound standardisation and (compared
with `lockstep.Lockstep`) one thread per run.

`setulb` is private SciPy API.  `driver_available()` checks the module layout and the version this was written
against (1.15.x); anything else falls back to the thread-based `lockstep.Lockstep`, which only uses public API.
tests/test_host_logic.py::test_lbfgsb_driver_equals_scipy_minimize pins the equivalence bit for bit.
"""
from __future__ import annotations

import contextlib
import threading

import numpy as np
from scipy.optimize import OptimizeResult

_FD_EPS = 1e-8                              # _minimize_lbfgsb: eps=1e-8 -> approx_derivative(abs_step=1e-8)
_SQRT_EPS = np.finfo(np.float64).eps ** 0.5

_TASK_FG, _TASK_NEW_X, _TASK_CONVERGENCE, _TASK_STOP = 3, 1, 4, 5


def _setulb():
    try:
        import scipy
        from scipy.optimize import _lbfgsb_py
    except Exception:   # pragma: no cover
        return None
    major_minor = tuple(int(p) for p in scipy.__version__.split(".")[:2])
    fn = getattr(getattr(_lbfgsb_py, "_lbfgsb", None), "setulb", None)
    doc = getattr(fn, "__doc__", "") or ""
    if fn is None or major_minor != (1, 15) or "ln_task" not in doc:
        return None
    return fn


_SELF_CHECK: bool | None = None
_BLAS_CONTROLLER = None


_BLAS_LOCK = threading.Lock()
_BLAS_USERS = 0
_BLAS_LIMIT = None


class _BlasSingleThread:
    """Re-entrant and thread-safe: the limit is set by the first user to enter and lifted by the last one to leave (a module
    lock + a count).  Without that, two optimizers in two threads could interleave — A saves 256 and sets 1, B saves 1, A
    restores 256, B restores 1 — and leave the process's BLAS at one thread for good."""

    def __enter__(self):
        global _BLAS_USERS, _BLAS_LIMIT, _BLAS_CONTROLLER
        with _BLAS_LOCK:
            _BLAS_USERS += 1
            if _BLAS_USERS == 1:
                try:
                    from threadpoolctl import ThreadpoolController

                    if _BLAS_CONTROLLER is None:
                        _BLAS_CONTROLLER = ThreadpoolController()
                    _BLAS_LIMIT = _BLAS_CONTROLLER.limit(limits=1, user_api="blas")
                    _BLAS_LIMIT.__enter__()
                except Exception:   # noqa: BLE001  (threadpoolctl is a scikit-learn dependency; without it nothing changes)
                    _BLAS_LIMIT = None
        return self

    def __exit__(self, *exc):
        global _BLAS_USERS, _BLAS_LIMIT
        with _BLAS_LOCK:
            _BLAS_USERS -= 1
            if _BLAS_USERS == 0 and _BLAS_LIMIT is not None:
                limit, _BLAS_LIMIT = _BLAS_LIMIT, None
                limit.__exit__(None, None, None)
        return False


def blas_single_thread():
    """Context manager: the process's BLAS pools limited to one thread while L-BFGS-B's bookkeeping runs.

    `setulb` does its linear algebra — Cholesky factors and triangular solves of 2m x 2m matrices, m = 10 — through
    BLAS / LAPACK.  With OpenBLAS's default pool (one thread per host core: 256 on the GPU box) every such call pays the
    pool's wake-up and hand-off for a few hundred flops: measured here (8 threads) 4.9-7.4 ms per theta search against
    2.8-3.0 ms with the pool limited to one thread, the objective memoised in both (the GPU box's host LML fit took 96 ms
    at N = 128 for 13 ms of LML evaluations, profiles/r04_lml_crossover.json).  Only for drivers whose objective does not
    itself need the host's BLAS (device evaluations): the limit is process-wide while ANY such driver runs (other threads'
    host BLAS is throttled meanwhile) and is restored when the last one leaves.  Same bits either way —
    operands this small never reach a threaded kernel's split (tests/test_host_logic.py pins the driver against
    scipy.optimize.minimize run WITHOUT the limit)."""
    return _BlasSingleThread()


def _self_check() -> bool:
    """`setulb` is private SciPy API driven with hand-built work arrays: before the first real use, run the driver on a
    small bound-constrained problem and require the result of the public `scipy.optimize.minimize` — iterates, counts
    and status — bit for bit.  A SciPy patch release that changed array sizes or task codes fails here (and the callers
    fall back to the thread-based lockstep over the public API) instead of corrupting memory or diverging silently."""
    try:
        from scipy.optimize import minimize

        def fg(x):
            x = np.asarray(x, dtype=np.float64)
            r = x - np.array([0.3, -1.7, 2.5])
            return float(r @ r + 0.1 * np.sum(x**4)), 2 * r + 0.4 * x**3

        box = np.array([[-1.0, 1.0], [-1.0, 2.0], [0.0, 2.0]])
        starts = [np.array([0.9, 1.5, 0.1]), np.array([-0.5, -0.5, 1.0])]

        def batch(X):
            vals = [fg(x) for x in X]
            return np.array([v[0] for v in vals]), np.array([v[1] for v in vals])

        mine = _drive(batch, 1, starts, box, 10, 2.2204460492503131e-09, 1e-5, 15000, 15000, 20)
        for r, s0 in zip(mine, starts):
            ref = minimize(fg, s0, jac=True, bounds=box, method="L-BFGS-B")
            if not (np.array_equal(r.x, ref.x) and r.fun == ref.fun and r.nit == ref.nit and r.nfev == ref.nfev
                    and r.status == ref.status):
                return False
        return True
    except Exception:   # noqa: BLE001  (any surprise from a private API = not available)
        return False


def driver_available() -> bool:
    global _SELF_CHECK
    if _setulb() is None:
        return False
    if _SELF_CHECK is None:
        _SELF_CHECK = _self_check()
    return _SELF_CHECK


def forward_difference_points(X0, lb, ub):
    """For every row x0 of X0 (S, d): the d + 1 points at which SciPy evaluates the objective to form L-BFGS-B's
    gradient when `jac` is absent — x0 and x0 + h_t e_t with `approx_derivative(method="2-point", abs_step=1e-8,
    bounds=(lb, ub))`'s steps, including `_adjust_scheme_to_bounds(..., "1-sided")` (scipy/optimize/_numdiff.py).
    Returns (pts (S, d + 1, d), steps (S, d)) with steps = the ACTUAL differences (x0_t + h_t) - x0_t, as
    `_dense_difference` uses them."""
    X0 = np.asarray(X0, dtype=np.float64)
    sign = (X0 >= 0).astype(np.float64) * 2 - 1
    h = np.full_like(X0, _FD_EPS)
    h = np.where((X0 + h) - X0 == 0, _SQRT_EPS * sign * np.maximum(1.0, np.abs(X0)), h)
    if not np.all((lb == -np.inf) & (ub == np.inf)):
        below, above = X0 - lb, ub - X0
        trial = X0 + h
        outside = (trial < lb) | (trial > ub)
        fits = np.abs(h) <= np.maximum(below, above)
        h = np.where(outside & fits, -h, h)
        h = np.where((above >= below) & ~fits, above, h)
        h = np.where((above < below) & ~fits, -below, h)
    S, d = X0.shape
    pts = np.repeat(X0[:, None, :], d + 1, axis=1)
    idx = np.arange(d)
    pts[:, idx + 1, idx] = X0 + h
    steps = pts[:, idx + 1, idx] - X0
    return pts, steps


class _Run:
    __slots__ = ("x", "f", "g", "wa", "iwa", "task", "ln_task", "lsave", "isave", "dsave", "nit", "nfev", "done",
                 "x_seen", "f_seen", "g_seen")

    def __init__(self, x0, n, m):
        self.x = np.array(x0, dtype=np.float64)
        self.f = np.array(0.0, dtype=np.int32)        # as _minimize_lbfgsb initialises them
        self.g = np.zeros((n,), dtype=np.int32)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32)
        self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32)
        self.isave = np.zeros(44, dtype=np.int32)
        self.dsave = np.zeros(29, dtype=np.float64)
        self.nit = 0
        self.nfev = 0
        self.done = False
        self.x_seen = None       # the point of the last evaluation and its (f, g): ScalarFunction's one-entry cache
        self.f_seen = self.g_seen = None


def _messages():
    try:
        from scipy.optimize._lbfgsb_py import status_messages, task_messages
        return status_messages, task_messages
    except Exception:   # pragma: no cover
        return {}, {}


def _drive(evaluate, evals_per_request, starts, box, maxcor, ftol, gtol, maxfun, maxiter, maxls, single_thread_blas=False):
    """The loop of `_minimize_lbfgsb` for all starts at once.  evaluate(X (S, n)) -> (f (S,), g (S, n)) for the S runs
    that asked this round; evals_per_request = what ScalarFunction adds to nfev per evaluated point.
    single_thread_blas: see blas_single_thread() — for objectives evaluated on the device."""
    if single_thread_blas:
        with blas_single_thread():
            return _drive(evaluate, evals_per_request, starts, box, maxcor, ftol, gtol, maxfun, maxiter, maxls)
    setulb = _setulb()
    if setulb is None:
        raise RuntimeError("scipy's L-BFGS-B reverse-communication routine is not available in the expected form")
    box = np.asarray(box, dtype=np.float64)
    lb, ub = box[:, 0].copy(), box[:, 1].copy()
    n = box.shape[0]
    if np.any(lb > ub):
        raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
    if np.any(lb == ub):
        raise ValueError("fixed variables (lb == ub) are not handled here; use scipy.optimize.minimize")
    factr = ftol / np.finfo(float).eps
    low_bnd, upper_bnd = np.zeros(n), np.zeros(n)
    nbd = np.zeros(n, np.int32)
    for t in range(n):
        has_l, has_u = not np.isinf(lb[t]), not np.isinf(ub[t])
        if has_l:
            low_bnd[t] = lb[t]
        if has_u:
            upper_bnd[t] = ub[t]
        nbd[t] = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}[has_l, has_u]

    runs = [_Run(np.clip(np.asarray(s, dtype=np.float64).ravel(), lb, ub), n, maxcor) for s in starts]

    def step(run):
        """Advance one run to its next f/g request at a NEW point (returns True) or to its end (False)."""
        while True:
            if run.g.dtype != np.float64:       # (_minimize_lbfgsb: g = g.astype(np.float64) — only the initial int32 zeros need it)
                run.g = run.g.astype(np.float64)
            setulb(maxcor, run.x, low_bnd, upper_bnd, nbd, run.f, run.g, factr, gtol, run.wa, run.iwa, run.task,
                   run.lsave, run.isave, run.dsave, maxls, run.ln_task)
            task = int(run.task[0])
            if task == _TASK_FG:
                # ScalarFunction answers a repeated x from its cache (np.array_equal: elementwise ==, so a NaN never matches)
                if run.x_seen is not None and bool((run.x == run.x_seen).all()):
                    run.f, run.g = run.f_seen, run.g_seen
                    continue
                return True
            if task == _TASK_NEW_X:
                run.nit += 1
                if run.nit >= maxiter:
                    run.task[0], run.task[1] = _TASK_STOP, 504
                elif run.nfev > maxfun:
                    run.task[0], run.task[1] = _TASK_STOP, 502
                continue
            run.done = True
            return False

    live = [r for r in runs if step(r)]
    while live:
        fs, gs = evaluate(np.array([r.x for r in live]))
        for r, v, g in zip(live, fs, gs):
            r.f, r.g = v, g
            r.x_seen, r.f_seen, r.g_seen = r.x.copy(), v, g
            r.nfev += evals_per_request
        live = [r for r in live if step(r)]

    status_messages, task_messages = _messages()
    out = []
    for r in runs:
        if r.task[0] == _TASK_CONVERGENCE:
            status = 0
        elif r.nfev > maxfun or r.nit >= maxiter:
            status = 1
        else:
            status = 2
        message = f"{status_messages.get(int(r.task[0]), r.task[0])}: {task_messages.get(int(r.task[1]), r.task[1])}"
        out.append(OptimizeResult(x=r.x, fun=r.f, jac=r.g, nit=r.nit, nfev=r.nfev, status=status, message=message,
                                  success=status == 0))
    return out


def minimize_many(acq, starts, box, maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxfun=15000, maxiter=15000,
                  maxls=20, single_thread_blas=False):
    """[OptimizeResult(x, fun, jac, nit, nfev, status, message, success)] of
    `scipy.optimize.minimize(acq_single, start, bounds=box, method="L-BFGS-B")` (no `jac`: forward differences) for
    every start, where `acq` maps a batch of points (P, d) to their P values and `acq_single(x) = acq(x[None])[0]`."""
    box = np.asarray(box, dtype=np.float64)
    lb, ub = box[:, 0], box[:, 1]
    n = box.shape[0]

    def evaluate(X0):
        pts, steps = forward_difference_points(X0, lb, ub)
        vals = np.asarray(acq(pts.reshape(-1, n)), dtype=np.float64).reshape(len(X0), n + 1)
        return vals[:, 0], (vals[:, 1:] - vals[:, :1]) / steps

    # ScalarFunction counts the point and its d finite-difference neighbours
    return _drive(evaluate, n + 1, starts, box, maxcor, ftol, gtol, maxfun, maxiter, maxls, single_thread_blas)


def minimize_many_with_grad(value_and_grad, starts, box, maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5,
                            maxfun=15000, maxiter=15000, maxls=20, single_thread_blas=False):
    """The same for an objective that returns its gradient: [OptimizeResult] of
    `scipy.optimize.minimize(fg_single, start, jac=True, bounds=box, method="L-BFGS-B")` for every start, where
    `value_and_grad` maps a batch X (S, n) to (f (S,), g (S, n)) — scikit-learn's theta search with restarts
    (sklearn/gaussian_process/_gpr.py:296-338, 656-668) over `gpbo_lml_batch`."""
    def evaluate(X):
        f, g = value_and_grad(X)
        return np.asarray(f, dtype=np.float64), np.asarray(g, dtype=np.float64).reshape(len(X), -1)

    return _drive(evaluate, 1, starts, box, maxcor, ftol, gtol, maxfun, maxiter, maxls, single_thread_blas)
