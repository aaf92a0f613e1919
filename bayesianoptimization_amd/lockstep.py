"""Lockstep execution of independent host optimiser runs over one batched device objective.

Two places in the reference run several independent SciPy optimisations one after another over an expensive objective:
the acquisition's local searches from the best random candidates (bayes_opt/acquisition.py:364-374) and scikit-learn's
theta search with restarts (sklearn/gaussian_process/_gpr.py:296-338).  On the device a batch of evaluations costs little
more than one, so the runs are advanced together (fused_acquisition._polish_in_lockstep, gpr.HipGPR._theta_search_lockstep).
"""
from __future__ import annotations

import threading

import numpy as np


class Lockstep:
    """Merges the objective evaluations of several independent optimiser runs into shared device batches.

    Every run lives in its own thread; a run that needs function values parks its points here (`ask`), and once ALL
    live runs are parked the thread that called `serve()` evaluates the concatenation with a single `evaluate(batch)`
    call and hands each run its rows.  The optimiser code and the per-point arithmetic are untouched, so every run
    visits the iterates it would visit alone; only the number of device round trips drops (by the number of runs).
    All device work stays on the serving thread.  `evaluate` maps an (n, ...) array to an array with n rows."""

    class Abort(Exception):
        pass

    def __init__(self, evaluate, n_runs: int) -> None:
        self._evaluate = evaluate
        self._cv = threading.Condition()
        self._parked: dict[int, np.ndarray] = {}
        self._answers: dict[int, np.ndarray] = {}
        self._live = n_runs
        self._failure: BaseException | None = None
        self.batches = 0

    def ask(self, run: int, pts):
        with self._cv:
            self._parked[run] = np.array(pts, dtype=np.float64, copy=True)
            self._cv.notify_all()
            while run not in self._answers and self._failure is None:
                self._cv.wait()
            if self._failure is not None:
                raise Lockstep.Abort
            return self._answers.pop(run)

    def retire(self, run: int) -> None:
        with self._cv:
            self._live -= 1
            self._cv.notify_all()

    def serve(self) -> None:
        with self._cv:
            while True:
                while self._live > 0 and len(self._parked) < self._live:
                    self._cv.wait()
                if self._live == 0:
                    return
                order = sorted(self._parked)
                sizes = [len(self._parked[r]) for r in order]
                try:
                    values = np.asarray(self._evaluate(np.concatenate([self._parked[r] for r in order])),
                                        dtype=np.float64)
                except BaseException as exc:   # wake the runs so that their threads end, then re-raise here
                    self._failure = exc
                    self._parked.clear()
                    self._cv.notify_all()
                    raise
                self.batches += 1
                for r, chunk in zip(order, np.split(values, np.cumsum(sizes)[:-1])):
                    self._answers[r] = chunk
                self._parked.clear()
                self._cv.notify_all()
