"""Candidate sharding across GPUs (one process per GPU) and the arg-best merge.

The candidate rows are independent (sklearn _gpr.py:443-494 is row-wise; bayes_opt/acquisition.py:312-317
needs only a global argmin and the k best), so the (M,d) matrix is block-partitioned in index order,
every rank fits the same GP redundantly (deterministic -> bit-identical L) and evaluates its block,
and one tiny exchange — an all-gather of (value, global index) records over RCCL/xGMI — precedes an
identical merge on every rank.  No other collective is on the data path.
"""
from __future__ import annotations

import numpy as np


def shard_range(M: int, world_size: int, rank: int) -> tuple[int, int]:
    """Rows [start, stop) owned by `rank`: contiguous blocks, so global index = start + local index."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return (rank * M) // world_size, ((rank + 1) * M) // world_size


def _order(vals: np.ndarray, idxs: np.ndarray) -> np.ndarray:
    """argsort by (value, index) with NaN last and -0.0 == 0.0 — the device selection order."""
    nan = np.isnan(vals)
    key_v = np.where(nan, np.inf, vals) + 0.0
    return np.lexsort((idxs, key_v, nan))


def merge_best(best_vals, best_idxs, seed_vals, seed_idxs, k: int):
    """Combine per-rank results into the global (best_idx, best_val, seed_idx[k], seed_val[k]).

    best_*: (world,) each rank's argmin record (value NaN means "my first NaN is at best_idx", and the
    first NaN overall wins, as numpy argmin).  seed_*: (world, k_local) each rank's sorted k smallest
    (index -1 = padding).  Identical inputs give identical outputs on every rank.
    """
    best_vals = np.asarray(best_vals, dtype=np.float64).ravel()
    best_idxs = np.asarray(best_idxs, dtype=np.int64).ravel()
    nan = np.isnan(best_vals)
    if nan.any():
        bi = int(best_idxs[nan].min())
        bv = float("nan")
    else:
        o = _order(best_vals, best_idxs)[0]
        bi, bv = int(best_idxs[o]), float(best_vals[o])
    sv = np.asarray(seed_vals, dtype=np.float64).ravel()
    si = np.asarray(seed_idxs, dtype=np.int64).ravel()
    keep = si >= 0
    sv, si = sv[keep], si[keep]
    o = _order(sv, si)[:k]
    return bi, bv, si[o], sv[o]


class ShardedAcquisition:
    """Runs fit + posterior + acquisition on this rank's shard and merges the arg-best across ranks.

    Transport: by default the engine's RCCL communicator — `engine.comm_acq_argbest` packs the shard's records on the
    device, all-gathers them with ncclAllGather on the engine's stream and merges them in the library (the same merge
    as `merge_best`).  `allgather(vals, idxs) -> (all_vals, all_idxs)` substitutes any host collective (the gloo tests).
    """

    def __init__(self, engine, world_size: int = 1, rank: int = 0, allgather=None):
        self.engine = engine
        self.world_size = int(world_size)
        self.rank = int(rank)
        self.allgather = allgather

    def set_candidates_global(self, Xc_global: np.ndarray):
        """Keep this rank's block of a global candidate matrix resident."""
        s, e = shard_range(Xc_global.shape[0], self.world_size, self.rank)
        self.offset = s
        self.engine.set_candidates(Xc_global[s:e])

    def generate_candidates_like(self, M_global: int, lo, hi, random_state):
        """This rank's block of `space.random_sample(M_global, random_state)` for an all-float space, generated on this rank's
        GPU from the ONE reference stream (`GpEngine.generate_candidate_rows_like`): every rank calls this with a RandomState
        in the same state, nobody draws on the host, nothing is uploaded, and every rank's RandomState ends where the
        reference's would — the one-process-per-GPU counterpart of `GroupEngine.generate_candidates_like`."""
        s, e = shard_range(int(M_global), self.world_size, self.rank)
        self.offset = s
        self.engine.generate_candidate_rows_like(int(M_global), lo, hi, random_state, s, e)

    def set_candidates_local(self, Xc_local: np.ndarray, offset: int):
        self.offset = int(offset)
        self.engine.set_candidates(Xc_local)

    def argbest(self, acq, param, y_max=0.0, lb=None, ub=None, k_seeds: int = 0):
        if self.world_size > 1 and self.allgather is None:       # device records -> RCCL all-gather -> merge
            bi, bv, si, sv, _ = self.engine.comm_acq_argbest(acq, param, y_max, lb, ub, k_seeds=k_seeds,
                                                             index_offset=self.offset)
            return bi, bv, si, sv
        bi, bv, si, sv, _ = self.engine.acq_argbest(acq, param, y_max, lb, ub, k_seeds=k_seeds,
                                                    index_offset=self.offset)
        if self.world_size == 1:
            return bi, bv, si, sv
        vals = np.concatenate([[bv], sv]).astype(np.float64)
        idxs = np.concatenate([[bi], si]).astype(np.int64)
        av, ai = self.allgather(vals, idxs)
        av = np.asarray(av).reshape(self.world_size, 1 + k_seeds)
        ai = np.asarray(ai).reshape(self.world_size, 1 + k_seeds)
        return merge_best(av[:, 0], ai[:, 0], av[:, 1:], ai[:, 1:], k_seeds)
