"""TEST INFRASTRUCTURE — golden vectors for BASELINE.json's SHARDED configs, from THE REFERENCE ITSELF.

    python -m oracle.gen_golden_shards C4 [first_shard [last_shard]]
    python -m oracle.gen_golden_shards C5 [first_shard [last_shard]]

C4 (d=16, N=4096, EI, 8 x 2^20 candidates) and C5 (d=32, N=8192, constrained EI with a second GP,
8 x 2^18 candidates) are quoted as 8-GPU jobs.  bench.py gives rank r the candidate block
`TargetSpace.random_sample(M_shard, RandomState(7 + r))` (weak scaling: shard r is the same matrix at
any world size), so the job at world size G evaluates the concatenation of shards 0..G-1 and the
reference's answer for it is `argmin / min / argsort[:k]` over the concatenated `ys`
(bayes_opt/acquisition.py:312-317) — rows are evaluated independently (sklearn _gpr.py:443-494), so the
concatenated pass equals the per-shard passes followed by the first-minimum merge.

Every number is produced by the reference's own objects exactly as oracle/gen_golden.py does
(real BayesianOptimization / TargetSpace / wrapped kernel / `_fit_gp` / `_get_acq` / `random_sample`),
one file per shard (tests/golden/<CASE>_s<r>.npz) so a run can be resumed; ~4 min (C4) / ~5 min (C5)
of CPU per shard on 8 cores.  The GPU box has no /root/reference: these committed files are what the
`-m gpu` tests and bench.py's `parity` block compare against.
"""
from __future__ import annotations

import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bayesianoptimization_amd import workloads as W  # noqa: E402
from oracle.gen_golden import CAND_SEED, GOLDEN_DIR, SAMPLE, _chunk_for, build_reference_objects  # noqa: E402

TOPK = 64
N_SHARDS = 8


def generate_shards(name: str, first: int = 0, last: int = N_SHARDS - 1):
    import scipy
    import sklearn

    w = W.ALL[name]
    M = w.M // N_SHARDS
    opt, fn, fit_s = build_reference_objects(w)
    space, gp = opt._space, opt._gp
    acq = fn._get_acq(gp=gp, constraint=space.constraint)
    chunk = _chunk_for(w.N)
    mpath = os.path.join(GOLDEN_DIR, "MANIFEST.json")
    for r in range(first, last + 1):
        t_wall = time.time()
        Xc = space.random_sample(M, np.random.RandomState(CAND_SEED + r))
        assert np.array_equal(Xc, W.make_candidates(w.bounds_array(), M, CAND_SEED + r))
        ys = np.empty(M)
        t0 = time.time()
        for s in range(0, M, chunk):
            ys[s:s + chunk] = acq(Xc[s:s + chunk])
        acq_s = time.time() - t0
        S = min(SAMPLE, M)
        order = np.argsort(ys, kind="stable")[:TOPK].astype(np.int64)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mu, sd = gp.predict(Xc[:S], return_std=True)
            out = {
                "shard": np.int64(r), "seed": np.int64(CAND_SEED + r), "M_evaluated": np.int64(M),
                "length_scale": np.atleast_1d(gp.kernel_.length_scale).astype(np.float64),
                "y_mean": np.float64(gp._y_train_mean), "y_std": np.float64(gp._y_train_std),
                "y_max": np.float64(fn.y_max) if getattr(fn, "y_max", None) is not None else np.float64("nan"),
                "alpha": gp.alpha_.copy(),
                "mu": mu, "sd": sd, "ys": ys[:S].copy(),
                "argmin": np.int64(ys.argmin()), "min": np.float64(ys.min()),
                "topk_idx": order, "topk_val": ys[order].copy(),
                "n_nan": np.int64(np.isnan(ys).sum()),
                # checksum of checksums over the whole shard (compared to 1e-9 relative on the device values)
                "ys_sum": np.float64(ys.sum()), "ys_abs_sum": np.float64(np.abs(ys).sum()),
                "ys_block_min": ys.reshape(-1, 4096).min(axis=1),
            }
            if w.constrained:
                cm = space.constraint._model[0]
                cmu, csd = cm.predict(Xc[:S], return_std=True)
                out.update({
                    "c_length_scale": np.atleast_1d(cm.kernel_.length_scale).astype(np.float64),
                    "c_y_mean": np.float64(cm._y_train_mean), "c_y_std": np.float64(cm._y_train_std),
                    "c_alpha": cm.alpha_.copy(), "c_mu": cmu, "c_sd": csd,
                    "p_c": space.constraint.predict(Xc[:S]),
                })
        np.savez_compressed(os.path.join(GOLDEN_DIR, f"{name}_s{r}.npz"), **out)
        meta = {"N": w.N, "d": w.d, "M_evaluated": int(M), "seed": CAND_SEED + r, "ref_fit_s": round(fit_s, 3),
                "ref_acq_s": round(acq_s, 3), "argmin": int(out["argmin"]), "min": float(out["min"]),
                "top2_gap": float(out["topk_val"][1] - out["topk_val"][0]),
                "gen_wall_s": round(time.time() - t_wall, 1)}
        manifest = json.load(open(mpath)) if os.path.exists(mpath) else {}
        manifest.setdefault("_versions_shards", {"bayes_opt": "3.3.0", "sklearn": sklearn.__version__,
                                                 "scipy": scipy.__version__, "numpy": np.__version__, "topk": TOPK})
        manifest[f"{name}_s{r}"] = meta
        json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)
        print(f"{name}_s{r}", meta, flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    generate_shards(a[0], int(a[1]) if len(a) > 1 else 0, int(a[2]) if len(a) > 2 else N_SHARDS - 1)
