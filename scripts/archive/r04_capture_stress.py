"""Stress of the hipGraph capture inside gpbo_lml_batch when several contexts capture at the same time (the theta search of
a device group: one thread per device, here three virtual ranks on the one GPU).  Usage:
    python scripts/archive/r04_capture_stress.py <libgpbo.so> <label> [n_fits]
Runs n_fits theta searches (N = 2100, d = 16: one stream + one graph per lane; and N = 300: one graph for all lanes) through
a GroupEngine([0, 0, 0]) and counts the calls that raised; a build with -DGPBO_CAPTURE_TRACE reports failed captures that were
recovered on stderr (counted by the caller, scripts/archive/r04_capture_stress.sh)."""
import json
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, ".")
from bayesianoptimization_amd import _lib  # noqa: E402

path, label = sys.argv[1], sys.argv[2]
n_fits = int(sys.argv[3]) if len(sys.argv) > 3 else 20
_lib._lib = _lib._bind(path, _lib.SIGNATURES)

from sklearn.gaussian_process.kernels import Matern  # noqa: E402

from bayesianoptimization_amd.engine import GroupEngine  # noqa: E402
from bayesianoptimization_amd.gpr import HipGPR  # noqa: E402

warnings.simplefilter("ignore")
out = {"label": label, "lib": path, "cases": []}
for N, d in ((2100, 16), (300, 5)):
    rng = np.random.RandomState(N)
    X = rng.uniform(size=(N, d))
    y = np.exp(-((X - 0.5) ** 2).sum(1)) + 0.01 * rng.standard_normal(N)
    raised, thetas, t0 = 0, [], time.perf_counter()
    grp = GroupEngine([0, 0, 0])
    for i in range(n_fits):
        try:
            gp = HipGPR(kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5,
                        random_state=np.random.RandomState(3), engine=grp, lml_on_device=True).fit(X, y)
            thetas.append(float(gp.kernel_.theta[0]))
        except Exception as e:  # noqa: BLE001
            raised += 1
            print(f"[{label}] N={N} fit {i}: {type(e).__name__}: {str(e)[:200]}", file=sys.stderr)
            grp.close()
            grp = GroupEngine([0, 0, 0])
    grp.close()
    out["cases"].append({"N": N, "d": d, "fits": n_fits, "raised": raised, "distinct_theta": len(set(thetas)),
                         "s_per_fit": (time.perf_counter() - t0) / n_fits})
# Phase 2, the sharp case: every iteration switches the kernel (the cached graphs no longer fit), evaluates once directly and
# then AGAIN with the inputs handed over (upload on every rank) — the call that captures, on three threads at once, while the
# other ranks may still be uploading.
from bayesianoptimization_amd import engine as O  # noqa: E402  (kernel ids)

rng = np.random.RandomState(5)
X = rng.uniform(size=(2100, 16))
yn = rng.standard_normal(2100)
scales = np.array([[0.5], [0.8], [1.0], [1.5], [2.0], [3.0]])
n_it = 8 * n_fits
raised, mismatch, t0 = 0, 0, time.perf_counter()
grp = GroupEngine([0, 0, 0])
want = {}
for i in range(n_it):
    kind = (O.MATERN25, O.RBF)[i & 1]
    try:
        a = grp.lml_batch(X, yn, kind, scales, 1e-6)
        b = grp.lml_batch(X, yn, kind, scales, 1e-6)
        c = grp.lml_batch(X, yn, kind, scales, 1e-6, reuse_inputs=True)
        want.setdefault(kind, a)
        for got in (a, b, c):
            for (v, g), (v1, g1) in zip(got, want[kind]):
                mismatch += int(not (v == v1 and np.array_equal(g, g1)))
    except Exception as e:  # noqa: BLE001
        raised += 1
        print(f"[{label}] recapture iteration {i}: {type(e).__name__}: {str(e)[:200]}", file=sys.stderr)
        grp.close()
        grp = GroupEngine([0, 0, 0])
grp.close()
out["recapture"] = {"iterations": n_it, "raised": raised, "lanes_not_bitwise": mismatch, "ms_per_iteration": 1e3 * (time.perf_counter() - t0) / n_it}
print(json.dumps(out))
