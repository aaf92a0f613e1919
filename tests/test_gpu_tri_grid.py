"""The lower-only fit GEMMs on a 1-D grid of their live tiles (launch_gemm's tri_grid, csrc/fit_kernels.hip) against the square grid
they ran on until round 5: the same tiles with the same arithmetic, so every result is BITWISE the same — the Cholesky factor (its
rank-`outer` trailing updates are lower-only SYRK-shaped products), the LML value and its gradient (K^-1 = W^T W) — on both GEMM
kernels (64 x 64 tiles: N = 2111; 128 x 128 tiles: the trailing updates at N = 4096).  Debug build: GPBO_TRI_GRID is read per launch.
Replaces in the reference: dpotrf's trailing updates (sklearn _gpr.py:349) and the K^-1 of the LML gradient (_gpr.py:627-629)."""
import os

import numpy as np
import pytest

from tests import helpers as H  # noqa: F401  (path set-up)
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,d", [(2111, 5), (4096, 16)])
def test_live_tile_grid_is_bitwise_the_square_grid(debug_engine, N, d):
    rng = np.random.RandomState(N)
    X = rng.uniform(size=(N, d))
    y = np.exp(-((X - 0.5) ** 2).sum(1)) + 0.01 * rng.standard_normal(N)
    yn = (y - y.mean()) / y.std()
    got = {}
    old = os.environ.get("GPBO_TRI_GRID")
    try:
        for setting in ("0", "1"):
            os.environ["GPBO_TRI_GRID"] = setting
            debug_engine.fit(X, yn, O.MATERN25, [0.8], 1e-6)
            got[setting] = (debug_engine.get_L(N), debug_engine.get_alpha(N), debug_engine.lml(X, yn, O.MATERN25, [0.8], 1e-6))
    finally:
        if old is None:
            os.environ.pop("GPBO_TRI_GRID", None)
        else:
            os.environ["GPBO_TRI_GRID"] = old
    a, b = got["0"], got["1"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2][0] == b[2][0] and np.array_equal(a[2][1], b[2][1])
    if N <= 2111:
        v, g = O.log_marginal_likelihood(O.MATERN25, X, yn, [0.8], 1e-6)
        assert abs(b[2][0] - v) <= 1e-10 * max(1.0, abs(v)) and np.max(np.abs(b[2][1] - g)) <= 1e-7 * max(1.0, float(np.max(np.abs(g))))
