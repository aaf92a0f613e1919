"""Property-based checks (hypothesis) of the host-side pieces whose contracts are pure functions:
the shard/merge pair behind the multi-GPU arg-best (any sharding must return what one pass returns,
NumPy's NaN-first / first-index tie rules included), the finite-difference point generator behind the lockstep
L-BFGS-B driver (must equal SciPy's approx_derivative steps), and the MT19937 block walk (any stream position, any
shape)."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from bayesianoptimization_amd import lbfgsb_lockstep as LL
from bayesianoptimization_amd.distributed import merge_best, shard_range


def _single_pass(ys, k):
    """What acquisition.py:313-317 computes on the whole array, with the device's documented tie rule for the seeds."""
    nan = np.isnan(ys)
    if nan.any():
        best = (int(np.flatnonzero(nan)[0]), float("nan"))
    else:
        best = (int(ys.argmin()), float(ys.min()))
    order = np.lexsort((np.arange(len(ys)), np.where(nan, np.inf, ys) + 0.0, nan))[:k]
    return best, order, ys[order]


values = st.one_of(st.floats(min_value=-5, max_value=5, allow_nan=False), st.just(float("nan")), st.just(-0.0),
                   st.sampled_from([0.0, 1.0, -1.0]))     # plenty of exact ties, signed zeros and NaNs


@settings(max_examples=200, deadline=None)
@given(ys=st.lists(values, min_size=1, max_size=200), world=st.integers(1, 9), k=st.integers(0, 12))
def test_sharded_argbest_equals_single_pass(ys, world, k):
    ys = np.array(ys, dtype=np.float64)
    M = len(ys)
    bv, bi, sv, si = [], [], [], []
    for r in range(world):
        s, e = shard_range(M, world, r)
        if e == s:                      # an empty shard contributes a +inf record, as an idle rank does
            bv.append(np.inf); bi.append(np.iinfo(np.int64).max); sv.append(np.full(k, np.nan)); si.append(np.full(k, -1))
            continue
        (b_i, b_v), order, vals = _single_pass(ys[s:e], k)
        bv.append(b_v); bi.append(b_i + s)
        pad = k - len(order)
        sv.append(np.concatenate([vals, np.full(pad, np.nan)])); si.append(np.concatenate([order + s, np.full(pad, -1)]))
    got = merge_best(bv, bi, np.array(sv).reshape(world, -1), np.array(si).reshape(world, -1), k)
    (b_i, b_v), order, vals = _single_pass(ys, k)
    assert got[0] == b_i and (got[1] == b_v or (np.isnan(got[1]) and np.isnan(b_v)))
    assert np.array_equal(got[2], order)
    assert np.array_equal(got[3], vals, equal_nan=True)


@settings(max_examples=100, deadline=None)
@given(M=st.integers(0, 10_000), world=st.integers(1, 64))
def test_shard_range_is_a_contiguous_partition(M, world):
    edges = [shard_range(M, world, r) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == M
    assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    sizes = [e - s for s, e in edges]
    assert max(sizes) - min(sizes) <= 1


bound = st.one_of(st.floats(-10, 10, allow_nan=False), st.just(-np.inf), st.just(np.inf))


@settings(max_examples=150, deadline=None)
@given(data=st.data(), d=st.integers(1, 5))
def test_forward_difference_points_equal_scipys(data, d):
    """Against scipy.optimize._numdiff.approx_derivative itself (it records the points it evaluates)."""
    from scipy.optimize._numdiff import approx_derivative

    lo, hi = [], []
    for _ in range(d):
        a, b = data.draw(bound), data.draw(bound)
        a, b = (a, b) if a <= b else (b, a)
        if a == b or (np.isinf(a) and a > 0) or (np.isinf(b) and b < 0):
            a, b = -1.0, 1.0
        lo.append(a); hi.append(b)
    lo, hi = np.array(lo), np.array(hi)
    x0 = np.array([data.draw(st.floats(max(l, -50.0), min(h, 50.0), allow_nan=False)) for l, h in zip(lo, hi)])
    seen = []

    def f(x):
        seen.append(np.array(x, dtype=np.float64))
        return float(np.sum(x))

    f0 = f(x0)
    approx_derivative(f, x0, method="2-point", abs_step=1e-8, f0=f0, bounds=(lo, hi))
    pts, steps = LL.forward_difference_points(x0[None], lo, hi)
    assert np.array_equal(pts[0], np.array(seen))
    assert np.array_equal(steps[0], np.array([seen[t + 1][t] - x0[t] for t in range(d)]))


@settings(max_examples=40, deadline=None)
@given(M=st.integers(1, 700), d=st.integers(1, 6), burn=st.integers(0, 1500), seed=st.integers(0, 2**31 - 1))
def test_mt19937_block_walk_any_position_any_shape(M, d, burn, seed):
    from helpers import mt19937_device_mirror

    ref = np.random.RandomState(seed)
    if burn:
        ref.random_sample(burn // 2)
        if burn % 2:
            ref.randint(0, 2, size=1, dtype=np.uint32) if False else ref.bytes(4)     # one 32-bit output -> odd position
    _, key, pos, hg, cg = ref.get_state()
    lo = np.linspace(-1.0, 2.0, d)
    hi = lo + 1.5
    Xc, key2, pos2 = mt19937_device_mirror(key, pos, M, d, lo, hi)
    assert np.array_equal(Xc, np.column_stack([ref.uniform(lo[t], hi[t], M) for t in range(d)]))
    cont = np.random.RandomState(0)
    cont.set_state(("MT19937", key2, pos2, hg, cg))
    assert np.array_equal(cont.random_sample(5), ref.random_sample(5))
