"""CPU: NumPy models of two device algorithms whose correctness rests on an argument rather than on arithmetic — run against
brute force here, so that the argument is checked where there is no GPU (the kernels themselves are checked bit for bit on the
GPU: tests/test_gpu_select.py, tests/test_gpu_seams.py).

1. Selection by threshold + rank counting (csrc/acq_kernels.hip, select_*_v2_kernel; replaces the k block reductions of
   ys.argmin() / argsort(ys)[:k], bayes_opt/acquisition.py:313-317).  The claim: with every thread's smallest key ranked inside
   its wave, tau = the least of the waves' k-th smallest minima is an upper bound of the workgroup's k-th smallest key, at
   least min(k, #keys) keys are <= tau, and ranking the keys <= tau among themselves gives the first k of the total order
   (value, index), -0.0 == 0.0, NaN last.  A list longer than its LDS capacity falls back to the k passes.
2. The MT19937 jump split (csrc/mt_jump.hip, mt_jump_part_kernel): the window XOR_{b in g} x_{b+t} computed in 8 ranges of
   bit positions, each from a slice of the sequence and byte offsets relative to the range, two adjacent words per thread."""
import struct

import numpy as np
import pytest

SEL_BLOCK, SEL_ITEMS = 256, 16
U64, I64 = (1 << 64) - 1, (1 << 63) - 1
SENTINEL = (U64, I64)


def _order_bits(v):
    if v != v:
        return U64
    b = struct.unpack("<Q", struct.pack("<d", 0.0 if v == 0.0 else v))[0]
    return (~b) & U64 if (b >> 63) else (b | (1 << 63))


def _threshold(mins, k):
    bounds = []
    for w in range(SEL_BLOCK // 64):
        lane = mins[w * 64:(w + 1) * 64]
        bound = SENTINEL
        for t in range(64):
            if lane[t][1] != I64 and sum(1 for j in range(64) if lane[j] < lane[t]) == k - 1:
                bound = lane[t]
        bounds.append(bound)
    return min(bounds)


def _workgroup(items_per_thread, k, cap, listed):
    mins = [min(items, default=SENTINEL) for items in items_per_thread]
    tau = _threshold(mins, k)
    keys = [c for items in items_per_thread for c in items if c[1] != I64]
    lst = [c for c in keys if not tau < c]
    listed.append(len(lst))
    assert len(lst) >= min(k, len(keys))                    # the claim the kernel's emit step relies on
    if len(lst) > cap:
        lst = keys                                          # the fall-back: k passes over everything
    out = [SENTINEL] * k
    for c in lst:
        r = sum(1 for q in lst if q < c)
        if r < k:
            out[r] = c
    return out


def _select_model(ys, k, cap=1024):
    M = len(ys)
    keys = [(_order_bits(float(v)), i) for i, v in enumerate(ys)]
    per_block = SEL_BLOCK * SEL_ITEMS
    partial, listed = [], []
    for b in range((M + per_block - 1) // per_block):
        ipt = [[keys[b * per_block + t + it * SEL_BLOCK] for it in range(SEL_ITEMS) if b * per_block + t + it * SEL_BLOCK < M]
               for t in range(SEL_BLOCK)]
        partial += _workgroup(ipt, k, cap, listed)
    ipt = [[partial[j] for j in range(t, len(partial), SEL_BLOCK)] for t in range(SEL_BLOCK)]
    picks = _workgroup(ipt, k, cap, listed)
    return [(-1 if c[1] == I64 else c[1]) for c in picks], listed


def _numpy_order(ys, k):
    ys = np.asarray(ys, dtype=np.float64)
    nan = np.isnan(ys)
    order = np.lexsort((np.arange(ys.shape[0]), np.where(nan, np.inf, np.where(ys == 0.0, 0.0, ys)), nan))[:k]
    return list(order) + [-1] * (k - len(order))


def _selection_cases():
    rng = np.random.RandomState(0)
    M = 9000
    out = [(f"random_{m}", rng.randn(m)) for m in (1, 5, 64, 257, 4097)]
    out.append(("all_equal", np.zeros(M)))
    out.append(("descending", -np.arange(M, dtype=np.float64)))
    a = rng.randn(M); a[[7, 300, 5000]] = np.nan
    out.append(("nans", a))
    out.append(("all_nan", np.full(700, np.nan)))
    a = rng.randn(M); a[::3] = -0.0; a[1::3] = 0.0
    out.append(("signed_zeros", a))
    out.append(("one_thread_owns_the_smallest", (np.arange(M) % 256).astype(np.float64) * 1000 + np.arange(M) // 256))
    out.append(("short_last_block", np.concatenate([np.full(4096, 5.0), rng.randn(100)])))
    return out


@pytest.mark.parametrize("name,ys", _selection_cases(), ids=[c[0] for c in _selection_cases()])
def test_threshold_and_rank_selection_is_the_sorted_prefix(name, ys):
    for k in (1, 10, 64):
        for cap in (1024, 16):
            got, _ = _select_model(ys, k, cap)
            assert got == _numpy_order(ys, k), (name, k, cap)


def test_the_list_stays_short_on_unstructured_values():
    _, listed = _select_model(np.random.RandomState(1).randn(3 * 4096), 10)
    assert max(listed) <= 64 and listed[-1] == 10           # a few times k per workgroup; the merge lists exactly its k


def test_mt19937_jump_split_equals_the_direct_window():
    N, DEG, PARTS, SEQ_WORDS = 624, 19937, 8, 34 * 624
    Q = (DEG + PARTS - 1) // PARTS
    rng = np.random.RandomState(0)
    seq = rng.randint(0, 2**32, size=SEQ_WORDS, dtype=np.uint64).astype(np.uint32)
    bits = np.unique(np.concatenate([rng.choice(DEG, size=10000, replace=False), [0, DEG - 1, Q - 1, Q, 2 * Q - 1]])).astype(np.int64)
    direct = np.zeros(N, dtype=np.uint32)
    for b in bits:
        direct ^= seq[N - 1 + b:N - 1 + b + N]
    part = [int(np.searchsorted(bits, min(p * Q, 65535), side="left")) for p in range(PARTS + 1)]
    assert part[0] == 0 and part[-1] == len(bits)
    window = np.zeros(N, dtype=np.uint32)
    for p in range(PARTS):
        e0, n = part[p], part[p + 1] - part[p]
        if n <= 0:
            continue
        base = p * Q
        idx = N - 1 + base + np.arange(Q + N + 2)
        xs = np.where(idx < SEQ_WORDS, seq[np.minimum(idx, SEQ_WORDS - 1)], 0).astype(np.uint32)
        rel4 = (bits[e0:e0 + n] - base) * 4
        assert rel4.min() >= 0 and rel4.max() < 65536       # stored as uint16 on the device
        for tid in range(N // 2):
            off = (rel4 + tid * 8) // 4
            window[2 * tid] ^= np.bitwise_xor.reduce(xs[off])
            window[2 * tid + 1] ^= np.bitwise_xor.reduce(xs[off + 1])
    assert np.array_equal(window, direct)
