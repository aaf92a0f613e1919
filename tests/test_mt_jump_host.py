"""CPU: the MT19937 jump-ahead machinery of csrc/mt_jump.hip (host side) against NumPy's RandomState.

`gpbo_mt19937_jump_blocks` computes the 624-word block n blocks ahead of a state as (x^(624 (n - 1)) mod phi)(F) —
phi found by Berlekamp-Massey at first use (checked inside the library: degree 19937, 135 terms) — which is how
`gpbo_generate_candidates_mt19937` starts its sub-streams on the device.  Drawing 312 n doubles from a RandomState
consumes exactly n blocks, so `get_state()[1]` afterwards is the reference answer.  No GPU involved."""
import ctypes as C

import numpy as np
import pytest

from bayesianoptimization_amd import _lib


def _jump(lib, key, n):
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(624, dtype=np.uint32)
    rc = lib.gpbo_mt19937_jump_blocks(key.ctypes.data_as(C.POINTER(C.c_uint32)), int(n), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == _lib.GPBO_OK, lib.gpbo_last_error(None)
    return out


@pytest.mark.parametrize("seed", [0, 7, 2**31 - 1])
def test_jump_blocks_equals_drawing_from_randomstate(seed):
    lib = _lib.load_library()
    name, key, pos, hg, cg = np.random.RandomState(seed).get_state()
    for n in (1, 2, 3, 33, 34, 35, 1000, 26215, 53431):      # 53431 ~ the blocks of a C3 candidate set (2^20 x 16)
        ref = np.random.RandomState()
        ref.set_state((name, key.copy(), 624, 0, 0.0))
        ref.random_sample(312 * n)
        state = ref.get_state()
        assert state[2] == 624
        assert np.array_equal(_jump(lib, key, n), state[1]), f"seed {seed}, {n} blocks"


def test_jumps_compose_and_reject_bad_arguments():
    lib = _lib.load_library()
    key = np.random.RandomState(3).get_state()[1]
    a = _jump(lib, _jump(lib, key, 700), 1300)
    assert np.array_equal(a, _jump(lib, key, 2000))
    out = np.zeros(624, dtype=np.uint32)
    p = out.ctypes.data_as(C.POINTER(C.c_uint32))
    assert lib.gpbo_mt19937_jump_blocks(p, 0, p) == _lib.ERR_INVALID
    assert lib.gpbo_mt19937_jump_blocks(None, 5, p) == _lib.ERR_INVALID


@pytest.mark.parametrize("burn,n_words", [(0, 2), (0, 624), (1, 622), (1, 624), (2, 620), (5, 10 ** 4), (311, 2 * 70001 * 5), (624, 1248),
                                          (333, 2 * (1 << 20) * 16)])
def test_advance_mt19937_leaves_a_randomstate_where_drawing_would(burn, n_words):
    """engine.advance_mt19937: what a rank that generated only ITS rows of a candidate matrix on its GPU does to its
    RandomState — position arithmetic inside a block, the polynomial jump across blocks — against NumPy drawing the
    doubles (two words each), from any position, a pending gaussian carried over."""
    from bayesianoptimization_amd.engine import advance_mt19937

    ref, mine = np.random.RandomState(12), np.random.RandomState(12)
    for r in (ref, mine):
        if burn:
            r.randint(0, 2**31 - 1, size=burn)      # one word each on this range: an arbitrary position inside the block
        r.standard_normal()                          # leaves a cached gaussian behind
    ref.random_sample(n_words // 2)
    advance_mt19937(mine, n_words)
    a, b = ref.get_state(), mine.get_state()
    assert np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    assert ref.standard_normal() == mine.standard_normal() and ref.uniform() == mine.uniform()
