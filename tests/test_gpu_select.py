"""GPU (-m gpu): the selection launches of gpbo_acq_argbest alone, in both forms, over values the posterior cannot be made
to produce on demand — ys.argmin() / argsort(ys)[:k] of the reference (bayes_opt/acquisition.py:313-317; NumPy: first NaN wins
the argmin, NaNs sort last, -0.0 == 0.0, ties keep the lower index).

The debug build's entry point gpbo_debug_select runs either form (the product uses the passes for k <= 2 and the threshold form
beyond): variant 1 = k block-reduction passes (GPBO_SELECT_V2=0), variant 2 = threshold + rank counting (the default), whose LDS list
is bounded: a tiny GPBO_SELECT_V2_CAP drives it into its fall-back, which must give the same picks."""
import numpy as np
import pytest

from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _reference(ys, k):
    ys = np.asarray(ys, dtype=np.float64)
    nan = np.isnan(ys)
    order = np.lexsort((np.arange(ys.shape[0]), np.where(nan, np.inf, np.where(ys == 0.0, 0.0, ys)), nan))[:k]
    idx = np.full(k, -1, dtype=np.int64)
    idx[:order.shape[0]] = order
    first_nan = int(np.flatnonzero(nan)[0]) if nan.any() else -1
    return idx, first_nan


def _cases():
    rng = np.random.RandomState(0)
    out = [(f"random_{M}", rng.randn(M)) for M in (1, 5, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 9000, 70001)]
    M = 9000
    out.append(("all_equal", np.zeros(M)))
    out.append(("ascending", np.arange(M, dtype=np.float64)))
    out.append(("descending", -np.arange(M, dtype=np.float64)))
    a = rng.randn(M); a[[7, 300, 5000]] = np.nan
    out.append(("nans", a))
    out.append(("all_nan", np.full(700, np.nan)))
    a = rng.randn(M); a[::3] = -0.0; a[1::3] = 0.0
    out.append(("signed_zeros", a))
    out.append(("one_thread_owns_the_smallest", (np.arange(M) % 256).astype(np.float64) * 1000 + np.arange(M) // 256))
    out.append(("three_values", rng.randint(0, 3, M).astype(np.float64)))
    out.append(("short_last_block", np.concatenate([np.full(4096, 5.0), rng.randn(100)])))
    a = rng.randn(M); a[a > 0] = np.inf; a[:10] = -np.inf
    out.append(("infinities", a))
    return out


@pytest.mark.parametrize("name,ys", _cases(), ids=[c[0] for c in _cases()])
def test_both_selection_forms_equal_numpy(debug_engine, monkeypatch, name, ys):
    engine = debug_engine      # gpbo_debug_select + GPBO_SELECT_V2_CAP: debug build
    for k in (1, 10, 64):
        want_idx, want_nan = _reference(ys, k)
        for variant, cap in ((1, None), (2, None), (2, "16"), (2, "1")):
            if cap is None:
                monkeypatch.delenv("GPBO_SELECT_V2_CAP", raising=False)
            else:
                monkeypatch.setenv("GPBO_SELECT_V2_CAP", cap)
            idx, vals, first_nan, _ = engine.debug_select(ys, k, variant=variant)
            tag = f"{name} k={k} variant={variant} cap={cap}"
            assert np.array_equal(idx, want_idx), tag
            assert first_nan == want_nan, tag
            got = vals[idx >= 0]
            assert np.array_equal(got.view(np.int64), ys[idx[idx >= 0]].view(np.int64)), tag      # the values themselves, bit for bit
            assert np.all(np.isnan(vals[idx < 0])), tag


def test_a_full_size_pass_selects_the_same_seeds_in_both_forms(debug_engine, monkeypatch):
    engine = debug_engine
    ys = np.random.RandomState(3).standard_normal(1 << 20)
    ys[123456] = ys[654321] = ys.min() - 1.0            # an exact tie for the minimum, blocks apart
    want_idx, _ = _reference(ys, 64)
    for variant in (1, 2):
        idx, _, first_nan, _ = engine.debug_select(ys, 64, variant=variant)
        assert np.array_equal(idx, want_idx) and first_nan == -1
    assert list(want_idx[:2]) == [123456, 654321]


def test_acq_argbest_gives_the_same_answer_through_either_selection_form(debug_engine, monkeypatch):
    engine = debug_engine      # GPBO_SELECT_V2: debug build
    rng = np.random.RandomState(11)
    X = rng.uniform(size=(200, 3))
    y = np.sin(3 * X.sum(1)) + 0.05 * rng.randn(200)
    yn, ym, ysd = O.normalize_targets(y)
    engine.fit(X, yn, O.MATERN25, 0.5, 1e-6)
    Xc = rng.uniform(size=(30000, 3))
    Xc[[17, 20000]] = Xc[5]                              # duplicates -> exact ties
    engine.set_candidates(Xc)
    engine.posterior(0, ym, ysd, fetch=False)
    monkeypatch.setenv("GPBO_SELECT_V2", "0")
    a = engine.acq_argbest(O.EI, 0.01, float(y.max()), k_seeds=10, return_values=True)
    monkeypatch.setenv("GPBO_SELECT_V2", "1")
    b = engine.acq_argbest(O.EI, 0.01, float(y.max()), k_seeds=10, return_values=True)
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
