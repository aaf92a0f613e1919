"""CPU: spaces with integer / categorical parameters (second half).

* `MixedSpace` — the stand-in the GPU box uses, where bayes_opt is not installed — against the reference's TargetSpace
  (bayes_opt/target_space.py:237-301, 340-347, 565-603; parameter.py:237-449): same bounds, masks, samples, RandomState
  consumption and kernel_transform (its categorical batch behaviour included), bit for bit.
* the host half of the device assembly (`GpEngine.generate_candidates_mixed`): float runs go to the device generator with
  the stream state threaded through, int / categorical parameters are drawn by their own `random_sample` from exactly the
  position the device hands back.  Here the two C entry points are replaced by NumPy stand-ins (RandomState itself), so the
  test pins the threading of (key, pos), the merging of float runs and the column bookkeeping — not the device kernels
  (tests/test_gpu_seams.py does that on the GPU).
* `_mixed_groups_on_device`: which spaces qualify."""
import ctypes as C
import warnings

import numpy as np
import pytest

from bayesianoptimization_amd import fused_acquisition as A
from bayesianoptimization_amd.engine import GpEngine
from bayesianoptimization_amd.float_space import FloatSpace, MixedSpace
from oracle.refenv import have_reference, import_reference

PB = {"a": (0.0, 2.0), "n": (-3, 7, int), "b": (1.0, 4.0), "c": ("x", "y", "z"), "e": (5.0, 6.0), "f": (0.0, 1.0), "k": (0, 1, int)}


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference")
def test_mixed_space_stand_in_is_the_reference_target_space():
    import_reference()
    from bayes_opt.target_space import TargetSpace

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ts = TargetSpace(None, PB)
    ms = MixedSpace(PB)
    assert ts.keys == ms.keys and ts.dim == ms.dim == 9
    assert np.array_equal(ts.bounds, ms.bounds)
    assert all(np.array_equal(ts.masks[k], ms.masks[k]) for k in ts.keys)
    assert np.array_equal(ts.continuous_dimensions, ms.continuous_dimensions)
    for n in (0, 1, 7, 5000):
        r1, r2 = np.random.RandomState(5), np.random.RandomState(5)
        a, b = ts.random_sample(n, r1), ms.random_sample(n, r2)
        assert np.array_equal(a, b) and r1.uniform() == r2.uniform()
        assert np.array_equal(ts.kernel_transform(a), ms.kernel_transform(b))
    # the categorical transform's batch behaviour (parameter.py:446-449): one row keeps its own one-hot, a batch gets every
    # column that is some row's argmax
    x = ms.random_sample(3, np.random.RandomState(1))
    assert np.array_equal(ms.kernel_transform(x[0])[0, 4:7], x[0, 4:7])
    assert ms.kernel_transform(ms.random_sample(500, 2))[:, 4:7].min() == 1.0
    for k in ts.keys:
        assert type(ts._params_config[k]).__name__ == type(ms._params_config[k]).__name__
    # and the reference's own TargetSpace qualifies for the device assembly exactly like the stand-in
    eng = _engine_over(_NumpyLib())
    g_ref = A._mixed_groups_on_device([_Gp(eng, ts.kernel_transform)], ts, np.random.RandomState(0), 10**5)
    g_ms = A._mixed_groups_on_device([_Gp(eng, ms.kernel_transform)], ms, np.random.RandomState(0), 10**5)
    assert g_ref is not None and [(g[0], g[1], g[2]) for g in g_ref] == [(g[0], g[1], g[2]) for g in g_ms]
    assert all(np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) for a, b in zip(g_ref, g_ms))
    r1, r2 = np.random.RandomState(9), np.random.RandomState(9)
    want = ts.random_sample(3000, r1)
    eng.generate_candidates_mixed(3000, g_ref, r2)
    assert np.array_equal(eng._lib.X, want) and r1.uniform() == r2.uniform()


class _NumpyLib:
    """NumPy stand-ins of the two column-group entry points: the float run is RandomState.uniform itself."""

    def __init__(self):
        self.X = None
        self.calls = []

    def gpbo_generate_candidate_columns_mt19937(self, h, M, d_total, col0, ncols, lo, hi, key, pos):
        if self.X is None or self.X.shape != (M, d_total):
            self.X = np.full((M, d_total), np.nan)
        lo = np.ctypeslib.as_array(lo, (ncols,))
        hi = np.ctypeslib.as_array(hi, (ncols,))
        k = np.ctypeslib.as_array(key, (624,))
        p = C.cast(pos, C.POINTER(C.c_int))
        rs = np.random.RandomState()
        rs.set_state(("MT19937", k.copy(), int(p.contents.value), 0, 0.0))
        for t in range(ncols):
            self.X[:, col0 + t] = rs.uniform(lo[t], hi[t], M)
        st = rs.get_state()
        k[:] = st[1]
        p.contents.value = int(st[2])
        self.calls.append(("device", col0, ncols))
        return 0

    def gpbo_set_candidate_columns(self, h, values, M, d_total, col0, ncols):
        if self.X is None or self.X.shape != (M, d_total):
            self.X = np.full((M, d_total), np.nan)
        self.X[:, col0:col0 + ncols] = np.ctypeslib.as_array(values, (M, ncols))
        self.calls.append(("host", col0, ncols))
        return 0


def _engine_over(lib):
    eng = GpEngine.__new__(GpEngine)          # no device: only the host logic of generate_candidates_mixed runs
    eng._lib, eng._h = lib, None
    eng.n_candidates = 0
    return eng


class _Gp:
    def __init__(self, eng, transform):
        self._eng, self.transform = eng, transform

    def _engine(self):
        return self._eng


@pytest.mark.parametrize("pb", [PB, {"n": (0, 9, int), "a": (0.0, 1.0)}, {"a": (0.0, 1.0), "c": ("p", "q")},
                                {"c": ("p", "q", "r", "s"), "n": (2, 3, int)}])
@pytest.mark.parametrize("M", [1000, 4097])
def test_the_assembly_threads_one_stream_through_device_runs_and_host_parameters(pb, M):
    ms = MixedSpace(pb)
    lib = _NumpyLib()
    eng = _engine_over(lib)
    groups = A._mixed_groups_on_device([_Gp(eng, ms.kernel_transform)], ms, np.random.RandomState(0), 10**6)
    assert groups is not None and [g[0] for g in groups] == [A._PARAM_KINDS[type(ms._params_config[k]).__name__] for k in ms.keys]
    ref, dev = np.random.RandomState(11), np.random.RandomState(11)
    for r in (ref, dev):
        r.standard_normal(3)                     # an odd stream position and a cached gaussian: both must survive
    want = ms.random_sample(M, ref)
    eng.generate_candidates_mixed(M, groups, dev)
    assert np.array_equal(lib.X, want)
    assert np.array_equal(dev.get_state()[1], ref.get_state()[1]) and dev.get_state()[2:] == ref.get_state()[2:]
    assert dev.standard_normal() == ref.standard_normal() and dev.uniform() == ref.uniform()
    # consecutive float parameters travel in ONE device call
    floats = [type(ms._params_config[k]).__name__ == "FloatParameter" for k in ms.keys]
    runs = sum(1 for i, f in enumerate(floats) if f and (i == 0 or not floats[i - 1]))
    assert sum(c[0] == "device" for c in lib.calls) == runs
    assert sum(c[0] == "host" for c in lib.calls) == len(floats) - sum(floats)


def test_which_spaces_the_device_assembles():
    ms = MixedSpace(PB)
    eng = _engine_over(_NumpyLib())
    rs = np.random.RandomState(0)
    ok = [_Gp(eng, ms.kernel_transform)]
    assert A._mixed_groups_on_device(ok, ms, rs, 10**5) is not None
    assert A._mixed_groups_on_device(ok, ms, rs, 10) is None                                    # too small to be worth a launch
    assert A._mixed_groups_on_device(ok, ms, np.random.default_rng(0), 10**5) is None            # not a legacy RandomState
    assert A._mixed_groups_on_device([_Gp(eng, None)], ms, rs, 10**5) is None                    # the GP would see raw inputs
    assert A._mixed_groups_on_device([_Gp(eng, lambda x: x)], ms, rs, 10**5) is None             # some other transform
    assert A._mixed_groups_on_device([_Gp(eng, MixedSpace(PB).kernel_transform)], ms, rs, 10**5) is None   # another space's
    fs = FloatSpace({"a": (0.0, 1.0)})
    assert A._mixed_groups_on_device([_Gp(eng, fs.kernel_transform)], fs, rs, 10**5) is None     # all-float: the plain device stream

    class Custom(type(ms._params_config["n"])):
        pass
    ms2 = MixedSpace(PB)
    ms2._params_config["n"] = Custom("n", (-3, 7))
    assert A._mixed_groups_on_device([_Gp(eng, ms2.kernel_transform)], ms2, rs, 10**5) is None   # a parameter class of the user's
    grp = _engine_over(_NumpyLib())
    grp.mixed_device_sampling = False                                                            # (a device group)
    assert A._mixed_groups_on_device([_Gp(grp, ms.kernel_transform)], ms, rs, 10**5) is None
