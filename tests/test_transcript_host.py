"""CPU: the call transcripts under tests/golden/transcript_*.npz are what oracle/gen_transcript.py records — the REAL bayes_opt
driving accelerate() — and the oracle reproduces every recorded return value bit for bit when the calls are replayed on a fresh
oracle engine (so the fixture decodes, is complete and self-consistent: what tests/test_gpu_transcript.py holds libgpbo.so to).
With the reference mounted (build container) the recording itself is repeated and compared with the committed file."""
import numpy as np
import pytest

import transcript as TR
from helpers import FakeEngine
from oracle import gp_oracle as O
from oracle.refenv import have_reference


@pytest.mark.parametrize("name", TR.NAMES)
def test_oracle_replays_the_transcript_bitwise(name):
    T = TR.Transcript(name)
    assert T.meta["n_calls"] == len(T.calls) > 100 and T.meta["versions"]["bayes_opt"] == "3.3.0"
    bars = TR.Bars(exact=True)
    TR.replay(T, FakeEngine(), bars, oracle=O)
    kinds = {c["name"] for c in T.calls}
    assert {"fit", "lml_batch", "posterior", "acq_argbest"} <= kinds
    if name != "mixed_space":
        assert {"generate_candidates_like", "get_candidate_rows", "polish_seeds"} <= kinds      # the default device path ran
    else:
        assert {"set_candidates", "predict", "fit_append"} <= kinds                            # host-sampled, host optimiser


def test_array_arguments_keep_dtype_shape_and_order():
    T = TR.Transcript("float_ucb")
    fit = next(c for c in T.calls if c["name"] == "fit")
    X = T.value(fit["args"]["X"])
    assert X.dtype == np.float64 and X.ndim == 2 and X.flags.c_contiguous == (fit["args"]["X"]["order"] == "C")
    gen = next(c for c in T.calls if c["name"] == "generate_candidates_like")
    rs = T.value(gen["args"]["random_state"])
    assert isinstance(rs, np.random.RandomState) and not TR.same_rng(rs, T.value(gen["rng_after"]["random_state"]))
    strided = {"ref": fit["args"]["X"]["ref"], "dtype": "<f8", "shape": fit["args"]["X"]["shape"], "order": "strided"}
    Xs = T.array(strided)
    assert not Xs.flags.c_contiguous and np.array_equal(Xs, X)


@pytest.mark.skipif(not have_reference(), reason="reference not mounted (GPU box)")
@pytest.mark.parametrize("name", ["constant_liar", "mixed_space"])
def test_recording_is_reproducible_from_the_live_reference(name, tmp_path):
    from oracle import gen_transcript as G

    G.generate(outdir=str(tmp_path), only={name})
    a, b = np.load(tmp_path / f"transcript_{name}.npz"), np.load(f"{TR.GOLDEN_DIR}/transcript_{name}.npz")
    assert bytes(a["__calls__"]) == bytes(b["__calls__"])
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        if not k.startswith("__"):
            assert np.array_equal(a[k], b[k])
