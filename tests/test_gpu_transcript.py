"""GPU (-m gpu): the REAL driver x the REAL engine, closed with a call transcript (VERDICT r5 missing #2 / next #4).

The GPU box has no /root/reference, and the build container has no GPU: `BayesianOptimization.maximize()` of the real bayes_opt
(bayes_opt/bayesian_optimization.py:124-130, 323-333, 348-391) and libgpbo.so never meet in one process.  What does travel is what
the driver SAID to the engine: oracle/gen_transcript.py records, in the build container, every engine call the real driver makes
through accelerate() — five drivers: all-float UCB with the default theta search (30 steps), constrained EI, a mixed float / int /
categorical space, GPHedge, ConstantLiar — with all arguments (arrays with dtype / shape / order, RandomState arguments as MT19937
states) and the oracle's return values.  Here every call is replayed, in order, on the product library, and each return value is
held to the bar of its kind (tests/transcript.Bars): L 1e-10, alpha 1e-8, mu / sd 1e-5 per element and 1e-8 max-norm, LML 1e-10,
its gradient 1e-7, arg-best and seed indices exact where the recorded gaps exceed twice the value bound, candidate matrices and
RandomState positions bitwise, local searches at least as good as SciPy's on the oracle."""
import json
import os

import numpy as np
import pytest

import transcript as TR
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", TR.NAMES)
def test_replay_of_the_real_drivers_calls_on_the_product_library(engine, name):
    T = TR.Transcript(name)
    bars = TR.Bars()
    TR.replay(T, engine, bars, oracle=O)
    # the exact-index branch must carry the test: most random stages have gaps far above 2e-8 of the acquisition's range
    n_arg = bars.exact_argbest + bars.loose_argbest
    assert n_arg > 0 and bars.exact_argbest >= 0.8 * n_arg, (bars.exact_argbest, bars.loose_argbest)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):      # worst error of every kind as a fraction of its bar: profiles/r06_transcript_replay.json
        rec = {"transcript": name, "calls": len(T.calls), "worst_error_over_bar": {k: round(v, 6) for k, v in sorted(bars.worst.items())},
               "argbest_exact": bars.exact_argbest, "argbest_gap_below_bound": bars.loose_argbest,
               "local_searches": len(getattr(bars, "polish", [])),
               "local_searches_better_than_scipy_by_1e-9": int(sum(b < b0 - 1e-9 * max(abs(b0), 1e-12) for b, b0 in getattr(bars, "polish", [])))}
        with open(os.path.join(out, f"transcript_replay_{name}.json"), "w") as fh:
            json.dump(rec, fh, indent=1)
