#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log
( time timeout 200 python scripts/archive/r04_post_small_np_ab.py ) > $O/post_ab.log 2>&1
tail -8 $O/post_ab.log
( time timeout 400 python bench.py --no-cpu-baseline ) > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b/bench_default.json'))
r=d['roofline']; print({k:r[k] for k in r if k not in ('mfma_probes','peak_measured_note','kernel')}); print(r.get('mfma_probes'))
print('C3', d['ms_per_step'], d['suggest_ms'])
c=d['configs']['C2']; print('C2', c['ms_per_step'], c['roofline'], c['fit_ms'], c.get('suggest_ms'))
print('C1', d['configs']['C1']['ms_per_step'])
PY
tail -3 $O/bench_default.err
