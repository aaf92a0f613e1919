#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
( time timeout 200 python scripts/theta_search_timing.py ) > $O/theta_search_timing.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/theta_search_timing.json'))
for k,v in d.items():
    r=v['lockstep_rounds_lanes_ms']
    print(k, 'rounds',len(r), 'in lml_batch %.2f ms'%sum(x[1] for x in r), 'fit lockstep %.2f ms'%(v['fit_theta_search_lockstep_s']*1e3), 'sequential %.2f ms'%(v['fit_theta_search_sequential_s']*1e3), v['same_theta'])
PY
( time timeout 100 python scripts/archive/r03_kmat_probe.py ) > $O/kmat_probe.log 2>&1
tail -3 $O/kmat_probe.log
( time timeout 400 python bench.py --no-cpu-baseline ) > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04c/bench_default.json'))
print('C3', d['ms_per_step'], d['roofline']['frac'], d['roofline_fit'])
s=d['suggest_ms']; print({k:s[k] for k in s if k not in('note','default_call_is')})
c=d['configs']['C2']; print('C2', c['ms_per_step'], c['roofline'], c['fit_ms']); s=c['suggest_ms']; print({k:s[k] for k in s if k not in('note','default_call_is')})
PY
tail -3 $O/bench_default.err
