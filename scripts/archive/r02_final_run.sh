#!/bin/bash
# End-of-round measurement run (on the GPU box through gpurun): the whole -m gpu suite, smoke(), the rocprofv3 passes of the
# bench command (their summaries are put under profiles/ ON THE BOX first so that the bench lines quote PMC numbers of the
# very library they run), the bench lines and the timing scripts.  Everything lands in gpurun_out/final/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/final; rm -rf $F; mkdir -p $F
timeout 1500 python -m pytest tests -x -q -m gpu > $F/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $F/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $F/smoke.log 2>&1; tail -1 $F/smoke.log
for c in C3 C5; do
  bash scripts/profile_pmc.sh final/pmc_$c --config $c > $F/pmc_$c.log 2>&1
  cp $F/pmc_$c/summary.json profiles/r02_pmc_$c.json; cp $F/pmc_$c/summary.txt profiles/r02_pmc_$c.txt
done
python bench.py > $F/bench_C3.json 2> $F/bench_C3.err
python bench.py --config C5 > $F/bench_C5.json 2> $F/bench_C5.err
python bench.py --config C4 --no-cpu-baseline --no-suggest > $F/bench_C4_1gpu.json 2> $F/bench_C4_1gpu.err
GPBO_BENCH_DEVICES=0,0 python bench.py --gpus 2 --no-cpu-baseline --no-suggest > $F/bench_C4_group2_virtual.json 2> $F/bench_C4_group2_virtual.err
python scripts/config_table.py > $F/config_table.log 2>&1; cp gpurun_out/config_table.json $F/ 2>/dev/null
python scripts/archive/r02_fit_probe.py > $F/fit_probe.log 2>&1; cp gpurun_out/r02_fit_probe.json $F/ 2>/dev/null
python scripts/theta_search_timing.py > $F/theta.log 2>&1; cp gpurun_out/theta_search_timing.json $F/ 2>/dev/null
python scripts/append_latency.py > $F/append.log 2>&1; cp gpurun_out/append_latency.json $F/ 2>/dev/null
python scripts/archive/r02_suggest_modes.py > $F/suggest_modes.log 2>&1; cp gpurun_out/r02_suggest_modes.json $F/ 2>/dev/null
python scripts/mt19937_timing.py > $F/mt_timing.log 2>&1
for f in bench_C3 bench_C5 bench_C4_1gpu bench_C4_group2_virtual; do python - "$F/$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("parity"), d.get("suggest_ms"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
