#!/bin/bash
# End-of-round measurement run (on the GPU box through gpurun): the whole -m gpu suite, smoke(), the rocprofv3 passes of the
# bench command (their summaries are put under profiles/ ON THE BOX first so that the bench line quotes PMC numbers of the
# very library it runs), the bench lines and the round's timing scripts.  Everything lands in gpurun_out/final3/.
# The Cholesky / LML traces and the latency probes of the earlier runs are not repeated here: those kernels have not changed
# since (scripts/r03_chol_run.sh, r03_lml_run.sh reproduce them).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/final3; rm -rf $F; mkdir -p $F
timeout 900 python -m pytest tests -x -q -m gpu > $F/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $F/pytest.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $F/smoke.log 2>&1; tail -1 $F/smoke.log
bash scripts/profile_pmc.sh final3/pmc_C3 --config C3 > $F/pmc_C3.log 2>&1
cp $F/pmc_C3/summary.json profiles/r03_pmc_C3.json; cp $F/pmc_C3/summary.txt profiles/r03_pmc_C3.txt
timeout 300 python bench.py > $F/bench_default.json 2> $F/bench_default.err
python - "$F/bench_default.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("C3", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("parity"))
    print("fit", d["roofline_fit"])
    print("suggest", d.get("suggest_ms"))
    for k, v in d.get("configs", {}).items():
        print(k, v.get("ms_per_step"), v.get("roofline", {}).get("frac"), v.get("parity"), v.get("suggest_ms"))
except Exception as e:
    print("ERR", e)
PY
timeout 100 python scripts/r03_select_probe.py > $F/select_probe.log 2>&1; cp gpurun_out/r03_select_probe.json $F/ 2>/dev/null; cat $F/select_probe.log | tail -9
timeout 120 rocprofv3 --kernel-trace --stats -d $F/c2_trace -o c2 -- python bench.py --config C2 --steps 20 --warmup 3 --no-cpu-baseline --no-suggest > $F/c2_trace.log 2>&1
f=$(find $F/c2_trace -name '*results.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py "$f" > $F/c2_trace_kernel_stats.txt
GPBO_BENCH_DEVICES=0,0 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $F/bench_C4_group2_virtual.json 2> $F/bench_C4_group2_virtual.err
# the k-pass selection (GPBO_SELECT_V2=0) keeps the whole suite green too
GPBO_SELECT_V2=0 timeout 900 python -m pytest tests -x -q -m gpu > $F/pytest_select_v1.log 2>&1; echo "pytest(select v1) rc=$?"; grep -E "passed|failed" $F/pytest_select_v1.log | tail -1
timeout 100 python scripts/theta_search_timing.py > $F/theta.log 2>&1; cp gpurun_out/theta_search_timing.json $F/ 2>/dev/null
timeout 100 python scripts/r03_polish_modes.py > $F/polish_modes.log 2>&1; cp gpurun_out/r03_polish_modes.json $F/ 2>/dev/null
timeout 200 python scripts/r03_chol_probe.py > $F/chol_probe.log 2>&1; cp gpurun_out/r03_chol_probe.json $F/ 2>/dev/null
find $F -name '*.db' -delete      # the summaries stay, the databases do not travel back
echo done
