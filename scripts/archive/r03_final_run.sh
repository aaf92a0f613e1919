#!/bin/bash
# End-of-round measurement run (on the GPU box through gpurun): the whole -m gpu suite, smoke(), the rocprofv3 passes of the
# bench command (their summaries are put under profiles/ ON THE BOX first so that the bench line quotes PMC numbers of the
# very library it runs), the bench line, kernel traces of whole suggest() calls at C2, and the suite once more on the
# round-2 forms of the two kernels this round replaced late (selection, MT19937 jump).  Everything lands in gpurun_out/final3/.
# Not repeated here (those kernels have not changed since their runs, see profiles/README.md): scripts/archive/r03_chol_run.sh,
# r03_lml_run.sh, r03_select_probe.py, r03_chol_probe.py, theta_search_timing.py, the C2 step trace, the 2-virtual-rank bench.
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
for spec in "0 reference" "10 device"; do
  set -- $spec
  tag=suggest_C2_n_smart_$1_$2
  timeout 100 rocprofv3 --kernel-trace --stats -d $F/$tag -o t -- python scripts/archive/r03_suggest_trace.py C2 $1 $2 20 > $F/$tag.log 2>&1
  f=$(find $F/$tag -name '*results.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py "$f" > $F/${tag}_kernel_stats.txt
  grep -E "median" $F/$tag.log
done
# the round-2 selection (k passes) and MT19937 jump (one workgroup per window) keep the whole suite green too
GPBO_SELECT_V2=0 GPBO_MT_JUMP_SPLIT=0 timeout 900 python -m pytest tests -x -q -m gpu > $F/pytest_round2_forms.log 2>&1; echo "pytest(round-2 forms) rc=$?"; grep -E "passed|failed" $F/pytest_round2_forms.log | tail -1
timeout 100 python scripts/archive/r03_polish_modes.py > $F/polish_modes.log 2>&1; cp gpurun_out/r03_polish_modes.json $F/ 2>/dev/null
find $F -name '*.db' -delete      # the summaries stay, the databases do not travel back
echo done
