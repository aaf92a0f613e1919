#!/bin/bash
# Round 4, end-of-round measurement run (on the GPU box through gpurun): the whole -m gpu suite, smoke(), the rocprofv3 passes
# of the bench command for C3 and C2 (their summaries are put under profiles/ ON THE BOX first, so that the bench line quotes PMC
# numbers of the very library it runs), the default bench line, the 2-virtual-rank group line, a kernel trace of the C2 step and
# the theta-search timing.  Everything lands in gpurun_out/r04f/ (scripts/archive/r04_collect_final.py copies it to profiles/).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/r04f; rm -rf $F; mkdir -p $F
( time timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $F/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $F/pytest.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $F/smoke.log 2>&1; tail -1 $F/smoke.log
bash scripts/profile_pmc.sh r04f/pmc_C3 --config C3 > $F/pmc_C3.log 2>&1
cp $F/pmc_C3/summary.json profiles/r04_pmc_C3.json; cp $F/pmc_C3/summary.txt profiles/r04_pmc_C3.txt
bash scripts/profile_pmc.sh r04f/pmc_C2 --config C2 > $F/pmc_C2.log 2>&1
cp $F/pmc_C2/summary.json profiles/r04_pmc_C2.json; cp $F/pmc_C2/summary.txt profiles/r04_pmc_C2.txt
( time timeout 400 python bench.py ) > $F/bench_default.json 2> $F/bench_default.err
python - "$F/bench_default.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("C3", d["value"], d["ms_per_step"], r["frac"], r.get("frac_of_measured"), r.get("peak_measured"), r.get("sustained_mhz"), r.get("traffic"), d.get("parity"))
    print("fit", d["roofline_fit"])
    s = d.get("suggest_ms", {}); print("suggest", {k: s[k] for k in s if k not in ("note", "default_call_is", "default_call_per_restart_seed")})
    for k, v in d.get("configs", {}).items():
        print(k, v.get("ms_per_step"), v.get("roofline", {}).get("frac"), v.get("parity", {}).get("argmin_equals_reference"), (v.get("cpu_baseline") or {}).get("value"))
        if "suggest_ms" in v:
            s = v["suggest_ms"]; print("   ", {q: s[q] for q in s if q not in ("note", "default_call_is", "default_call_per_restart_seed")})
except Exception as e:
    print("ERR", e)
PY
( time GPBO_BENCH_DEVICES=0,0 timeout 400 python bench.py --gpus 2 ) > $F/bench_C4_group2_virtual.json 2> $F/bench_C4_group2_virtual.err
tail -c 1500 $F/bench_C4_group2_virtual.json; tail -2 $F/bench_C4_group2_virtual.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $F/c2_trace -o t -- python bench.py --config C2 --steps 20 --warmup 3 --no-cpu-baseline --no-suggest > $F/c2_trace.json 2> $F/c2_trace.err
timeout 200 python scripts/theta_search_timing.py > $F/theta_search_timing.log 2>&1; cp gpurun_out/theta_search_timing.json $F/ 2>/dev/null
timeout 100 python scripts/archive/r04_chol_chain.py 128 512 2048 4096 > $F/chol_chain.log 2>&1; cp gpurun_out/r04_chol_chain.json $F/ 2>/dev/null
find $F -name '*.db' -delete
ls $F
echo done
