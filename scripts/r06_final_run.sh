#!/bin/bash
# Round 6, end-of-round measurement run (on the GPU box through gpurun): the whole -m gpu suite, smoke(), the rocprofv3 passes of the
# bench command for C3 and C2 (their summaries are put under profiles/ ON THE BOX first, so that the bench line quotes PMC numbers of
# the very library it runs), the default bench line, the 8-virtual-rank group line, the kernel trace + SQ counters of gpbo_lml alone
# at N = 2048 / 4096, the theta-search timing, the small-fit timing of the three fit paths, the maximize() loop (seed 1 with the host
# columns, seeds 2-4 without), the local-search A/B, the host profile and the kernel trace of the small-N loop, the per-call breakdown of a C1 / C2 step, the
# lanes-per-group sweep and the fit-side GEMM benchmark.  Everything lands in
# gpurun_out/r06f/ (scripts/r06_collect_final.py copies it to profiles/).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/r06f; rm -rf $F; mkdir -p $F
( time timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $F/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $F/pytest.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $F/smoke.log 2>&1; tail -1 $F/smoke.log
bash scripts/profile_pmc.sh r06f/pmc_C3 --config C3 > $F/pmc_C3.log 2>&1
cp $F/pmc_C3/summary.json profiles/r06_pmc_C3.json; cp $F/pmc_C3/summary.txt profiles/r06_pmc_C3.txt
bash scripts/profile_pmc.sh r06f/pmc_C2 --config C2 > $F/pmc_C2.log 2>&1
( time timeout 600 python bench.py ) > $F/bench_default.json 2> $F/bench_default.err
python - "$F/bench_default.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("C3", d["value"], d["ms_per_step"], r["frac"], r.get("frac_of_measured"), r.get("peak_measured"), r.get("sustained_mhz"), r.get("traffic"), d.get("parity"))
    print("summary", json.dumps(d.get("summary")))
except Exception as e:
    print("ERR", e)
PY
( time GPBO_BENCH_DEVICES=0,0,0,0,0,0,0,0 timeout 600 python bench.py --gpus 8 ) > $F/bench_C4_group8_virtual.json 2> $F/bench_C4_group8_virtual.err
tail -c 500 $F/bench_C4_group8_virtual.json; tail -2 $F/bench_C4_group8_virtual.err
timeout 400 bash scripts/r06_lml_evidence.sh r06f/lml > $F/lml_evidence.log 2>&1
python scripts/r06_lml_timeline.py $F/lml/trace_4096/t_kernel_trace.csv > $F/lml_4096_timeline.txt 2>/dev/null
python scripts/r06_lml_timeline.py $F/lml/trace_2048/t_kernel_trace.csv > $F/lml_2048_timeline.txt 2>/dev/null
tail -3 $F/lml_4096_timeline.txt | head -1; grep "# total" $F/lml_4096_timeline.txt $F/lml_2048_timeline.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $F/c2_trace -o t -- python bench.py --config C2 --steps 20 --warmup 3 --no-cpu-baseline --no-suggest > $F/c2_trace.json 2> $F/c2_trace.err
timeout 200 python scripts/theta_search_timing.py > $F/theta_search_timing.log 2>&1; cp gpurun_out/theta_search_timing.json $F/ 2>/dev/null
timeout 300 python scripts/r05_small_fit_timing.py > $F/small_fit_timing.json 2> $F/small_fit_timing.err
timeout 500 python scripts/r06_maximize_loop.py > $F/maximize_loop.json 2> $F/maximize_loop.err; tail -1 $F/maximize_loop.err | cut -c1-300
for s in 2 3 4; do timeout 200 python scripts/r06_maximize_loop.py --no-cpu --seed $s > $F/maximize_loop_seed$s.json 2>/dev/null; done
timeout 200 python scripts/r06_polish_fused_ab.py > $F/polish_fused_ab.json 2> $F/polish_fused_ab.err; tail -1 $F/polish_fused_ab.err | cut -c1-200
timeout 200 python scripts/r06_suggest_host_profile.py > $F/suggest_host_profile.txt 2>&1; head -2 $F/suggest_host_profile.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $F/small_n_trace -o t -- python scripts/r06_suggest_host_profile.py > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $F/n300_trace -o t -- python scripts/r06_suggest_host_profile.py 300 340 > /dev/null 2>&1
timeout 200 python scripts/r06_suggest_host_profile.py 16 64 > $F/suggest_host_profile_n16_64.txt 2>&1; head -1 $F/suggest_host_profile_n16_64.txt
timeout 200 python scripts/r06_small_step_breakdown.py > $F/small_step_breakdown.txt 2>&1; grep "whole step" $F/small_step_breakdown.txt
timeout 300 python scripts/r06_lanes_grouping.py > $F/lanes_grouping.json 2> $F/lanes_grouping.err
timeout 200 python scripts/r06_gemm_bench.py > $F/gemm_bench.json 2> $F/gemm_bench.err
timeout 200 python scripts/r06_post_10k_ab.py > $F/post_10k_ab.json 2> $F/post_10k_ab.err
cp gpurun_out/r04_polish_sweep.json $F/polish_sweep.json 2>/dev/null; cp gpurun_out/r05_conditioning.json $F/conditioning.json 2>/dev/null
cp gpurun_out/transcript_replay_*.json $F/ 2>/dev/null
find $F -name '*.db' -delete; find $F -name '*_kernel_trace.csv' -size +3M -delete; find $F -name 'p_counter_collection.csv' -size +3M -delete
ls $F
echo done
