#!/bin/bash
# Round 6 (VERDICT r5 next #1a): the evaluation the theta search waits for, alone.  Kernel trace of gpbo_lml at N = 2048 and
# 4096 (four evaluations each; the first carries the module load) + one PMC pass at 4096.  Run on the GPU box through gpurun;
# copies its summaries into gpurun_out/r06_lml/ (-> profiles/r06_trace_lml_{2048,4096}_kernel_stats.csv, r06_pmc_lml_4096.{txt,json}).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r06_lml}
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
for N in 2048 4096; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$N -o t -- python scripts/lml_trace.py $N > $O/trace_$N.log 2>&1
  tail -2 $O/trace_$N.log
  f=$(find $O/trace_$N -name 't_kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/trace_lml_${N}_kernel_stats.csv
done
mkdir -p $O/pmc; 
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc/pmc_sq -o p -- python scripts/lml_trace.py 4096 > $O/pmc.log 2>&1
mkdir -p $O/pmc/trace; cp $O/trace_lml_4096_kernel_stats.csv $O/pmc/trace/t_kernel_stats.csv
f=$(find $O/pmc/pmc_sq -name 'p_counter_collection.csv' | head -1); [ -n "$f" ] && [ "$f" != "$O/pmc/pmc_sq/p_counter_collection.csv" ] && cp "$f" $O/pmc/pmc_sq/p_counter_collection.csv
python scripts/pmc_summary.py $O/pmc $O/pmc_lml_4096 4
find $O -name '*.db' -delete; find $O -name '*_trace.csv' -size +4M -delete
head -30 $O/trace_lml_4096_kernel_stats.csv | cut -c1-160
