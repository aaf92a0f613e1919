#!/bin/bash
# rocprofv3 passes for the bench command (run on the GPU box through gpurun).  Kernel-trace stats first,
# then PMC counters in their own passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots).
# usage: scripts/profile_pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; shift
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-suggest $*"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o p -- $CMD > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o p -- $CMD > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o p -- $CMD > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
python scripts/pmc_summary.py "$OUT" "$OUT/summary" 2
find "$OUT" -name "*.csv" | head -30
