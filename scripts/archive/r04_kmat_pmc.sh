#!/bin/bash
# Round 4: counters for kmat_kernel alone — is the kernel-matrix assembly bound by its fp64 VALU stream, as
# DESIGN.md claims from an instruction count, or by something the counters show?  Kernel trace first, then PMC passes in
# their own runs (no trace domains besides --kernel-trace), counters restricted to the kernel.
# usage (on the GPU box, through gpurun): scripts/archive/r04_kmat_pmc.sh <outdir-under-gpurun_out>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1
mkdir -p "$OUT"
CMD="python scripts/archive/r04_kmat_target.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/trace.log" 2> "$OUT/trace.err"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-include-regex kmat_kernel --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o p -- $CMD > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE \
  --kernel-include-regex kmat_kernel --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -o p -- $CMD > /dev/null 2> "$OUT/pmc_sq2.err"
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex kmat_kernel --kernel-trace --output-format csv -d "$OUT/pmc_write" -o p -- $CMD > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex kmat_kernel --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o p -- $CMD > /dev/null 2> "$OUT/pmc_fetch.err"
python scripts/archive/r04_kmat_pmc_summary.py "$OUT" "$OUT/summary"
tail -3 "$OUT"/*.err | tail -30
