#!/bin/bash
# Round 5: where do the cycles of BASELINE config 2's posterior kernel go?  Two rocprofv3 --pmc passes (eight SQ
# counters each) and a kernel trace over the C2 bench command; scripts/r05_pmc_C2_posterior.py turns them into
# profiles/r05_pmc_C2_posterior.{txt,json}.  Run on the GPU box through gpurun.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05_pmc_C2_posterior; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --config C2 --steps 20 --warmup 3 --no-cpu-baseline --no-suggest"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_a -o p -- $CMD > /dev/null 2> $OUT/pmc_a.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_b -o p -- $CMD > /dev/null 2> $OUT/pmc_b.err
python scripts/r05_pmc_C2_posterior.py $OUT gpurun_out/r05_pmc_C2_posterior_summary
find $OUT -name '*.db' -delete
tail -3 $OUT/pmc_a.err $OUT/pmc_b.err
