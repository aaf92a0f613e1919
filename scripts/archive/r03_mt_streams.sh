#!/bin/bash
# candidate generation (reference stream on the device) by sub-stream count, with the split jump kernel and with the
# round-2 one (on the GPU box through gpurun)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_mt_streams.log; : > $O
for split in 1 0; do
  for s in 16 64 128 256 512; do
    echo "split=$split streams=$s" >> $O
    GPBO_MT_JUMP_SPLIT=$split GPBO_MT_STREAMS=$s timeout 40 python scripts/mt19937_timing.py 2>/dev/null | grep "^M=" >> $O
  done
done
cat $O
