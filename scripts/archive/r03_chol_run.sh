#!/bin/bash
# Cholesky probe + rocprofv3 kernel trace of the Cholesky alone (on the GPU box, through gpurun).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_chol; rm -rf $O; mkdir -p $O
timeout 240 python scripts/archive/r03_chol_probe.py > $O/probe.log 2>&1; cp gpurun_out/r03_chol_probe.json $O/ 2>/dev/null
grep "chol=3" $O/probe.log | cut -c1-900
for n in 4096 512; do
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/trace_$n -o chol -- python scripts/archive/r03_chol_trace.py $n 3 > $O/trace_$n.log 2>&1
  f=$(find $O/trace_$n -name '*results.db' | head -1)
  echo "== kernel stats n=$n"; [ -n "$f" ] && python scripts/rocpd_summary.py "$f" | cut -c1-140 | head -9
done
