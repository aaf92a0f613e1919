#!/bin/bash
# theta-search timing + rocprofv3 kernel trace of single LML evaluations (on the GPU box, through gpurun).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_lml; rm -rf $O; mkdir -p $O
timeout 200 python scripts/theta_search_timing.py > $O/theta.log 2>&1; cp gpurun_out/theta_search_timing.json $O/ 2>/dev/null
cut -c1-1500 $O/theta.log | tail -4
timeout 120 rocprofv3 --kernel-trace --stats -d $O/trace -o lml -- python scripts/lml_trace.py 4096 > $O/trace.log 2>&1
tail -4 $O/trace.log
f=$(find $O/trace -name '*results.db' | head -1)
[ -n "$f" ] && python scripts/rocpd_summary.py "$f" | cut -c1-150 | head -24
