#!/bin/bash
# kernel trace + host profile of whole suggest() calls at C2 (on the GPU box through gpurun)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_suggest_trace; rm -rf $O; mkdir -p $O
for spec in "0 reference" "10 device"; do
  set -- $spec
  tag=n_smart_$1_$2
  timeout 100 rocprofv3 --kernel-trace --stats -d $O/$tag -o t -- python scripts/archive/r03_suggest_trace.py C2 $1 $2 20 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name '*results.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py "$f" > $O/${tag}_kernel_stats.txt
  grep -E "median" $O/$tag.log
done
timeout 60 python scripts/archive/r03_suggest_trace.py C2 0 reference 30 > $O/n_smart_0_untraced.log 2>&1; grep median $O/n_smart_0_untraced.log
find $O -name '*.db' -delete
