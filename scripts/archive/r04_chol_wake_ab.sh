#!/bin/bash
# Whether a wave defers its catch-up while its SIMD-mate owns the chain (GPBO_CHOL_DEFER_MATE), how often the owner of a column block wakes the waiting waves (s_wakeup behind every 2nd / 4th / 8th column / never) and how long
# they sleep between two looks at a marker: builds of chol_kernels.hip with -DGPBO_CHOL_WAKE_MASK / -DGPBO_CHOL_POLL_SLEEP linked
# with the tree's other debug objects, each run through scripts/archive/r04_chol_chain.py.
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_wake; mkdir -p $O
B=bayesianoptimization_amd/build_dbg
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Iinclude -Ibayesianoptimization_amd/csrc -I/opt/rocm/include -DGPBO_DEBUG"
OTHERS=$(ls $B/*.o | grep -v chol_kernels.o)
for v in "product:" "no_defer_mate:-DGPBO_CHOL_DEFER_MATE=0" "defer_never_sleep2:-DGPBO_CHOL_WAKE_MASK=8 -DGPBO_CHOL_POLL_SLEEP=2" "nodefer_never_sleep2:-DGPBO_CHOL_DEFER_MATE=0 -DGPBO_CHOL_WAKE_MASK=8 -DGPBO_CHOL_POLL_SLEEP=2" "wake4_sleep8:-DGPBO_CHOL_WAKE_MASK=3" "wake8_sleep8:-DGPBO_CHOL_WAKE_MASK=7" "wake1_sleep8:-DGPBO_CHOL_WAKE_MASK=0" "wake2_sleep64:-DGPBO_CHOL_POLL_SLEEP=64"; do
  label=${v%%:*}; flags=${v#*:}
  mkdir -p /tmp/exp/$label
  hipcc $COMMON $flags -c bayesianoptimization_amd/csrc/chol_kernels.hip -o /tmp/exp/$label/chol_kernels.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC /tmp/exp/$label/chol_kernels.o $OTHERS -o /tmp/exp/$label/libgpbo_dbg.so -ldl || exit 1
  GPBO_CHAIN_LIB=/tmp/exp/$label/libgpbo_dbg.so timeout 100 python scripts/archive/r04_chol_chain.py 128 512 4096 > $O/$label.log 2>&1
  python - "$O/$label.log" "$label" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line[:1].isdigit():
        n, rest = line.split(" ", 1); d = json.loads(rest)
        print(sys.argv[2], n, d["ms"], list(d["phases_cycles"].values()), d["diag_workgroup_cycles"], "err", d["rel_err_L"], "info", d["info"])
PY
done
