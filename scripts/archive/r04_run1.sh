#!/bin/bash
# round 4, first GPU call: the whole -m gpu suite on the split libraries (+ the new parity matrix and the polish sweep),
# the LML crossover, kmat counters, the theta-search timing baseline and the default bench line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log
( time timeout 300 python scripts/archive/r04_lml_crossover.py ) > $O/lml_crossover.log 2>&1
tail -12 $O/lml_crossover.log
( time timeout 200 python scripts/theta_search_timing.py ) > $O/theta_search_timing.log 2>&1
tail -5 $O/theta_search_timing.log
( time timeout 400 scripts/archive/r04_kmat_pmc.sh r04a/kmat_pmc ) > $O/kmat_pmc.log 2>&1
tail -40 $O/kmat_pmc.log
( time timeout 400 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json
tail -5 $O/bench_default.err
