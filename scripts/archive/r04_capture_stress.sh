#!/bin/bash
# Four builds of gpbo_api.hip (theta-search inputs uploaded on the context's stream / on the legacy stream as in rounds 2-3  x
# process-wide capture lock on / off), each linked with the
# tree's other objects into /tmp/exp/<label>/libgpbo.so and run through scripts/archive/r04_capture_stress.py.
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_capture; mkdir -p $O
B=bayesianoptimization_amd/build
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Iinclude -Ibayesianoptimization_amd/csrc -I/opt/rocm/include -DGPBO_CAPTURE_TRACE"
OTHERS=$(ls $B/*.o | grep -v -e gpbo_api.o -e latency_probe.o)
for v in "product:" "async_upload_nolock:-DGPBO_CAPTURE_NOLOCK" "legacy_upload_lock:-DGPBO_LML_SYNC_UPLOAD" "legacy_upload_nolock_r3:-DGPBO_LML_SYNC_UPLOAD -DGPBO_CAPTURE_NOLOCK"; do
  label=${v%%:*}; flags=${v#*:}
  mkdir -p /tmp/exp/$label
  hipcc $COMMON $flags -c bayesianoptimization_amd/csrc/gpbo_api.hip -o /tmp/exp/$label/gpbo_api.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC /tmp/exp/$label/gpbo_api.o $OTHERS -o /tmp/exp/$label/libgpbo.so -ldl || exit 1
  timeout 400 python scripts/archive/r04_capture_stress.py /tmp/exp/$label/libgpbo.so $label ${1:-25} > $O/$label.json 2> $O/$label.err
  echo "$label: $(cat $O/$label.json | cut -c1-400)  capture_failed=$(grep -c 'capture failed' $O/$label.err) stream_replaced=$(grep -c 'replaced the stream' $O/$label.err)"
done
