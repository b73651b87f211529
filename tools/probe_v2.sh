#!/bin/bash
# usage (on the GPU box): tools/probe_v2.sh <variant> ...  -- single-interval phase probes of the lean core (LZ4HIP_V2_PROBE builds)
# on the GPU box: per-interval timing of the lean loop, one interval per library variant (tools/build_variant.sh probeK -DLZ4HIP_V2_PROBE=K)
cd $GRAFT_REPO_ROOT
cp lz4-java_amd/liblz4hip.so /tmp/base.so
for k in 1 2 3 4 5 6; do
  cp lz4-java_amd/variants/probe$k.so lz4-java_amd/liblz4hip.so
  CC=3 RING=0 timeout 100 python tools/gpu_phase_profile.py 1024 ${1:-synth} 2>&1 | grep -v "^ *-\|total\|amdgpu.ids" | awk -v k=$k 'NR==1{print} NR==k+1{print}'
done
cp /tmp/base.so lz4-java_amd/liblz4hip.so
