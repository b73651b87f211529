#!/bin/bash
# round 6, GPU session d: the headline decoder against the traffic-mix benchmark's finding (fewer wavefronts per CU, more match sources in flight per block):
# the staged, deep and ring loops with LDS padding that limits the wavefronts a CU holds, the ring loop with 4 / 8 slots
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06d}; mkdir -p $out
V="d,4:0:1:0,8:2:0:0,8:3:0:512,4:3:0:512,4:3:0:1024"
echo "== product" > $out/occ.log
timeout 300 python tools/ring_matrix.py appf65536 $V >> $out/occ.log 2>&1
for v in pad8k pad16k pad24k pad40k pad8k_s8 pad16k_s8 pad24k_s8 s8; do
  [ -f lz4-java_amd/variants/$v.so ] || continue
  echo "== $v" >> $out/occ.log
  LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/$v.so timeout 300 python tools/ring_matrix.py appf65536 $V >> $out/occ.log 2>&1
done
grep -v amdgpu.ids $out/occ.log
