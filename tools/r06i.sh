#!/bin/bash
# round 6, GPU session i: which stage of the trio loop waits for which (developer counters)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06i}; mkdir -p $out
for w in cfg2_256 cfg2_1024 appf1 book1 book512; do
  echo "== $w" >> $out/stats.log
  LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/rdbg.so timeout 300 python tools/trio_stats.py $w 2>&1 | grep -v amdgpu >> $out/stats.log
done
cat $out/stats.log
