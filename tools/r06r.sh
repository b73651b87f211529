#!/bin/bash
# round 6, GPU session r: what the parallel wave loop does on the shard of 8 (developer counters: trips, sequences per trip, rounds, one-sequence steps, short stream ring)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06r}; mkdir -p $out
for w in cfg2_2048 appf2048 book2048; do
  LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/rdbg.so timeout 300 python tools/wave_stats.py $w 2>&1 | grep -v amdgpu | tail -3 >> $out/stats.log
done
cat $out/stats.log
