#!/bin/bash
# round 5, GPU session y: the lane-per-run trip -- PMC passes at 2048 blocks (instructions per sequence, waits, LDS), and the two-window form again
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r05y}; mkdir -p $out
head -3 tools/pmc_sets_ring.txt > $out/sets.txt
for n in 2048; do
bash tools/pmc_run.sh $out/n$n decode_wave "python tools/ring_matrix.py cfg2_$n d" < $out/sets.txt > $out/n$n.log 2>&1
echo "== n $n"; cat $out/n$n/summary.txt | cut -c40-
done
for v in nwin2; do
  echo "== variant $v"
  LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/$v.so timeout 600 python tools/ring_matrix.py cfg2_1024,cfg2_2048,appf1024,appf2048,book2048 d 2>&1 | tail -10
done > $out/variants.log 2>&1
cat $out/variants.log
