#!/bin/bash
# round 5, GPU session e: PMC passes of the parallel wave kernel (decode_pipe 5) at 1 and 4 wavefronts per SIMD
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r05e}; mkdir -p $out
head -3 tools/pmc_sets_ring.txt > $out/sets.txt
for n in ${2:-1024 4096}; do
bash tools/pmc_run.sh $out/n$n decode_wave "python tools/ring_matrix.py cfg2_$n 64:${3:-5}:0:0" < $out/sets.txt > $out/n$n.log 2>&1
echo "== n $n"; cat $out/n$n/summary.txt | cut -c40-
done
