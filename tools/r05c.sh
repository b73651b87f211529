#!/bin/bash
# round 5, GPU session c: PMC passes of the wave kernel at 1 and 4 wavefronts per SIMD (tools/pmc_run.sh; kernel-trace + pmc only)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/r05c; mkdir -p $out
for n in 1024 4096; do
  bash tools/pmc_run.sh $out/n$n decode_wave "python tools/ring_matrix.py cfg2_$n 64:4:0:0" < tools/pmc_sets_ring.txt > $out/n$n.log 2>&1
  echo "== cfg2_$n"; cat $out/n$n/summary.txt
done
