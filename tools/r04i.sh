#!/bin/bash
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/r04i; mkdir -p $out
bash tools/pmc_run.sh $out/lane decode_ring 'python tools/ring_matrix.py book65536 1:3:0:256' < tools/pmc_sets_mem.txt > $out/lane.log 2>&1
bash tools/pmc_run.sh $out/stg decode_kernel 'python tools/ring_matrix.py book65536 4:0:1:0' < tools/pmc_sets_mem.txt > $out/stg.log 2>&1
for w in lane stg; do echo "== $w"; grep -v "^$" $out/$w/summary.txt | sed 's/void lz4hip::decode_ring_kernel//; s/void lz4hip::decode_kernel//' | cut -c1-110; done
