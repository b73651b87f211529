#!/bin/bash
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/r04e; mkdir -p $out
bash tools/pmc_run.sh $out/book decode_ring 'python tools/ring_matrix.py book16384 4:3:0:512' < tools/pmc_sets_ring.txt > $out/book.log 2>&1
bash tools/pmc_run.sh $out/cfg2 decode_ring 'python tools/ring_matrix.py cfg2_4096 16:3:0:4096' < tools/pmc_sets_ring.txt > $out/cfg2.log 2>&1
bash tools/pmc_run.sh $out/stg "decode_kernel" 'python tools/ring_matrix.py book16384 4:0:1:0' < tools/pmc_sets_ring.txt > $out/stg.log 2>&1
cat $out/book/summary.txt $out/cfg2/summary.txt $out/stg/summary.txt
