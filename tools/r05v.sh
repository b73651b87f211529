#!/bin/bash
# round 5, GPU session v: volume checks of the wave loops (long streams intact and damaged, every ring; the default routing on small launches)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/r05v; mkdir -p $out
(for pr in "5 0" "5 8192" "5 16384" "5 32768" "5 65536" "4 0" "4 8192"; do set -- $pr; FUZZ_PIPE=$1 FUZZ_RING=$2 timeout 600 python tools/gpu_fuzz_deep.py 1500 $((500 + $1 * 7 + $2 / 4096)) 2>&1 | tail -2; done) > $out/fuzz_wave.log 2>&1
timeout 600 python tools/gpu_small_batches.py 300 > $out/small_batches.log 2>&1
timeout 900 python tools/gpu_fuzz.py 6000 77 > $out/gpu_fuzz.log 2>&1
cat $out/fuzz_wave.log; tail -3 $out/small_batches.log; tail -6 $out/gpu_fuzz.log
timeout 280 python tools/gpu_ring_edge.py 400 11 2>&1 | tail -1 >> $out/fuzz_wave.log
