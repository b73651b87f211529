#!/bin/bash
# round 6, GPU session v: volume checks of the pair and trio loops (long streams intact and damaged, every ring; ring-edge / wild-piece streams; the default routing --
# trio up to 5 blocks per CU -- on hundreds of small launches and on the corpus fuzz)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06v}; mkdir -p $out
(for pr in "8 0" "8 8192" "8 16384" "8 32768" "8 65536" "7 0" "7 16384" "7 65536"; do set -- $pr; FUZZ_PIPE=$1 FUZZ_RING=$2 timeout 600 python tools/gpu_fuzz_deep.py 1500 $((900 + $1 * 7 + $2 / 4096)) 2>&1 | tail -2; done) > $out/fuzz_pair_trio.log 2>&1
RING_EDGE_PIPES=78 timeout 400 python tools/gpu_ring_edge.py 400 23 2>&1 | tail -1 >> $out/fuzz_pair_trio.log
timeout 600 python tools/gpu_small_batches.py 300 > $out/small_batches.log 2>&1
timeout 900 python tools/gpu_fuzz.py 6000 78 > $out/gpu_fuzz.log 2>&1
cat $out/fuzz_pair_trio.log; tail -3 $out/small_batches.log; tail -6 $out/gpu_fuzz.log
