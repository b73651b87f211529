#!/bin/bash
# round 6, GPU session al: volume checks of the one-wavefront parallel loop at its final form (segments, per-sequence end rules, SHORT instance): long streams intact and
# damaged at every ring, ring-edge / wild-piece streams, hundreds of small launches and the corpus fuzz through the default routing
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06al}; mkdir -p $out
(for pr in "5 0" "5 8192" "5 16384" "5 32768" "5 65536" "8 0"; do set -- $pr; FUZZ_PIPE=$1 FUZZ_RING=$2 timeout 600 python tools/gpu_fuzz_deep.py 2000 $((1900 + $1 * 7 + $2 / 4096)) 2>&1 | tail -2; done) > $out/fuzz_wave.log 2>&1
timeout 400 python tools/gpu_ring_edge.py 400 29 2>&1 | tail -1 >> $out/fuzz_wave.log
timeout 600 python tools/gpu_small_batches.py 300 > $out/small_batches.log 2>&1
timeout 900 python tools/gpu_fuzz.py 6000 79 > $out/gpu_fuzz.log 2>&1
cat $out/fuzz_wave.log; tail -3 $out/small_batches.log; tail -6 $out/gpu_fuzz.log
