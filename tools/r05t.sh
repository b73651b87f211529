#!/bin/bash
# round 5, GPU session t: where the wave kernel stops winning -- 6144 / 8192 blocks, default routing (the lane-group loops) against the wave kernel forced
cd "$(dirname "$0")/.."
out=gpurun_out/${1:-r05t}; mkdir -p $out
timeout 900 python tools/ring_matrix.py cfg2_6144,cfg2_8192,appf6144,appf8192,appf12288,book6144,book8192 d,64:5:0:0 > $out/matrix.log 2>&1
cat $out/matrix.log | tail -24
