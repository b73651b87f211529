#!/bin/bash
# round 5, GPU session b: the wave loop after a change -- the small-batch matrix only (tools/r05a.sh has the tests)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r05b}; mkdir -p $out
timeout 600 python tools/ring_matrix.py cfg2_256,cfg2_2048,cfg2_4096,cfg2_8192 d,64:4:0:0,64:4:0:8192 > $out/matrix_4MiB.log 2>&1
timeout 400 python tools/ring_matrix.py appf1,appf512,appf2048,appf4096,book512,book2048 d,64:4:0:0,64:4:0:8192 > $out/matrix_64k.log 2>&1
cat $out/matrix_4MiB.log $out/matrix_64k.log | grep -v "^$\|amdgpu.ids" | cut -c1-200
