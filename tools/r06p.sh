#!/bin/bash
# round 6, GPU session p: what the device-side route costs a launch it does not change (default routing against the same kernel forced, one process)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06p}; mkdir -p $out
timeout 600 python tools/ring_matrix.py appf65536,appf6144,appf16384,geo65536 d,4:0:1:0,8:2:0:0,d,4:0:1:0,8:2:0:0 > $out/tax.log 2>&1
grep -v amdgpu $out/tax.log
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -q -x -k "routed" --timeout 600 2>&1 | tail -2
