#!/bin/bash
# round 6, GPU session ak: two decoders at once on the headline's blocks (tools/concurrent_mix.py): the staged loop is memory-bound, the wave kernel issue-bound
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06ak}; mkdir -p $out
timeout 900 python tools/concurrent_mix.py appf 65536 > $out/mix.log 2>&1
grep -v amdgpu $out/mix.log
