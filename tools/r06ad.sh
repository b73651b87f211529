#!/bin/bash
# round 6, GPU session ad: the decode route's crossovers again, with the wave loop of sessions w .. ac (segments, end rules, short round forms)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06ad}; mkdir -p $out
timeout 1200 python tools/route_sweep.py 6144,8192,12288,16384,65536 book,lit2,lit8,lit8w4k,appfw4k,appf,pic,geo > $out/sweep.log 2>&1
for n in 5000 6144 8192 10240 12288; do timeout 600 python tools/route_sweep.py $n cfg2_$n >> $out/sweep.log 2>&1; done
grep -v amdgpu $out/sweep.log
