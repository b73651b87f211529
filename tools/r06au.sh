#!/bin/bash
# round 6, GPU session au: the route's crossover between the wave kernel and the deep loop again, with self-overlapping matches copied inside the passes
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06au}; mkdir -p $out
timeout 1200 python tools/route_sweep.py 8192,10240,12288,16384,65536 appf,appfw4k,pic,lit8,geo > $out/sweep.log 2>&1
for n in 10240 12288; do timeout 600 python tools/route_sweep.py $n cfg2_$n >> $out/sweep.log 2>&1; done
grep -v amdgpu $out/sweep.log | cut -c1-330
