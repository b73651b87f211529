#!/bin/bash
# round 6, GPU session ai: the wave loop in three instances (plain / SHORT / plain without a way back), the route's wave kernel for every batch of up to 32
# blocks per CU that is not mostly literals, the ring loop from 14336 blocks on: decoder tests, the matrix against notiers (-DLZ4HIP_RUN_TIERS=0), what the route picks
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06ai}; mkdir -p $out
tools/r06af.sh ${1:-r06ai}
timeout 900 python tools/route_sweep.py 6144,8192,10240 book,appf,pic,geo > $out/sweep.log 2>&1
for n in 8192 16384; do timeout 600 python tools/route_sweep.py $n cfg2_$n >> $out/sweep.log 2>&1; done
grep -v amdgpu $out/sweep.log | cut -c1-330
