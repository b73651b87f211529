#!/bin/bash
# usage (on the GPU box): tools/ubench/run_mix.sh <out.txt>  -- the headline decoder's traffic mix without the decoder (tools/ubench/mix.hip)
out=${1:-gpurun_out/mix.txt}
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/mix.hip -o /tmp/mix || exit 1
{
echo "# tools/ubench/mix.hip on 1x MI355X: sequential stream reads (0.5 B/B) + sequential 64-byte output stores (1 B/B) + R random 64-byte reads per"
echo "# 64 bytes of output inside the 64 KiB behind the block's write pointer (1.5 lines each), 4 lanes per block, 16 blocks per wavefront, 65536 blocks x 64 KiB"
for R in 2 1 3; do for D in 1 2 4; do for W in 4 8 12 16; do timeout 120 /tmp/mix $W $R $D; done; done; done
} > $out 2>&1
cat $out
