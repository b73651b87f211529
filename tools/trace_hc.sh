#!/bin/bash
# usage: tools/trace_hc.sh <outdir> <n_blocks> <block_bytes> [level]   -- per-kernel times of one HC probe run
out=$1; n=$2; blk=$3; lvl=${4:-9}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/t -o tr -- python tools/gpu_hc_probe.py $n $blk $lvl > $out/t.log 2>&1
db=$(find $out/t -name "*.db" | head -1)
python tools/rocprof_summary.py $db hc_ | grep -v "^$"
rm -rf $out/t
