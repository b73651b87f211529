#!/bin/bash
# usage: tools/pmc_decode.sh <outdir> <n_blocks> <data>   -- instruction-mix PMC passes of the decode kernel (own runs, kernel-trace only)
out=$1; n=$2; data=${3:-book1}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o pmc -- python tools/gpu_one.py $n 2 0 $data > $out/p$i.log 2>&1
  db=$(find $out/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db decode_kernel | grep -v "^$" >> $out/summary.txt
  rm -rf $out/p$i
done <<SETS
SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INSTS_LDS
SETS
cat $out/summary.txt
