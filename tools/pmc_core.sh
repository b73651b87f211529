#!/bin/bash
# usage: tools/pmc_core.sh <outdir> <n_blocks> <data> <core>   -- instruction-mix PMC passes of the fast-compress kernel (own runs, kernel-trace only)
out=$1; n=$2; data=${3:-synth}; export CC=${4:-1}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o pmc -- python tools/gpu_one.py $n 2 0 $data > $out/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
  db=$(find $out/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db compress_fast | grep -v "^$" >> $out/summary.txt
done <<SETS
SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM
SETS
cat $out/summary.txt
