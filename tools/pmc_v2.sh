#!/bin/bash
# usage: tools/pmc_v2.sh <outdir> <n_blocks> <ring 0|1>  -- instruction-mix PMC passes of the lean fast-compress kernel (kernel-trace only)
out=$1; n=$2; export CC=3; export RING=${3:-1}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o pmc -- python tools/gpu_one.py $n 2 0 synth > $out/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
  db=$(find $out/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db compress_fast | grep -v "^$" >> $out/summary.txt
done <<SETS
SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
SETS
cat $out/summary.txt
