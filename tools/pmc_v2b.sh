#!/bin/bash
# usage: tools/pmc_v2b.sh <outdir> <n_blocks> <core>  -- instruction-cache / latency PMC passes of a fast-compress core
out=$1; n=$2; export CC=${3:-3}; export RING=0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o pmc -- python tools/gpu_one.py $n 2 0 synth > $out/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
  db=$(find $out/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db compress_fast | grep -v "^$" | grep -v "kernel-trace\|^kernel \|calls" >> $out/summary.txt
done <<SETS
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS
SETS
cat $out/summary.txt
