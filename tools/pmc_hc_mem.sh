#!/bin/bash
# usage: tools/pmc_hc_mem.sh <outdir> <n_blocks> <block_bytes> [level]   -- memory-side PMC passes of hc_parse_kernel (own runs, kernel-trace only)
out=$1; n=$2; blk=$3; lvl=${4:-9}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o pmc -- python tools/gpu_hc_probe.py $n $blk $lvl > $out/p$i.log 2>&1
  db=$(find $out/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db hc_parse | grep -v "^$" | grep -v "^##\|^kernel " >> $out/summary.txt
  rm -rf $out/p$i
done <<SETS
FETCH_SIZE
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INSTS_LDS
TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
SETS
cat $out/summary.txt
