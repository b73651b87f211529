#!/bin/bash
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/r04g; mkdir -p $out
head -3 tools/pmc_sets_ring.txt > /tmp/sets.txt
bash tools/pmc_run.sh $out/book decode_ring 'python tools/ring_matrix.py book65536 4:3:0:512' < /tmp/sets.txt > $out/book.log 2>&1
bash tools/pmc_run.sh $out/appf decode_ring 'python tools/ring_matrix.py appf65536 4:3:0:512' < /tmp/sets.txt > $out/appf.log 2>&1
bash tools/pmc_run.sh $out/cfg2 decode_ring 'python tools/ring_matrix.py cfg2_16384 8:3:0:1024' < /tmp/sets.txt > $out/cfg2.log 2>&1
for w in book appf cfg2; do echo "== $w"; grep -E "Args\)|chArgs|WAVE_CYCLES|WAIT_ANY|WAIT_INST_ANY|ACTIVE_INST_ANY|INSTS_VALU|INSTS_SALU|INSTS_LDS|INSTS_BRANCH|INSTS_VMEM|LDS_BANK|LDS_IDX|UNALIGNED|WAIT_INST_LDS|ACTIVE_INST_LDS|SQ_WAVES|BUSY" $out/$w/summary.txt | sed 's/void lz4hip::decode_ring_kernel//' ; done
