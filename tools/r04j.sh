#!/bin/bash
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/r04j; mkdir -p $out
timeout 200 python tools/ring_matrix.py cfg2_16384 d,4:3:0:2048,8:3:0:2048,8:3:0:1024,16:3:0:4096 2>&1 | grep -v amdgpu > $out/m.log
cat > /tmp/sets3.txt <<'EOT'
SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
EOT
bash tools/pmc_run.sh $out/ring decode_ring 'python tools/ring_matrix.py cfg2_16384 4:3:0:2048' < /tmp/sets3.txt > $out/ring.log 2>&1
bash tools/pmc_run.sh $out/deep decode_deep 'python tools/ring_matrix.py cfg2_16384 d' < /tmp/sets3.txt > $out/deep.log 2>&1
cat $out/m.log; for w in ring deep; do echo "== $w"; grep -v "^$" $out/$w/summary.txt | sed 's/void lz4hip::decode_ring_kernel//; s/void lz4hip::decode_deep_kernel//' | cut -c1-100; done
