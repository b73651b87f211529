#!/bin/bash
# round-4 GPU session A: ring-loop matrix first, then the whole gpu suite
cd "$(dirname "$0")/.."
out=gpurun_out/r04a; mkdir -p $out
timeout 500 python tools/ring_matrix.py appf65536,book65536,appf16384,book16384,geo32768,pic32768 d,4:3:0:512,4:3:0:1024,8:3:0:512,8:3:0:1024 > $out/m1.log 2>&1
echo "m1 rc=$?" >> $out/m1.log
timeout 500 python tools/ring_matrix.py cfg2_16384,cfg2_4096 d,8:3:0:4096,16:3:0:4096,8:3:0:2048,16:3:0:2048,8:3:0:512 > $out/m2.log 2>&1
echo "m2 rc=$?" >> $out/m2.log
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/m1.log $out/m2.log; tail -30 $out/pytest.log
