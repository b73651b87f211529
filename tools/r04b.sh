#!/bin/bash
# round-4 GPU session B: the branch-free ring loop
cd "$(dirname "$0")/.."
out=gpurun_out/r04b; mkdir -p $out
timeout 400 python tools/ring_matrix.py appf65536,book65536,appf16384,book16384,geo32768,pic32768 d,4:3:0:512,4:3:0:1024,8:3:0:512,8:3:0:1024 > $out/m1.log 2>&1
timeout 400 python tools/ring_matrix.py cfg2_16384,cfg2_4096 d,8:3:0:4096,16:3:0:4096,16:3:0:2048,8:3:0:512 > $out/m2.log 2>&1
for v in p8 p2; do
  LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/$v.so timeout 300 python tools/ring_matrix.py appf65536,book65536,cfg2_16384 4:3:0:512,16:3:0:4096 > $out/m_$v.log 2>&1
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "decode or deep or fuzz" > $out/pytest.log 2>&1
tail -40 $out/m1.log $out/m2.log $out/m_p8.log $out/m_p2.log; tail -5 $out/pytest.log
