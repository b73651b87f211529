#!/bin/bash
# round-4 GPU session C: ring loop v3 (near source read when the piece goes in; 2 / 3 / 4 slots)
cd "$(dirname "$0")/.."
out=gpurun_out/r04c; mkdir -p $out
timeout 400 python tools/ring_matrix.py appf65536,book65536,cfg2_16384 d,4:3:0:512,4:3:0:1024,8:3:0:1024,16:3:0:4096,8:3:0:4096 > $out/m_s3.log 2>&1
for v in s2 s4; do
  LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/$v.so timeout 300 python tools/ring_matrix.py appf65536,book65536,cfg2_16384 4:3:0:512,4:3:0:1024,16:3:0:4096,8:3:0:4096 > $out/m_$v.log 2>&1
done
timeout 300 python tools/ring_matrix.py appf16384,book16384,geo32768,pic32768,cfg2_4096 d,4:3:0:1024,8:3:0:1024,16:3:0:4096 > $out/m_small.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "decode or deep or fuzz" > $out/pytest.log 2>&1
grep -v amdgpu.ids $out/m_s3.log $out/m_s2.log $out/m_s4.log $out/m_small.log; tail -5 $out/pytest.log
