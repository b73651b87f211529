#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r04h; mkdir -p $out
(export LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/rdbg.so; for w in "book65536 4 512" "appf65536 4 512" "cfg2_16384 8 1024"; do timeout 200 python tools/ring_stats.py $w 2>&1 | grep -v amdgpu | tail -3; done) > $out/stats.log 2>&1
timeout 400 python tools/ring_matrix.py appf65536,book65536,cfg2_16384 d,4:3:0:512,4:3:0:1024,8:3:0:512,8:3:0:1024,8:3:0:2048,16:3:0:4096 > $out/m_s3.log 2>&1
LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/s4.so timeout 300 python tools/ring_matrix.py appf65536,book65536,cfg2_16384 4:3:0:512,4:3:0:1024,8:3:0:1024 > $out/m_s4.log 2>&1
timeout 300 python tools/ring_matrix.py appf16384,book16384,geo32768,pic32768,cfg2_4096 d,4:3:0:512,8:3:0:1024 > $out/m_small.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "decode or deep or fuzz" > $out/pytest.log 2>&1
cat $out/stats.log; grep -v amdgpu.ids $out/m_s3.log $out/m_s4.log $out/m_small.log | sed 's/gpurun_out.r04h.//'; tail -3 $out/pytest.log
