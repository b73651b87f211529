#!/bin/bash
# round 6, GPU session j: the trio copier's flusher in front of its wait + HEAD and slot in one round trip; the parallel wave loop's flusher at the top of the trip (variant ftop)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06j}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=3 --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
echo "== product" > $out/matrix.log
timeout 600 python tools/ring_matrix.py cfg2_8,cfg2_256,cfg2_1024,cfg2_1280,appf1,appf1024,book1,book512 64:8:0:0 >> $out/matrix.log 2>&1
timeout 600 python tools/ring_matrix.py cfg2_2048,cfg2_4096,appf2048,appf4096,book2048,book4096 64:5:0:0 >> $out/matrix.log 2>&1
echo "== ftop" >> $out/matrix.log
LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/ftop.so timeout 600 python tools/ring_matrix.py cfg2_2048,cfg2_4096,appf2048,appf4096,book2048,book4096 64:5:0:0 >> $out/matrix.log 2>&1
LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/ftop.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 > $out/pytest_ftop.log 2>&1; echo "pytest ftop rc=$?" >> $out/pytest_ftop.log
tail -2 $out/pytest_ftop.log
grep -v amdgpu $out/matrix.log
