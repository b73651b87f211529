#!/bin/bash
# round 6, GPU session h: the trio loop at 5 .. 8 blocks per CU (five trios in one workgroup; two workgroups of four trios with 8 KB rings) against the wave loop
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06h}; mkdir -p $out
timeout 180 python tools/ring_matrix.py appf64,book64 64:8:0:8192,64:8:0:16384 > $out/sanity.log 2>&1; grep -v amdgpu $out/sanity.log; grep -q "ok=True" $out/sanity.log || exit 1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=5 --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout 900 python tools/ring_matrix.py cfg2_1280,cfg2_1536,cfg2_2048,cfg2_4096,appf1280,appf2048,appf4096,book1280,book2048,book4096 64:5:0:0,64:8:0:16384,64:8:0:8192 > $out/matrix.log 2>&1
grep -v amdgpu $out/matrix.log
