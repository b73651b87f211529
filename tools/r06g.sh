#!/bin/bash
# round 6, GPU session g: the trio loop (decode_pipe 8: scanner / planner / copier wavefronts per block) meets the GPU
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06g}; mkdir -p $out
timeout 180 python tools/ring_matrix.py appf64,book64 64:8:0:0,64:8:0:32768 > $out/sanity.log 2>&1; cat $out/sanity.log | grep -v amdgpu; grep -q "ok=True" $out/sanity.log || exit 1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=5 --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
timeout 600 python tools/ring_matrix.py cfg2_8,cfg2_256,cfg2_512,cfg2_1024,appf1,appf256,appf512,appf1024,book1,book512 64:5:0:0,64:7:0:0,64:8:0:0 > $out/matrix.log 2>&1
grep -v amdgpu $out/matrix.log
