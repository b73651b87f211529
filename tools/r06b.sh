#!/bin/bash
# round 6, GPU session b: the pair loop (decode_pipe 7) meets the GPU -- decoder parity tests with its variants in the lists, then the launches it is for
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06b}; mkdir -p $out
timeout 180 python tools/ring_matrix.py appf64,book64 64:7:0:0,64:7:0:16384 > $out/sanity.log 2>&1; cat $out/sanity.log; grep -q "ok=True" $out/sanity.log || exit 1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=5 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
timeout 900 python tools/ring_matrix.py cfg2_256,cfg2_1024,cfg2_2048,appf256,appf2048,book2048 d,64:7:0:0 > $out/matrix.log 2>&1
tail -24 $out/matrix.log
timeout 300 python tools/ring_matrix.py appf1,book1,cfg2_8 d,64:7:0:0 > $out/matrix_one.log 2>&1
tail -12 $out/matrix_one.log
