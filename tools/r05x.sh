#!/bin/bash
# round 5, GPU session x: the lane-per-run trip of the parallel wave loop -- parity first (decoder tests, fuzz slices), then the launches it is for
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r05x}; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=5 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout 600 python tools/ring_matrix.py cfg2_256,cfg2_1024,cfg2_2048,cfg2_4096,appf256,appf2048,appf4096,book2048,book4096 d > $out/matrix.log 2>&1
cat $out/matrix.log | tail -14
