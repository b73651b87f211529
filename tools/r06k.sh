#!/bin/bash
# round 6, GPU session k: trio with merged control-word reads (scanner: STOP + IPDONE, planner: SHEAD + entry) -- decoder tests, the small launches
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06k}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=3 --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
timeout 600 python tools/ring_matrix.py cfg2_8,cfg2_256,cfg2_512,cfg2_1024,cfg2_1280,appf1,appf256,appf1024,appf1280,book1,book512,book1280 64:8:0:0 > $out/matrix.log 2>&1
grep -v amdgpu $out/matrix.log
