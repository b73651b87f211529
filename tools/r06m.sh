#!/bin/bash
# round 6, GPU session m: the walk by pointer doubling for windows full of sequences (text) -- decoder tests, text / App. F / 4 MiB launches at every routing class
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06m}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=3 --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
timeout 900 python tools/ring_matrix.py book1,book512,book2048,book4096,book8192,book65536,appf1,appf2048,appf4096,cfg2_256,cfg2_2048,cfg2_4096 d > $out/matrix.log 2>&1
grep -v amdgpu $out/matrix.log
timeout 600 python tools/route_sweep.py 65536 lit2,lit8 >> $out/matrix.log 2>&1; tail -2 $out/matrix.log | cut -c1-330
