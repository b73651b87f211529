#!/bin/bash
# round 6, GPU session w: the parallel wave loop goes on in the same window after a cut pass / a one-sequence step (LZ4HIP_WAVE_CONT; variant nocont = before)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06w}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=3 --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
shapes=cfg2_2048,cfg2_4096,appf2048,appf4096,book2048,book4096,book8192,book65536
for rep in 1 2; do
echo "== product (cont) $rep" >> $out/matrix.log
timeout 600 python tools/ring_matrix.py $shapes 64:5:0:0 >> $out/matrix.log 2>&1
echo "== nocont $rep" >> $out/matrix.log
LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/nocont.so timeout 600 python tools/ring_matrix.py $shapes 64:5:0:0 >> $out/matrix.log 2>&1
done
grep -v amdgpu $out/matrix.log
