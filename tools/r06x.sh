#!/bin/bash
# round 6, GPU session x: the parallel wave loop with the trio's per-sequence end rules + the stream's partial last step (product), going on in the same
# window after a cut (LZ4HIP_WAVE_CONT; variant nocont = end rules only); what the loop does with either (rdbg / rdbg_nocont: -DLZ4HIP_RING_DBG builds)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06x}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=3 --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
shapes=cfg2_2048,cfg2_4096,appf2048,appf4096,book2048,book4096,book8192,book65536
for rep in 1 2; do
echo "== product (end rules + cont) $rep" >> $out/matrix.log
timeout 600 python tools/ring_matrix.py $shapes 64:5:0:0 >> $out/matrix.log 2>&1
echo "== nocont (end rules) $rep" >> $out/matrix.log
LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/nocont.so timeout 600 python tools/ring_matrix.py $shapes 64:5:0:0 >> $out/matrix.log 2>&1
done
grep -v amdgpu $out/matrix.log
for v in rdbg rdbg_nocont; do for w in appf2048 cfg2_2048 book2048; do
echo "== $v $w" >> $out/stats.log
LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/$v.so timeout 300 python tools/wave_stats.py $w 2>&1 | grep -v amdgpu | tail -2 >> $out/stats.log
done; done
cat $out/stats.log
