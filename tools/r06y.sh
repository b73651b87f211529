#!/bin/bash
# round 6, GPU session y: the parallel wave loop requests a second stream step when its ring runs low (product); A/B against nocont (-DLZ4HIP_WAVE_CONT=0) and
# wp8 (-DLZ4HIP_WALK_PAR_MIN=8u: the walk by pointer doubling from 8 starts per window on); what the loop does (rdbg)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06y}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=3 --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
shapes=cfg2_2048,cfg2_4096,appf2048,appf4096,book2048,book4096,book65536,geo4096,pic4096
for rep in 1 2; do
echo "== product $rep" >> $out/matrix.log
timeout 600 python tools/ring_matrix.py $shapes 64:5:0:0 >> $out/matrix.log 2>&1
for v in nocont wp8; do
echo "== $v $rep" >> $out/matrix.log
LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/$v.so timeout 600 python tools/ring_matrix.py $shapes 64:5:0:0 >> $out/matrix.log 2>&1
done; done
grep -v amdgpu $out/matrix.log | paste - - | cut -c1-140
for v in rdbg; do for w in appf2048 cfg2_2048 cfg2_4096 book2048; do
echo "== $v $w" >> $out/stats.log
LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/$v.so timeout 300 python tools/wave_stats.py $w 2>&1 | grep -v amdgpu | tail -2 >> $out/stats.log
done; done
cat $out/stats.log
