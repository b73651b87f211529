#!/bin/bash
# round 6, GPU session aa: the copy rounds in three forms by their longest run (LZ4HIP_RUN_TIERS: < 16 / < 32 / <= 64 bytes; group_dev.h vcopy_run: wave loop, pair and trio copiers);
# this session: overlapping matches inside the passes of the TRIO too (planner marks the round, copier copies wave-wide) against notrioovl (-DLZ4HIP_WAVE_OVL=0: neither loop)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06af}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=3 --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
shapes=cfg2_2048,cfg2_4096,appf2048,appf4096,book2048,book4096,book65536,geo4096,pic4096
small=cfg2_256,cfg2_1024,appf1,appf512,book1,book512
for rep in 1 2; do
echo "== product $rep" >> $out/matrix.log
timeout 600 python tools/ring_matrix.py $shapes 64:5:0:0 >> $out/matrix.log 2>&1
timeout 600 python tools/ring_matrix.py $small 64:8:0:0 >> $out/matrix.log 2>&1
for v in notrioovl; do
echo "== $v $rep" >> $out/matrix.log
LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/$v.so timeout 600 python tools/ring_matrix.py $shapes 64:5:0:0 >> $out/matrix.log 2>&1
LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/$v.so timeout 600 python tools/ring_matrix.py $small 64:8:0:0 >> $out/matrix.log 2>&1
done; done
grep -v amdgpu $out/matrix.log | paste - - | cut -c1-140
