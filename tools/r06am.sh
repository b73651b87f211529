#!/bin/bash
# round 6, GPU session am: the wave / pair / trio kernels' wavefronts take the blocks blockIdx + k * grid (a batch that is not a multiple of W x CUs spreads over all CUs);
# A/B against premap (the build before: blockIdx * W + wave) on ragged batch sizes, and on the multiples the bench uses
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06am}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=3 --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
shapes=cfg2_300,cfg2_768,cfg2_1100,cfg2_1536,cfg2_2048,cfg2_2560,cfg2_3072,cfg2_4096,appf700,appf1800,appf2300,appf3000,appf4096,book3000
for rep in 1 2; do
echo "== product $rep" >> $out/matrix.log
timeout 900 python tools/ring_matrix.py $shapes d >> $out/matrix.log 2>&1
echo "== premap $rep" >> $out/matrix.log
LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/premap.so timeout 900 python tools/ring_matrix.py $shapes d >> $out/matrix.log 2>&1
done
python tools/matrix_table.py $out/matrix.log
