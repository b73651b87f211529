#!/bin/bash
# round 6, GPU session c: the density route (decode_route_kernel samples the middle of 32 streams) -- its test, the crossover sweep, the pair loop as the default up to 4 blocks per CU
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06c}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -q -x --durations=5 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
timeout 900 python tools/route_sweep.py 6144,16384,65536 book,lit2,lit8,lit8w4k,lit16w4k,appfw4k,appf,pic,geo > $out/sweep.log 2>&1
cat $out/sweep.log
timeout 600 python tools/route_sweep.py 8192 cfg2_8192 >> $out/sweep.log 2>&1
tail -3 $out/sweep.log
echo
tail -24 $out/matrix.log
