#!/bin/bash
# round 6, GPU session c: the density route (decode_route_kernel samples the middle of 32 streams) -- its test, the crossover sweep, the pair loop as the default up to 4 blocks per CU
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06c}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -q -x --durations=5 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
timeout 900 python tools/route_sweep.py 6144,16384,65536 book,lit2,lit4,lit8,lit8w4k,appfw4k,appf,pic,geo > $out/sweep.log 2>&1
cat $out/sweep.log
timeout 600 python tools/route_sweep.py 8192 cfg2_8192 >> $out/sweep.log 2>&1
tail -3 $out/sweep.log
echo
tail -24 $out/matrix.log
# the configs[2] traffic mix without the decoder (tools/ubench/mix_cfg2.hip): what the memory system sustains where decode_ring_kernel<4, 2048> takes 83 ms
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/mix_cfg2.hip -o /tmp/mix_cfg2 && { for W in 4 8; do for D in 1 2 4 8; do for P in 61 100; do timeout 120 /tmp/mix_cfg2 $W $P $D; done; done; done; } > $out/mix_cfg2.log 2>&1
cat $out/mix_cfg2.log
