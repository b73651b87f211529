#!/bin/bash
# round-4 final GPU session: volume fuzz (ring decoder loop, byU32 compress), the whole gpu suite, smoke, the committed evidence (tools/collect_profiles.sh <tag>)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
tag=${1:-r04b}
out=gpurun_out/r04z; mkdir -p $out
(FUZZ_PIPE=3 timeout 300 python tools/gpu_fuzz_deep.py 1000 91; FUZZ_PIPE=3 FUZZ_RING=2048 timeout 300 python tools/gpu_fuzz_deep.py 800 93) > $out/fuzz_ring.log 2>&1
(U32_TIMING=0 timeout 600 python tools/gpu_fuzz_u32.py 1500 101 1200000; U32_TIMING=0 timeout 300 python tools/gpu_fuzz_u32.py 40 102 6000000) > $out/fuzz_u32.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
bash tools/collect_profiles.sh $tag > $out/collect.log 2>&1
tail -3 $out/fuzz_ring.log; grep -v amdgpu $out/fuzz_u32.log | tail -4; tail -5 $out/pytest.log; tail -2 $out/smoke.log; tail -3 $out/collect.log | cut -c1-600
