#!/bin/bash
# round-4 final GPU session: volume fuzz of the ring loop, the whole gpu suite, smoke, the committed evidence (tools/collect_profiles.sh)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/r04z; mkdir -p $out
(FUZZ_PIPE=3 timeout 300 python tools/gpu_fuzz_deep.py 1500 91; FUZZ_PIPE=3 FUZZ_RING=512 timeout 300 python tools/gpu_fuzz_deep.py 1000 92; FUZZ_PIPE=3 FUZZ_RING=2048 timeout 300 python tools/gpu_fuzz_deep.py 1000 93) > $out/fuzz_ring.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
bash tools/collect_profiles.sh r04a > $out/collect.log 2>&1
tail -4 $out/fuzz_ring.log; tail -5 $out/pytest.log; tail -2 $out/smoke.log; tail -3 $out/collect.log | cut -c1-600
