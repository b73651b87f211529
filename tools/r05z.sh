#!/bin/bash
# round 5 GPU session z: the whole gpu suite, smoke, the committed evidence (tools/collect_profiles.sh <tag>: kernel traces, PMC traffic passes incl. the
# text legs, the bench line), the mix micro-benchmark
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
tag=${1:-r05a}
out=gpurun_out/r05z; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
bash tools/collect_profiles.sh $tag > $out/collect.log 2>&1
tail -5 $out/pytest.log; tail -2 $out/smoke.log; tail -3 $out/collect.log | cut -c1-1500
