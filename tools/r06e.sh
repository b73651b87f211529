#!/bin/bash
# round 6, GPU session e: checkpoint at the current sources -- whole GPU suite, smoke, the evidence collection (bench line, kernel traces, PMC traffic passes)
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
tag=${1:-r06fin5}
out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -14 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log; tail -2 $out/smoke.log
grep -q "pytest rc=0" $out/pytest.log || exit 1
timeout 2400 tools/collect_profiles.sh $tag > $out/collect.log 2>&1; echo "collect rc=$?" >> $out/collect.log
tail -c 600 $out/collect.log
cp profiles/traffic.json gpurun_out/traffic_$tag.json 2>/dev/null
cp profiles/shard_prediction.json gpurun_out/shard_prediction_$tag.json 2>/dev/null
