#!/bin/bash
# round 6, GPU session f: the route's sampling position for the fast decoder -- the scale tests, the bench line
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06f}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -x --durations=5 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
timeout 900 python bench.py > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/bench_line.json; python - <<PYEOF
import json
l = json.load(open("$out/bench_line.json"))
print("value", l["value"], "verified", l["verified"], "decompress_fast", l["configs"]["decompress_fast"]["value"], l["configs"]["decompress_fast"]["roofline"]["kernel"])
PYEOF
