#!/bin/bash
# round 6, GPU session a: the lean discovery back in the product -- whole GPU suite, smoke, the wave kernel's launches, the default bench line
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06a}; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log; tail -2 $out/smoke.log
timeout 600 python tools/ring_matrix.py cfg2_256,cfg2_1024,cfg2_2048,cfg2_4096,appf256,appf2048,appf4096,book2048,book4096 d > $out/matrix.log 2>&1
tail -14 $out/matrix.log
timeout 900 python bench.py > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/bench_line.json; tail -c 1500 $out/bench.log
