#!/bin/bash
# round 5, GPU session q: the parallel LZ4Block walk -- the stream tests, the JNI scenario, the bench line's container legs
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r05q}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_jni.py -m gpu -q -x --durations=5 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
tail -12 $out/pytest.log; tail -2 $out/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05q/bench_line.json").read().strip().splitlines()[-1])
for k in ("frame_blocks_dev","frame_read_dev","block_read_dev"):
    v=d["configs"].get(k); print(k, {kk:vv for kk,vv in (v or {}).items() if kk not in ("workload","note")})
PY
