#!/bin/bash
# round 5, GPU session k: the whole gpu suite with the new default routing (a wavefront per block up to 16 blocks per CU), smoke, bench line
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r05k}; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
tail -14 $out/pytest.log; tail -2 $out/smoke.log; tail -3 $out/bench.err; python - <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/%s/bench_line.json" % (sys.argv[1] if len(sys.argv)>1 else "r05k")).read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"])
    c=d.get("configs",{})
    for k in ("configs2_decode_4MiB","configs2_shard8","configs2_shard8_compress","decode_small_launches","compress_4MiB","real_book1","frame_read_dev","headline_blocks_vs_reference"):
        v=c.get(k); print(k, json.dumps({kk:vv for kk,vv in (v or {}).items() if kk not in ("workload","cpu_baseline","roofline","note")})[:400])
except Exception as e: print("no bench line", e)
PY
