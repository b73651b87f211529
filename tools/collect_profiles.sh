#!/bin/bash
# usage (on the GPU box): tools/collect_profiles.sh <tag>     e.g. r01i
# Writes into gpurun_out/: <tag>_bench_line.json, <tag>_bench_kernel_trace.txt (rocprofv3 --kernel-trace --stats of bench.py),
# <tag>_traffic_pmc.txt + traffic_<tag>.json (separate --pmc passes, tools/traffic_passes.sh).  Copy them to profiles/.
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# headline launches only (--no-extra-configs): the per-kernel averages of this trace are the ones bench.py's HIP events must agree with
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic --no-extra-configs > gpurun_out/prof_$tag.log 2>&1
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
(echo "# profiles/${tag}_bench_kernel_trace.txt"
 echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic --no-extra-configs   (1x MI355X; summary by tools/rocprof_summary.py)"
 python tools/rocprof_summary.py $db
 echo "# the line that run printed:"
 tail -1 gpurun_out/prof_$tag.log) > gpurun_out/${tag}_bench_kernel_trace.txt
# the whole default command (all configs): one trace
rm -rf gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic > gpurun_out/prof_$tag.log 2>&1
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
(echo "# profiles/${tag}_bench_all_configs_kernel_trace.txt"
 echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic   (all configs; 1x MI355X)"
 python tools/rocprof_summary.py $db lz4hip) > gpurun_out/${tag}_bench_all_configs_kernel_trace.txt
tools/traffic_passes.sh gpurun_out/traffic_$tag 2 > /dev/null
tools/traffic_passes.sh gpurun_out/traffic_text_$tag 2 text > /dev/null     # the real-text legs by themselves (tools/gpu_text_legs.py)
python tools/traffic_json.py gpurun_out/traffic_$tag 65536 65536 "profiles/${tag}_traffic_pmc.txt (tools/traffic_passes.sh: separate rocprofv3 --pmc passes of bench.py, 1x MI355X)" gpurun_out/traffic_text_$tag
cp profiles/traffic.json gpurun_out/traffic_$tag.json
(echo "# profiles/${tag}_traffic_pmc.txt"
 echo "# command: tools/traffic_passes.sh (one rocprofv3 --kernel-trace --pmc <set> run per counter set; bench.py --steps 2 --warmup 1 --no-cpu-baseline, all configs)"
 for t in gpurun_out/traffic_$tag/*/; do db=$(find $t -name "*.db" | head -1); python tools/rocprof_summary.py $db lz4hip | grep -v "^$"; done
 echo "# the real-text legs alone (tools/gpu_text_legs.py):"
 for t in gpurun_out/traffic_text_$tag/*/; do db=$(find $t -name "*.db" | head -1); python tools/rocprof_summary.py $db lz4hip | grep -v "^$"; done) > gpurun_out/${tag}_traffic_pmc.txt
rm -rf gpurun_out/prof_$tag gpurun_out/traffic_$tag gpurun_out/traffic_text_$tag
# the bench line last: it quotes profiles/traffic.json, which the passes above have just rewritten
python bench.py 2>&1 | tail -1 > gpurun_out/${tag}_bench_line.json
cat gpurun_out/${tag}_bench_line.json
# what one GPU delivers on the shard of each rank count (configs2_shards_per_gpu_GBps of the line): the prediction `bench.py --gpus N` prints beside its measurement
python - <<PYEOF
import json
l = json.load(open("gpurun_out/${tag}_bench_line.json"))
sh = (l.get("configs") or {}).get("configs2_shards_per_gpu_GBps")
if sh:
    json.dump({"configs2_decode_per_gpu_GBps": sh, "source": "profiles/${tag}_bench_line.json (1x MI355X: each rank count's shard of BASELINE configs[2] decoded on one GPU)"},
              open("profiles/shard_prediction.json", "w"), indent=1, sort_keys=True)
    json.dump(json.load(open("profiles/shard_prediction.json")), open("gpurun_out/shard_prediction_${tag}.json", "w"), indent=1, sort_keys=True)
PYEOF
