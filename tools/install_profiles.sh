#!/bin/bash
# usage: tools/install_profiles.sh <old tag> <new tag> -- after `gpurun -- 'tools/r06e.sh <new tag>'`: copy the collection's files from gpurun_out/ into
# profiles/ (bench line, kernel traces, PMC summary, traffic.json, shard_prediction.json, the GPU suite's log), remove the old tag's files, rename the tag in the docs.
# Refuses when the collection was not taken at the sources in the tree (traffic.json's kernel_source_hash).
set -e
cd "$(dirname "$0")/.."
old=$1; new=$2
have=$(python -c "import bench; print(bench.kernel_source_hash())")
got=$(python -c "import json; print(json.load(open('gpurun_out/traffic_$new.json'))['kernel_source_hash'])")
[ "$have" = "$got" ] || { echo "gpurun_out/traffic_$new.json was measured at $got, the tree is $have"; exit 1; }
grep -q "pytest rc=0" gpurun_out/$new/pytest.log && grep -q "smoke rc=0" gpurun_out/$new/smoke.log || { echo "suite or smoke not green"; exit 1; }
for f in bench_line.json bench_kernel_trace.txt bench_all_configs_kernel_trace.txt traffic_pmc.txt; do
  sed "1s#${old}_#${new}_#" gpurun_out/${new}_$f > profiles/${new}_$f
done
sed "s#profiles/${old}_#profiles/${new}_#" gpurun_out/traffic_$new.json > profiles/traffic.json
sed "s#profiles/${old}_#profiles/${new}_#" gpurun_out/shard_prediction_$new.json > profiles/shard_prediction.json
{ echo "# profiles/${new}_gpu_suite.txt -- round 6, GPU session ${new#r06} (tools/r06e.sh $new; 1x MI355X): whole GPU suite + smoke at the FINAL sources (profiles/traffic.json's kernel_source_hash $have)"
  tail -14 gpurun_out/$new/pytest.log; grep -v amdgpu gpurun_out/$new/smoke.log; } > profiles/${new}_gpu_suite.txt
if [ "$old" != "$new" ]; then
  git rm -q -f profiles/${old}_bench_line.json profiles/${old}_bench_kernel_trace.txt profiles/${old}_bench_all_configs_kernel_trace.txt profiles/${old}_traffic_pmc.txt profiles/${old}_gpu_suite.txt
  sed -i "s/${old}_/${new}_/g" DESIGN.md README.md profiles/README.md
fi
git add profiles/${new}_* profiles/traffic.json profiles/shard_prediction.json
echo "installed $new (was $old) at $have"
