#!/bin/bash
# usage (on the GPU box): tools/collect_config_profiles.sh <tag>  -- rocprofv3 --kernel-trace --stats summaries of the BASELINE.json configs
# that bench.py does not time: configs[2] (4 MiB-block decode), configs[3] (HC level 9, 4096 x 1 MiB), configs[4] (xxhash 1 Mi x 4 KiB).
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/${tag}_configs_kernel_trace.txt
echo "# profiles/${tag}_configs_kernel_trace.txt -- rocprofv3 --kernel-trace --stats (1x MI355X; summaries by tools/rocprof_summary.py)" > $out
run() {  # name, filter, command...
  name=$1; filt=$2; shift 2
  rm -rf gpurun_out/cfgprof
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/cfgprof -o kt -- "$@" > gpurun_out/cfgprof.log 2>&1
  echo "## $name: $*" >> $out
  grep -v "amdgpu.ids\|^W2\|^E2\|rocprofv3" gpurun_out/cfgprof.log | tail -6 >> $out
  db=$(find gpurun_out/cfgprof -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db $filt | head -8 >> $out
  rm -rf gpurun_out/cfgprof gpurun_out/cfgprof.log
}
run "configs[2] safe decompress, 4096 x 4 MiB x 4 passes" decode_kernel python tools/gpu_cfg3.py 4096 4
run "configs[3] HC level 9, 4096 x 1 MiB" hc_ python tools/gpu_hc_probe.py 4096 1048576 9
run "configs[4] XXH32 / XXH64, 1 Mi x 4 KiB" xxh python tools/xxh_cfg5.py
cat $out
