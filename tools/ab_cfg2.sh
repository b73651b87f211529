#!/bin/bash
# usage (on the GPU box): tools/ab_cfg2.sh <variant> ...  -- 4 MiB decode (BASELINE configs[2]) + headline decode with variant libraries
cd $GRAFT_REPO_ROOT
cp lz4-java_amd/liblz4hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so lz4-java_amd/liblz4hip.so; else cp lz4-java_amd/variants/$v.so lz4-java_amd/liblz4hip.so; fi
  echo "== $v"
  timeout 300 python tools/gpu_cfg3.py ${AB_N:-16384} 1 2>&1 | tail -1
  for d in synth book1; do echo -n "$d: "; timeout 120 python tools/gpu_one.py 65536 2 0 $d 2>&1 | tail -1; done
done
cp /tmp/base.so lz4-java_amd/liblz4hip.so
