#!/bin/bash
# usage (on the GPU box): tools/ab_hc.sh <variant> ...  -- times HC level 9 on 1 MiB blocks with variant libraries; base = the built one
cd $GRAFT_REPO_ROOT
cp lz4-java_amd/liblz4hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so lz4-java_amd/liblz4hip.so; else cp lz4-java_amd/variants/$v.so lz4-java_amd/liblz4hip.so; fi
  echo "== $v"
  for n in ${AB_N:-4096 8192}; do timeout 120 python tools/gpu_hc_probe.py $n ${AB_BLK:-1048576} 9 2>&1 | tail -1; done
done
cp /tmp/base.so lz4-java_amd/liblz4hip.so
