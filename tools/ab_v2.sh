#!/bin/bash
# usage (on the GPU box): tools/ab_v2.sh <variant> ...  -- times lean-core variant libraries (RING env passes through); base = the built one
cd $GRAFT_REPO_ROOT
cp lz4-java_amd/liblz4hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so lz4-java_amd/liblz4hip.so; else cp lz4-java_amd/variants/$v.so lz4-java_amd/liblz4hip.so; fi
  echo "== $v"
  timeout 120 python tools/gpu_v2_check.py ${AB_N:-12800} 2>&1 | tail -7 | head -${AB_LINES:-3}
done
cp /tmp/base.so lz4-java_amd/liblz4hip.so
