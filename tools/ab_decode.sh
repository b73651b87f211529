#!/bin/bash
# usage (on the GPU box): tools/ab_decode.sh <variant> ...  -- compress+decode timing on several inputs with variant libraries; base = the built one
cd $GRAFT_REPO_ROOT
cp lz4-java_amd/liblz4hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so lz4-java_amd/liblz4hip.so; else cp lz4-java_amd/variants/$v.so lz4-java_amd/liblz4hip.so; fi
  echo "== $v"
  for d in ${AB_DATA:-synth book1 geo pic}; do echo -n "$d: "; timeout 120 python tools/gpu_one.py ${AB_N:-32768} 2 0 $d 2>&1 | tail -1; done
done
cp /tmp/base.so lz4-java_amd/liblz4hip.so
