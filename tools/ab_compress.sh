#!/bin/bash
# usage (on the GPU box): tools/ab_compress.sh <variant> ...  -- fast compress (+ round trip check) with variant libraries; base = the built one
cd $GRAFT_REPO_ROOT
cp lz4-java_amd/liblz4hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so lz4-java_amd/liblz4hip.so; else cp lz4-java_amd/variants/$v.so lz4-java_amd/liblz4hip.so; fi
  echo -n "== $v: "
  timeout 120 python tools/gpu_one.py ${AB_N:-65536} 2 0 ${AB_DATA:-synth} 2>&1 | tail -1
done
cp /tmp/base.so lz4-java_amd/liblz4hip.so
