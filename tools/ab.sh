#!/bin/bash
# usage (on the GPU box): tools/ab.sh <variant.so> ...   -- times every variant library on the same box (base = the built one)
cd $GRAFT_REPO_ROOT
cp lz4-java_amd/liblz4hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so lz4-java_amd/liblz4hip.so; else cp lz4-java_amd/variants/$v.so lz4-java_amd/liblz4hip.so; fi
  echo "== $v"
  python tools/gpu_one.py ${AB_N:-32768} 3 2>&1 | tail -2 | cut -c1-40
  [ -n "$AB_TEXT" ] && python tools/gpu_one.py 16384 2 0 book1 2>&1 | tail -1 | cut -c1-40
done
cp /tmp/base.so lz4-java_amd/liblz4hip.so
