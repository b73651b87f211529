#!/bin/bash
# usage (on the GPU box): tools/ab_sizes.sh <variant> ...  -- decode at several batch sizes (App. F and text) with variant libraries
cd $GRAFT_REPO_ROOT
cp lz4-java_amd/liblz4hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so lz4-java_amd/liblz4hip.so; else cp lz4-java_amd/variants/$v.so lz4-java_amd/liblz4hip.so; fi
  echo "== $v"
  for n in ${AB_SIZES:-1024 4096 16384 32768 49152}; do for d in synth book1; do echo -n "$n $d: "; timeout 120 python tools/gpu_one.py $n 2 0 $d 2>&1 | tail -1 | sed 's/compress.*GB.s)  //'; done; done
done
cp /tmp/base.so lz4-java_amd/liblz4hip.so
