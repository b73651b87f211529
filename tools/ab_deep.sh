#!/bin/bash
# usage (GPU box): tools/ab_deep.sh <variant ...>   -- decoder defaults of the built library and of variant libraries: 4 MiB blocks, text, App. F
cd $GRAFT_REPO_ROOT
cp lz4-java_amd/liblz4hip.so /tmp/base.so
for v in base "$@"; do
  if [ $v != base ]; then cp lz4-java_amd/variants/$v.so lz4-java_amd/liblz4hip.so; fi
  for n in 4096 16384; do echo -n "$v cfg3 $n: "; timeout 200 python tools/gpu_cfg3.py $n 2 0 2>&1 | tail -1; done
  for data in book1 synth; do for n in 4096 16384 32768; do echo -n "$v $data $n: "; timeout 200 python tools/gpu_one.py $n 2 0 $data 2>&1 | tail -1; done; done
done
cp /tmp/base.so lz4-java_amd/liblz4hip.so
