#!/bin/bash
# usage (GPU box): tools/deep_matrix.sh <variant ...>  -- decoder loops by batch size: pipe 1 (two-trip), pipe 2 (deep) of the built library and of variants
cd $GRAFT_REPO_ROOT
cp lz4-java_amd/liblz4hip.so /tmp/base.so
for v in base "$@"; do
  if [ $v != base ]; then cp lz4-java_amd/variants/$v.so lz4-java_amd/liblz4hip.so; fi
  for data in synth book1; do
    for n in ${NS:-8192 16384 24576 32768 49152 65536}; do
      for dp in ${DPS:-1 2}; do echo -n "$v $data $n pipe $dp: "; DP=$dp DS=0 timeout 100 python tools/gpu_one.py $n 2 ${LANES:-8} $data 2>&1 | tail -1 | sed 's/compress.*decode/decode/'; done
    done
  done
  echo -n "$v cfg3 16384 deep: "; DP=2 DS=0 timeout 200 python tools/gpu_cfg3.py 16384 2 8 2>&1 | tail -1
done
cp /tmp/base.so lz4-java_amd/liblz4hip.so
