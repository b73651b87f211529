#!/bin/bash
# compact-entry kernel: chains per CU (variant libraries pk<pairs>x<workgroups>)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r04y
{
for v in ${VARIANTS:-base pk4x2 pk3x3 pk2x4 pk2x5}; do
  echo "== $v"
  if [ $v = base ]; then L=""; else L=$PWD/lz4-java_amd/variants/$v.so; fi
  LZ4HIP_LIBRARY=$L U32_BLOCKS=${U32_BLOCKS:-8192} python tools/gpu_fuzz_u32.py 3 1 2>&1 | grep "pack"
done
} > gpurun_out/r04y/chains.log 2>&1
cat gpurun_out/r04y/chains.log
