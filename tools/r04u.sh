#!/bin/bash
# byU32 ISA loop: fuzz + timing (U32_BLOCKS x 4 MiB), optionally against the compiler-generated loop (variant build noasm32)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r04u
U32_BLOCKS=${U32_BLOCKS:-8192} timeout 1200 python tools/gpu_fuzz_u32.py ${U32_N:-1000} ${U32_SEED:-7} ${U32_MAX:-1500000} > gpurun_out/r04u/fuzz_asm.log 2>&1; echo "fuzz asm rc=$?" >> gpurun_out/r04u/fuzz_asm.log
if [ -n "$U32_AB" ] && [ -f lz4-java_amd/variants/noasm32.so ]; then
  U32_BLOCKS=${U32_BLOCKS:-8192} LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/noasm32.so timeout 600 python tools/gpu_fuzz_u32.py 4 1 > gpurun_out/r04u/fuzz_c.log 2>&1; echo "rc=$?" >> gpurun_out/r04u/fuzz_c.log
fi
tail -6 gpurun_out/r04u/fuzz_asm.log; [ -n "$U32_AB" ] && tail -4 gpurun_out/r04u/fuzz_c.log
