#!/bin/bash
# byU32 blocks: fuzz + timing (U32_BLOCKS x 4 MiB; compact entries on ten pairs per CU vs 64-bit entries on five)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r04u
{
U32_TIMING=0 timeout 900 python tools/gpu_fuzz_u32.py ${U32_N2:-24} 11 5500000
U32_BLOCKS=${U32_BLOCKS:-8192} timeout 1200 python tools/gpu_fuzz_u32.py ${U32_N:-600} ${U32_SEED:-8} ${U32_MAX:-1500000}
echo "rc=$?"
} > gpurun_out/r04u/fuzz_asm.log 2>&1
grep -v amdgpu.ids gpurun_out/r04u/fuzz_asm.log | tail -14
