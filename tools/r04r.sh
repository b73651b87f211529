#!/bin/bash
# last GPU call of round 4: the committed evidence at the final sources, then the retry-path build (variants/retry.so) on the GPU for the first time
cd "$(dirname "$0")/.."; export GRAFT_REPO_ROOT=$PWD; mkdir -p gpurun_out/r04r
bash tools/collect_profiles.sh r04f > gpurun_out/r04r/collect.log 2>&1
{
export LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/retry.so
timeout 60 python tools/gpu_fuzz.py 1200 55 2>&1 | grep -v amdgpu | tail -3
U32_BLOCKS=8192 timeout 60 python tools/gpu_fuzz_u32.py 200 56 900000 2>&1 | grep -v amdgpu | tail -5
unset LZ4HIP_LIBRARY
AB_DATA=synth timeout 60 bash tools/ab_compress.sh retry
} > gpurun_out/r04r/retry.log 2>&1
tail -2 gpurun_out/r04r/collect.log | cut -c1-200; cat gpurun_out/r04r/retry.log
