#!/bin/bash
# after a compress change: the compress-side gpu tests + fuzz slices + 4 MiB timing
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r04p
{
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streams.py tests/test_gpu_fuzz_slice.py tests/test_gpu_jni.py tests/test_gpu_multidev.py -x -q -m gpu 2>&1 | tail -5
U32_BLOCKS=${U32_BLOCKS:-8192} timeout 900 python tools/gpu_fuzz_u32.py 300 21 4300000 2>&1 | grep -v amdgpu.ids | tail -6
} > gpurun_out/r04p/log.txt 2>&1
cat gpurun_out/r04p/log.txt
