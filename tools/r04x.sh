#!/bin/bash
# text core: A/B of the current build against variant libraries + parity slices
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r04x
export GRAFT_REPO_ROOT=$PWD
{
for d in book1 geo synth; do echo "-- $d"; AB_DATA=$d AB_N=${AB_N:-65536} bash tools/ab_compress.sh ${VARIANTS:-prewalk}; done
timeout 900 python tools/gpu_fuzz.py ${FUZZ_N:-2500} ${FUZZ_SEED:-91} | tail -6
timeout 600 python tools/gpu_fuzz_u32.py 200 5 600000 | grep -v amdgpu.ids | head -3
} > gpurun_out/r04x/ab.log 2>&1
cat gpurun_out/r04x/ab.log
