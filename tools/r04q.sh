#!/bin/bash
# PMC passes of big-block fast compress: ten chains with packed entries (compress_pack 1) against five with 64-bit entries (0)
cd "$(dirname "$0")/.."; export GRAFT_REPO_ROOT=$PWD; mkdir -p gpurun_out/r04q
for pack in 1 0; do
  bash tools/pmc_run.sh gpurun_out/r04q/pack$pack compress_fast_v2w "python tools/gpu_u32_one.py 2560 $pack" < tools/pmc_sets_chains.txt > gpurun_out/r04q/pack$pack.txt 2>&1
done
tail -40 gpurun_out/r04q/pack1.txt; echo ======; tail -40 gpurun_out/r04q/pack0.txt
