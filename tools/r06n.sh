#!/bin/bash
# round 6, GPU session n: A/B in one session -- the product (walk by pointer doubling from 24 starts on) against a build without it
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06n}; mkdir -p $out
W="cfg2_2048,cfg2_4096,appf2048,appf4096,book2048,book8192,book65536"
for rep in 1 2; do
  echo "== product $rep" >> $out/ab.log
  timeout 600 python tools/ring_matrix.py $W d >> $out/ab.log 2>&1
  echo "== nowp $rep" >> $out/ab.log
  LZ4HIP_LIBRARY=$PWD/lz4-java_amd/variants/nowp.so timeout 600 python tools/ring_matrix.py $W d >> $out/ab.log 2>&1
done
grep -v amdgpu $out/ab.log | grep "==\|  d" | paste - - - - - - - - | cut -c1-400
