#!/bin/bash
# developer probe: a build WITHOUT the far-source loads inside the pass / round loops (wrong bytes for far sources; BASELINE configs[2]'s blocks have 0.2 % of them):
# does the compiler's vmcnt(0) at the loop headers -- it waits for the flusher's stores and the refill load -- cost time?
cd "$(dirname "$0")/.."
out=gpurun_out/${1:-r05y2}; mkdir -p $out
for v in base nofar; do
  echo "== variant $v"
  if [ $v = base ]; then L=""; else L=$PWD/lz4-java_amd/variants/$v.so; fi
  LZ4HIP_LIBRARY=$L timeout 600 python tools/ring_matrix.py cfg2_256,cfg2_1024,cfg2_2048,cfg2_4096 d 2>&1 | tail -8
done > $out/variants.log 2>&1
cat $out/variants.log
