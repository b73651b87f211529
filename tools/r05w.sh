#!/bin/bash
# byU16 chains per CU (probe builds, NOT bit-exact: 4096-bucket tables of 16 KB so that 5 / 6 / 8 finder-writer pairs fit one workgroup;
# built from a copy of csrc with the hand-scheduled loop's bucket shift moved to 20): is the 64 KiB kernel linear in its chains like the packed one?
cd "$(dirname "$0")/.."; out=gpurun_out/${1:-r05w}; mkdir -p $out
{
for v in base u16h12c5 u16h12c6 u16h12c8; do
  if [ $v = base ]; then L=""; else L=$PWD/lz4-java_amd/variants/$v.so; fi
  [ -n "$L" ] && [ ! -f "$L" ] && continue
  for rep in 1 2; do
    LZ4HIP_LIBRARY=$L timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-configs --no-live-traffic 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'compress', l['compress_GBps'], 'decompress', l['decompress_GBps'], 'value', l['value'], 'ratio', l['config']['ratio'], 'verified', l['verified'])"
  done
done
} > $out/chains_u16.log 2>&1
cat $out/chains_u16.log
