#!/bin/bash
# HBM traffic per launch (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE need separate --pmc passes; FETCH_SIZE
# under-reports wide coalesced reads by 2x on gfx950).  usage: tools/traffic_passes.sh <outdir> <steps>
out=$1; steps=${2:-2}
# third argument "text": the passes run tools/gpu_text_legs.py (the real-text legs alone) instead of bench.py
cmd="python bench.py --steps $steps --warmup 1 --no-cpu-baseline --no-live-traffic"
[ "$3" = "text" ] && cmd="python tools/gpu_text_legs.py"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 420 rocprofv3 --kernel-trace --pmc $c -d $out/$tag -o pmc -- $cmd > $out/$tag.log 2>&1
  echo "pass $tag rc=$?"
done
