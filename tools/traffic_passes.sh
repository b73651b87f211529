#!/bin/bash
# HBM traffic per launch (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE need separate --pmc passes; FETCH_SIZE
# under-reports wide coalesced reads by 2x on gfx950).  usage: tools/traffic_passes.sh <outdir> <steps>
out=$1; steps=${2:-2}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 420 rocprofv3 --kernel-trace --pmc $c -d $out/$tag -o pmc -- python bench.py --steps $steps --warmup 1 --no-cpu-baseline > $out/$tag.log 2>&1
  echo "pass $tag rc=$?"
done
