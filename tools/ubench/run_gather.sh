#!/bin/bash
# usage (on the GPU box): tools/ubench/run_gather.sh <out.txt>   -- the random-line gather matrix + the request-size mix of three shapes
out=${1:-gpurun_out/gather.txt}
cd $GRAFT_REPO_ROOT
G=tools/ubench/gather
{
echo "# random-line gather throughput, 1x MI355X (tools/ubench/gather.hip; independent loads, 8 in flight per lane, 4 waves per workgroup)"
for mib in 32 256 8192; do
  for shape in "1 4" "1 16" "4 16" "8 16"; do
    for w in 4 8 16 32; do timeout 60 $G $mib $w $shape 0 32; done
  done
done
echo "# non-temporal loads (same shapes, 16 waves per CU)"
for mib in 256 8192; do for shape in "1 4" "4 16" "8 16"; do timeout 60 $G $mib 16 $shape 1 32; done; done
} > $out 2>&1
# request-size mix at the fabric (TCC_EA0_RDREQ: all / 32 B / 64 B / 128 B) for three shapes on the 8 GiB footprint
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for shape in "1 4 0" "4 16 0" "8 16 0" "4 16 1"; do
  tag=$(echo $shape | tr ' ' '_')
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/gp; timeout 120 rocprofv3 --kernel-trace --pmc $set -d /tmp/gp -o pmc -- $G 8192 16 $shape 32 > /tmp/gp.log 2>&1
    db=$(find /tmp/gp -name "*.db" | head -1)
    echo "## shape (lanes/line bytes/lane nt) = $shape ; counters: $set" >> $out
    [ -n "$db" ] && python tools/rocprof_summary.py $db gather | grep -v "^$" | tail -4 >> $out
  done
done
cat $out
