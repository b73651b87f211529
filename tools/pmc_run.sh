#!/bin/bash
# usage: tools/pmc_run.sh <outdir> <kernel-name-filter> '<command>' < counter-sets (one --pmc set per line)
# Each set is collected in its own run of <command> (rocprofv3 --kernel-trace --pmc only); per-dispatch averages go to <outdir>/summary.txt
out=$1; filt=$2; cmd=$3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out; : > $out/summary.txt
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o pmc -- $cmd > $out/p$i.log 2>&1
  db=$(find $out/p$i -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python tools/rocprof_summary.py $db $filt | grep -v "^$" | grep -v "^##\|^kernel " | sed 's/  */ /g' | cut -c1-40,60- >> $out/summary.txt; else echo "set failed: $set" >> $out/summary.txt; fi
  rm -rf $out/p$i
done
cat $out/summary.txt
