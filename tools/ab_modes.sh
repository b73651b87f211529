#!/bin/bash
# usage (on the GPU box): tools/ab_modes.sh   -- decoder variants (lanes / pipelined / staged) by batch size and data
cd $GRAFT_REPO_ROOT
for n in ${AB_SIZES:-16384 32768}; do for d in ${AB_DATA:-synth book1}; do
  for mode in "0 -1 -1" "4 0 0" "4 0 1" "8 0 0" "8 0 1" "8 1 0" "16 1 0"; do set -- $mode
    echo -n "$n $d lanes=$1 pipe=$2 stage=$3: "; DP=$2 DS=$3 timeout 120 python tools/gpu_one.py $n 2 $1 $d 2>&1 | tail -1 | sed 's/compress.*GB.s)  //'
  done; done; done
