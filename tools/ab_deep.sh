#!/bin/bash
# usage (GPU box): tools/ab_deep.sh   -- the deep decoder loop (decode_pipe 2) against the defaults: 4 MiB blocks, text, App. F 64 KiB
cd $GRAFT_REPO_ROOT
N3=${N3:-16384}
echo -n "cfg3 $N3 default:      "; timeout 200 python tools/gpu_cfg3.py $N3 2 0 2>&1 | tail -1
for L in ${LANES:-4 8 16}; do echo -n "cfg3 $N3 deep lanes $L: "; DP=2 DS=0 timeout 200 python tools/gpu_cfg3.py $N3 2 $L 2>&1 | tail -1; done
for data in book1 synth; do
  for n in 16384 65536; do
    echo -n "$data $n default:        "; timeout 200 python tools/gpu_one.py $n 2 0 $data 2>&1 | tail -1
    for L in ${LANES:-4 8 16}; do echo -n "$data $n deep lanes $L:   "; DP=2 DS=0 timeout 200 python tools/gpu_one.py $n 2 $L $data 2>&1 | tail -1; done
  done
done
