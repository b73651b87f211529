#!/bin/bash
# usage (on the GPU box): tools/decode_matrix.sh  -- decoder lanes x pipelined/staged x batch size x data matrix (round 1)
# decode defaults study (on the GPU box): lanes x {plain, pipelined, staged} x batch size x data
for data in synth book1; do for n in 4096 16384 65536; do for cfg in "4 0 0" "8 0 0" "8 1 0" "16 1 0" "4 0 1" "8 0 1" "16 0 1"; do set -- $cfg
  r=$(DP=$2 DS=$3 python tools/gpu_one.py $n 2 $1 $data | tail -1 | sed 's/.*decode/decode/' | cut -c1-40); echo "$data n=$n GL=$1 PIPE=$2 STAGE=$3: $r"; done; done; done
for cfg in "16 1 0" "16 0 1" "8 0 1" "32 0 1"; do set -- $cfg; echo "cfg3 4096 GL=$1 PIPE=$2 STAGE=$3: $(DP=$2 DS=$3 python tools/gpu_cfg3.py 4096 2 $1 2>&1 | grep 'safe decompress' | sed 's/.*= //' | cut -c1-30)"; done
