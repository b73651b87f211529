#!/bin/bash
# round 5, GPU session a: the wave loop (a wavefront per block) -- the decoder tests with its variants in them, then the small-batch
# matrix: library defaults against the wave loop by batch size, 4 MiB blocks (the 8-GPU shard of BASELINE configs[2] is 2048 per GPU),
# 64 KiB App. F blocks and 64 KiB text blocks
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/r05a; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x --durations=6 -k "decode or deep or cfg3 or routed or slots or golden or malformed" > $out/pytest_decode.log 2>&1; echo "pytest rc=$?" >> $out/pytest_decode.log
timeout 600 python tools/ring_matrix.py cfg2_256,cfg2_1024,cfg2_2048,cfg2_4096,cfg2_8192 d,64:4:0:0,64:4:0:8192,64:4:0:16384 > $out/matrix_4MiB.log 2>&1
timeout 400 python tools/ring_matrix.py appf1,appf64,appf512,appf2048,appf4096,appf8192,appf16384 d,64:4:0:0,64:4:0:8192 > $out/matrix_appf.log 2>&1
timeout 400 python tools/ring_matrix.py book1,book512,book2048,book4096,book8192,geo2048,pic2048 d,64:4:0:0,64:4:0:8192 > $out/matrix_text.log 2>&1
timeout 300 python tools/ring_matrix.py cfg2_16384 d,64:4:0:8192 > $out/matrix_cfg2_full.log 2>&1
tail -6 $out/pytest_decode.log; cat $out/matrix_4MiB.log $out/matrix_appf.log $out/matrix_text.log $out/matrix_cfg2_full.log | grep -v "^$" | cut -c1-200
