#!/bin/bash
# round 5, GPU session d: the PARALLEL wave loop (decode_pipe 5) -- decoder tests with its variants, then the small-batch matrix
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r05d}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x --durations=6 -k "decode or deep or cfg3 or routed or slots or golden or malformed" > $out/pytest_decode.log 2>&1; echo "pytest rc=$?" >> $out/pytest_decode.log
timeout 600 python tools/ring_matrix.py cfg2_256,cfg2_1024,cfg2_2048,cfg2_4096,cfg2_8192 d,64:4:0:0,64:5:0:0,64:5:0:8192 > $out/matrix_4MiB.log 2>&1
timeout 400 python tools/ring_matrix.py appf1,appf512,appf2048,appf4096,appf8192,book1,book512,book2048,book4096,geo2048,pic2048 d,64:4:0:0,64:5:0:0,64:5:0:8192 > $out/matrix_64k.log 2>&1
tail -4 $out/pytest_decode.log; cat $out/matrix_4MiB.log $out/matrix_64k.log | grep -v "^$\|amdgpu.ids" | cut -c1-200
