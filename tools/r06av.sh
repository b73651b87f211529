#!/bin/bash
# round 6, GPU session av: trio against the one-wavefront loop at 2 .. 5 blocks per CU, now that the latter copies overlapping matches inside its passes
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
out=gpurun_out/${1:-r06av}; mkdir -p $out
for rep in 1 2; do
echo "== trio (64:8) $rep" >> $out/matrix.log
timeout 600 python tools/ring_matrix.py appf512,appf768,appf1024,appf1280,book1024,book1280,geo1280,pic1280,cfg2_768,cfg2_1024,cfg2_1280 64:8:0:0 >> $out/matrix.log 2>&1
echo "== wave (64:5) $rep" >> $out/matrix.log
timeout 600 python tools/ring_matrix.py appf512,appf768,appf1024,appf1280,book1024,book1280,geo1280,pic1280,cfg2_768,cfg2_1024,cfg2_1280 64:5:0:0 >> $out/matrix.log 2>&1
done
python tools/matrix_table.py $out/matrix.log
