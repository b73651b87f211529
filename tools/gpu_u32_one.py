#!/usr/bin/env python3
"""ONE big-block compress workload a few times (for rocprofv3 --pmc passes): gpu_u32_one.py <n_blocks> <compress_pack 0|1> [block_bytes]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
nb = int(sys.argv[1]); pack = int(sys.argv[2]); b = int(sys.argv[3]) if len(sys.argv) > 3 else 4 << 20
dev = torch.device("cuda:0")
s = torch.empty(nb * b, dtype=torch.uint8, device=dev)
amd.DeviceBatch.gen_blocks(s, b, b, nb, first_idx=1 << 24, litmax=38, win=4096)
cap = amd.maxCompressedLength(b)
c = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
so = torch.arange(nb, dtype=torch.int64, device=dev) * b; sl = torch.full((nb,), b, dtype=torch.int32, device=dev)
do = torch.arange(nb, dtype=torch.int64, device=dev) * cap; dc = torch.full((nb,), cap, dtype=torch.int32, device=dev)
res = torch.empty(nb, dtype=torch.int32, device=dev)
amd.set_option("compress_core", 3); amd.set_option("compress_pack", pack)
for _ in range(3):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); amd.DeviceBatch.compress_fast(s, so, sl, c, do, dc, res); e.record(); torch.cuda.synchronize()
print("pack %d: %d x %d B: %.2f ms, %.1f GB/s" % (pack, nb, b, a.elapsed_time(e), nb * b / a.elapsed_time(e) / 1e6))
