#!/usr/bin/env python3
"""HC timing probe: gpu_hc_probe.py <n_blocks> <block_bytes> [level]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
n = int(sys.argv[1]); blk = int(sys.argv[2]); level = int(sys.argv[3]) if len(sys.argv) > 3 else 9
dev = torch.device("cuda:0"); cap = amd.maxCompressedLength(blk)
src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
amd.DeviceBatch.gen_blocks(src, blk, blk, n, win=4096 if blk > 65536 else 65535)
so = torch.arange(n, dtype=torch.int64, device=dev) * blk; sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
comp = torch.empty(n * cap, dtype=torch.uint8, device=dev); co = torch.arange(n, dtype=torch.int64, device=dev) * cap
cc = torch.full((n,), cap, dtype=torch.int32, device=dev); clen = torch.zeros(n, dtype=torch.int32, device=dev)
back = torch.zeros(n * blk, dtype=torch.uint8, device=dev); dlen = torch.zeros(n, dtype=torch.int32, device=dev)
for _ in range(2):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); amd.DeviceBatch.compress_hc(src, so, sl, comp, co, cc, clen, level); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    amd.DeviceBatch.decompress_safe(comp, co, clen, back, so, sl, dlen); torch.cuda.synchronize()
    print("HC level %d: %d x %d B  %.2f ms  %.2f GB/s  ratio %.3f  roundtrip ok=%s" % (level, n, blk, ms, n * blk / ms / 1e6, n * blk / float(clen.sum()), bool(torch.equal(back, src))), flush=True)
