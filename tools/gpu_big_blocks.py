#!/usr/bin/env python3
"""Fast compress of big (byU32) blocks: gpu_big_blocks.py <n_blocks> <block_bytes> [win]  -- App. F blocks, time + round trip +
bytes of 4 blocks against the reference library"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
from oracle import oracle as O
n, blk = int(sys.argv[1]), int(sys.argv[2]); win = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
dev = torch.device("cuda:0"); cap = amd.maxCompressedLength(blk)
src = torch.empty(n * blk, dtype=torch.uint8, device=dev); amd.DeviceBatch.gen_blocks(src, blk, blk, n, win=win)
so = torch.arange(n, dtype=torch.int64, device=dev) * blk; sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
comp = torch.empty(n * cap, dtype=torch.uint8, device=dev); co = torch.arange(n, dtype=torch.int64, device=dev) * cap
cc = torch.full((n,), cap, dtype=torch.int32, device=dev); clen = torch.zeros(n, dtype=torch.int32, device=dev)
back = torch.zeros(n * blk, dtype=torch.uint8, device=dev); dlen = torch.zeros(n, dtype=torch.int32, device=dev)
for _ in range(2):
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a.record(); amd.DeviceBatch.compress_fast(src, so, sl, comp, co, cc, clen); b.record()
    amd.DeviceBatch.decompress_safe(comp, co, clen, back, so, sl, dlen); c.record(); torch.cuda.synchronize()
ok = bool(torch.equal(back, src))
ref = O.ref() if O.ref_path() else None
if ref:
    cl = clen.cpu().tolist()
    for i in (0, 1, n // 2, n - 1):
        want = ref.compress_fast(src[i * blk:(i + 1) * blk].cpu().numpy().tobytes())
        ok = ok and cl[i] == len(want) and comp[i * cap:i * cap + cl[i]].cpu().numpy().tobytes() == want
print("%d x %d B (win %d): compress %.2f ms (%.1f GB/s)  decode %.2f ms (%.1f GB/s)  ratio %.3f  ok=%s" % (
    n, blk, win, a.elapsed_time(b), n * blk / a.elapsed_time(b) / 1e6, b.elapsed_time(c), n * blk / b.elapsed_time(c) / 1e6, n * blk / float(clen.sum()), ok))
