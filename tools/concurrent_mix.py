#!/usr/bin/env python3
"""Two decoders at once on one batch: the 4-lane staged loop (memory-bound: 4.3x the algorithmic bytes through HBM, issue slots idle) on the first part of the
headline's 65536 x 64 KiB blocks, the wave kernel (issue-bound, window on chip, traffic 1.0x) on the rest, on two streams.  concurrent_mix.py [kind] [n]
-> ms and GB/s for shares of 0 .. 50 % to the wave kernel, every result compared with the input."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "appf"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
blk = 65536
src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
amd.DeviceBatch.gen_blocks(src, blk, blk, n)
cap = amd.maxCompressedLength(blk)
so = torch.arange(n, dtype=torch.int64, device=dev) * blk
sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
comp = torch.empty(n * cap, dtype=torch.uint8, device=dev)
co = torch.arange(n, dtype=torch.int64, device=dev) * cap
cc = torch.full((n,), cap, dtype=torch.int32, device=dev)
clen = torch.zeros(n, dtype=torch.int32, device=dev)
dlen = torch.zeros(n, dtype=torch.int32, device=dev)
amd.DeviceBatch.compress_fast(src, so, sl, comp, co, cc, clen)
torch.cuda.synchronize()
back = torch.zeros(n * blk, dtype=torch.uint8, device=dev)
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def knobs(lanes, pipe, stage, ring):
    amd.set_option("decode_lanes", lanes); amd.set_option("decode_pipe", pipe); amd.set_option("decode_stage", stage); amd.set_option("decode_ring", ring)


def run(share, order):
    k = n - int(n * share) // 4096 * 4096          # whole rounds of the wave kernel's 4096 blocks
    back.zero_(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); eA = torch.cuda.Event(enable_timing=True); eB = torch.cuda.Event(enable_timing=True)
        e0.record(); sA.wait_event(e0); sB.wait_event(e0)
        def a():
            if k:
                with torch.cuda.stream(sA):
                    knobs(4, 0, 1, 0)
                    amd.DeviceBatch.decompress_safe(comp, co[:k], clen[:k], back, so[:k], sl[:k], dlen[:k])
                    eA.record()
        def b():
            if k < n:
                with torch.cuda.stream(sB):
                    knobs(64, 5, -1, 8192)
                    amd.DeviceBatch.decompress_safe(comp, co[k:], clen[k:], back, so[k:], sl[k:], dlen[k:])
                    eB.record()
        (a(), b()) if order == 0 else (b(), a())
        torch.cuda.synchronize()
        t = max(e0.elapsed_time(eA) if k else 0.0, e0.elapsed_time(eB) if k < n else 0.0)
        best = min(best, t)
    knobs(0, -1, -1, 0)
    return k, best, bool(torch.equal(back, src))


for share in (0.0, 0.0625, 0.125, 0.1875, 0.25, 0.3125, 0.375, 0.5, 1.0):
    for order in (0, 1):
        k, t, ok = run(share, order)
        print("%s n %d: staged loop %6d blocks | wave kernel %6d blocks (%s first)  %8.3f ms  %7.1f GB/s  ok=%s" % (kind, n, k, n - k, "staged" if order == 0 else "wave", t, n * blk / t / 1e6, ok), flush=True)
