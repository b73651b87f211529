#!/usr/bin/env python3
"""CPU baseline for row a4: the reference's own liblz4 1.9.3 LZ4_compress_HC (oracle/_ref) on all host threads of this box,
on the same App. F blocks the GPU probe uses.  usage: cpu_hc_baseline.py <n_blocks> <block_bytes> [level]
(ctypes releases the GIL during the call, so a thread pool scales across cores)"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
n = int(sys.argv[1]); blk = int(sys.argv[2]); level = int(sys.argv[3]) if len(sys.argv) > 3 else 9
ref = O.ref()
threads = os.cpu_count() or 1
blocks = [O.gen_block(blk, i, win=4096 if blk > 65536 else 65535) for i in range(n)]
cap = ref.compress_bound(blk)
def work(b):
    r, _ = ref.compress_hc_raw(b, level, cap)
    return r
best = None
for _ in range(3):
    t = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        sizes = list(ex.map(work, blocks))
    dt = time.perf_counter() - t
    best = dt if best is None else min(best, dt)
print("CPU LZ4_compress_HC level %d (reference liblz4 1.9.3): %d x %d B, %d threads, best of 3: %.1f ms = %.2f GB/s, ratio %.3f"
      % (level, n, blk, threads, best * 1e3, n * blk / best / 1e9, n * blk / float(sum(sizes))))
