#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer batch API (what a JNI caller with direct ByteBuffers gets):
host_batch_probe.py <n_blocks> [block_bytes]   -- App. F blocks generated on the host side (oracle generator, setup only)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
amd = importlib.import_module("lz4-java_amd")
from oracle import oracle as O
n = int(sys.argv[1]); blk = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
cap = amd.maxCompressedLength(blk)
one = [O.gen_block(blk, i) for i in range(min(n, 256))]
src = bytearray(b"".join(one[i % len(one)] for i in range(n)))
dst = bytearray(n * cap); back = bytearray(n * blk)
so = np.arange(n, dtype=np.uint64) * blk; sl = np.full(n, blk, dtype=np.int32)
do = np.arange(n, dtype=np.uint64) * cap; dc = np.full(n, cap, dtype=np.int32)
for rep in range(3):
    t0 = time.perf_counter(); sizes = amd.LZ4HIPBatch.compress(src, so, sl, dst, do, dc); t1 = time.perf_counter()
    res = amd.LZ4HIPBatch.decompressSafe(dst, do, sizes, back, so, sl); t2 = time.perf_counter()
    print("host batch %d x %d B: compress %.1f ms (%.2f GB/s)  decompress %.1f ms (%.2f GB/s)  ok=%s" % (
        n, blk, (t1 - t0) * 1e3, n * blk / (t1 - t0) / 1e9, (t2 - t1) * 1e3, n * blk / (t2 - t1) / 1e9, bytes(back) == bytes(src) and bool((np.asarray(res) == blk).all())), flush=True)
