#!/usr/bin/env python3
"""Host-pointer batch API on headline blocks in pageable host memory: GB/s per call (PCIe both ways) and, with
LZ4HIP_HOST_PROF=1 in the environment, the calling thread's time split per call (stderr).
usage: host_path_probe.py [n_blocks=16384] [calls=4]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
amd = importlib.import_module("lz4-java_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
blk = 65536; cap = amd.maxCompressedLength(blk)
dev = torch.device("cuda:0")
src = torch.empty(n * blk, dtype=torch.uint8, device=dev); amd.DeviceBatch.gen_blocks(src, blk, blk, n)
h_src = src.cpu().numpy().copy()
del src
so = np.arange(n, dtype=np.uint64) * blk; sl = np.full(n, blk, dtype=np.int32)
co = np.arange(n, dtype=np.uint64) * cap; cc = np.full(n, cap, dtype=np.int32)
h_dst = np.zeros(n * cap, dtype=np.uint8); h_back = np.zeros(n * blk, dtype=np.uint8)
clen = np.zeros(n, dtype=np.int32); dlen = np.zeros(n, dtype=np.int32)
import ctypes as C
lib = amd.lib()
def p(a):
    t = {np.uint64: C.POINTER(C.c_uint64), np.int32: C.POINTER(C.c_int32)}.get(a.dtype.type)
    return a.ctypes.data_as(t) if t else a.ctypes.data_as(C.c_void_p)
for c in range(calls):
    t0 = time.perf_counter()
    rc = lib.lz4hip_compress_fast_batch(p(h_src), p(so), p(sl), p(h_dst), p(co), p(cc), p(clen), n)
    t1 = time.perf_counter()
    assert rc == 0, rc
    rc = lib.lz4hip_decompress_safe_batch(p(h_dst), p(co), p(clen), p(h_back), p(so), p(sl), p(dlen), n)
    t2 = time.perf_counter()
    assert rc == 0, rc
    print("call %d: compress %.1f ms (%.1f GB/s)  decompress %.1f ms (%.1f GB/s)  ok=%s" % (
        c, 1e3 * (t1 - t0), n * blk / (t1 - t0) / 1e9, 1e3 * (t2 - t1), n * blk / (t2 - t1) / 1e9, bool((h_back == h_src).all() and (dlen == blk).all())), flush=True)
