#!/usr/bin/env python3
"""At how many 64 KiB blocks per call does the HIP family beat the reference's nativeInstance() end to end?
Host-pointer batch API (pageable host memory in, host memory out: staging + PCIe + kernels) against the reference's liblz4 on
the same box: one thread calling block after block (what one Java thread does through JNI) and all host threads (oracle/cpu_bench).
usage: crossover_probe.py [block_bytes]"""
import importlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
amd = importlib.import_module("lz4-java_amd")
from oracle import oracle as O
ref = O.ref()
blk = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cap = amd.maxCompressedLength(blk)
cores = os.cpu_count()
blocks = [O.gen_block(blk, i) for i in range(256)]
print("# %d-byte App. F blocks; GPU = lz4hip_compress_fast_batch / lz4hip_decompress_safe_batch from pageable host memory (best of 5 calls);" % blk)
print("# CPU 1 thread = the reference's liblz4 called block after block (oracle/cpu_bench, 1 thread, per-block time x blocks); CPU %d threads = oracle/cpu_bench on the same blocks (best of 5)" % cores)
print("%8s | %12s %12s | %12s %12s | %12s %12s" % ("blocks", "GPU comp us", "GPU dec us", "CPU1 comp us", "CPU1 dec us", "CPUall comp", "CPUall dec"))
exe = os.path.join(ROOT, "oracle", "cpu_bench")
def cpu(n, threads):
    return json.loads(subprocess.check_output([exe, "reference", O.ref_path(), os.path.join(ROOT, "oracle", "liblz4oracle.so"), str(n), str(blk),
                                               str(threads), "5", "0", "38", "65535"]).decode().strip().splitlines()[-1])
r1 = cpu(256, 1)
c1, d1 = blk / r1["compress_GBps"] / 1e9, blk / r1["decompress_safe_GBps"] / 1e9     # seconds per block, one thread
for n in (1, 4, 16, 64, 256, 1024, 4096, 16384):
    src = bytearray(b"".join(blocks[i % 256] for i in range(n)))
    dst, back = bytearray(n * cap), bytearray(n * blk)
    so, sl = np.arange(n, dtype=np.uint64) * blk, np.full(n, blk, dtype=np.int32)
    do, dc = np.arange(n, dtype=np.uint64) * cap, np.full(n, cap, dtype=np.int32)
    tc, td = [], []
    for _ in range(6):
        t0 = time.perf_counter(); sizes = amd.LZ4HIPBatch.compress(src, so, sl, dst, do, dc); t1 = time.perf_counter()
        amd.LZ4HIPBatch.decompressSafe(dst, do, sizes, back, so, sl); t2 = time.perf_counter()
        tc.append(t1 - t0); td.append(t2 - t1)
    assert bytes(back) == bytes(src)
    r = cpu(n, min(cores, n))
    ca, da = n * blk / r["compress_GBps"] / 1e3, n * blk / r["decompress_safe_GBps"] / 1e3
    print("%8d | %12.0f %12.0f | %12.0f %12.0f | %12.0f %12.0f" % (n, min(tc[1:]) * 1e6, min(td[1:]) * 1e6, c1 * n * 1e6, d1 * n * 1e6, ca, da), flush=True)
