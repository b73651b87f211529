#!/usr/bin/env python3
"""Latency of the single-block entry points (what LZ4HIPCompressor.compress(byte[]...) costs per call through the JNI shim)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
amd = importlib.import_module("lz4-java_amd")
from oracle import oracle as O
f = amd.LZ4Factory.hipInstance()
blk = O.gen_block(65536, 0)
c = f.fastCompressor(); d = f.safeDecompressor(); h32 = amd.XXHashFactory.hipInstance().hash32()
out = bytearray(amd.maxCompressedLength(len(blk))); back = bytearray(len(blk))
n = c.compress(blk, 0, len(blk), out, 0, len(out))
comp = bytes(out[:n])
def bench(name, fn, reps=300, nbytes=len(blk)):
    fn(); t = time.perf_counter()
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t) / reps
    print("%-28s %8.1f us/call  (%.2f GB/s, one buffer at a time)" % (name, dt * 1e6, nbytes / dt / 1e9))
bench("compress_fast 64 KiB", lambda: c.compress(blk, 0, len(blk), out, 0, len(out)))
bench("decompress_safe 64 KiB", lambda: d.decompress(comp, 0, len(comp), back, 0, len(back)))
bench("xxh32 64 KiB", lambda: h32.hash(blk, 0, len(blk), 0))
big = bytes(bytearray(os.urandom(1 << 20)) * 64)
bench("xxh32 64 MiB (one buffer)", lambda: h32.hash(big, 0, len(big), 0), reps=3, nbytes=len(big))
