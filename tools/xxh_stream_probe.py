#!/usr/bin/env python3
"""Times one long buffer through the one-shot and streaming xxhash paths (host pointer and device pointer)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
n = 256 << 20
t = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
host = t.cpu().numpy().tobytes()
f = amd.XXHashFactory.hipInstance()
for name, mk in (("xxh32", f.newStreamingHash32), ("xxh64", f.newStreamingHash64)):
    h = mk(0)
    h.update_device(t.data_ptr(), 1 << 20); h.getValue(); h.reset()
    t0 = time.time(); h.update_device(t.data_ptr(), n); v = h.getValue(); dt = time.time() - t0
    print("%s stream, device pointer: %.2f GB/s (%x)" % (name, n / dt / 1e9, v))
    h.reset()
    t0 = time.time()
    for o in range(0, n, 4 << 20):
        h.update(host, o, 4 << 20)
    v2 = h.getValue(); dt = time.time() - t0
    print("%s stream, host pointer, 4 MiB updates: %.2f GB/s ok=%s" % (name, n / dt / 1e9, v == v2))
    one = (f.hash32() if name == "xxh32" else f.hash64())
    t0 = time.time(); v3 = one.hash(host, 0, n, 0); dt = time.time() - t0
    print("%s one-shot, host pointer: %.2f GB/s ok=%s" % (name, n / dt / 1e9, v == v3))
