#!/usr/bin/env python3
"""A/B of the fast-compress cores on the GPU: bytes of core `CC` (default 3 = lean core) against core 0 on App. F blocks, text and edge
blocks, then timings.  usage: gpu_v2_check.py [n_blocks]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cc = int(os.environ.get("CC", "3"))
dev = torch.device("cuda:0"); blk = 65536; cap = amd.maxCompressedLength(blk)

def run(src, n, core, reps=1, same_src=False):
    so = torch.arange(n, dtype=torch.int64, device=dev) * (0 if same_src else blk)
    sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
    comp = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
    co = torch.arange(n, dtype=torch.int64, device=dev) * cap
    ccap = torch.full((n,), cap, dtype=torch.int32, device=dev)
    clen = torch.zeros(n, dtype=torch.int32, device=dev)
    amd.set_option("compress_core", core)
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); amd.DeviceBatch.compress_fast(src, so, sl, comp, co, ccap, clen); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return comp, clen, best

def same(c0, l0, c1, l1, n):
    if not torch.equal(l0, l1):
        bad = (l0 != l1).nonzero().flatten()
        return "LEN MISMATCH in %d blocks, first %d: %d vs %d" % (len(bad), bad[0], l0[bad[0]], l1[bad[0]])
    idx = torch.arange(cap, device=dev)[None, :] < l0[:, None].to(torch.int64)
    a = c0.view(n, cap)[idx]; b = c1.view(n, cap)[idx]
    return "bit-exact" if torch.equal(a, b) else "BYTES DIFFER"

import numpy as np
for name in ("synth", "synth1", "book1", "geo", "pic", "zeros", "random"):
    m = n if name in ("synth", "synth1") else min(n, 2560)
    if name == "synth":
        src = torch.empty(m * blk, dtype=torch.uint8, device=dev)
        amd.DeviceBatch.gen_blocks(src, blk, blk, m)
    elif name == "synth1":   # one App. F block repeated: every candidate fetch is an L2 hit (what prefetching could reach)
        one = torch.empty(blk, dtype=torch.uint8, device=dev)
        amd.DeviceBatch.gen_blocks(one, blk, blk, 1)
        src = one.repeat(2)
    else:
        if name == "book1": b = open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read()[:blk]
        elif name in ("geo", "pic"): b = open(os.path.join(ROOT, "tests/golden/%s_65536.bin" % name), "rb").read()
        elif name == "zeros": b = bytes(blk)
        else: b = os.urandom(blk)
        src = torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(dev).repeat(m)
    c0, l0, t0 = run(src, m, 0, 2, name == "synth1")
    c1, l1, t1 = run(src, m, cc, 3, name == "synth1")
    print("%-7s %6d blocks: core0 %.3f ms (%.1f GB/s)  core%d %.3f ms (%.1f GB/s)  %s" % (name, m, t0, m * blk / t0 / 1e6, cc, t1, m * blk / t1 / 1e6, same(c0, l0, c1, l1, m)), flush=True)
