#!/usr/bin/env python3
"""Decoder loops by workload in ONE process: ring_matrix.py [workloads] [variants]
workloads (comma list): appf65536, appf16384, appf4096, book65536, book16384, geo32768, pic32768, cfg2_16384, cfg2_4096
variants (comma list of lanes:pipe:stage:ring, 'd' = the library's defaults): e.g. d,4:3:0:512,8:3:0:4096
Every timing is the best of three launches (HIP events); every result is compared with the input."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
amd = importlib.import_module("lz4-java_amd")
dev = torch.device("cuda:0")
wl = (sys.argv[1] if len(sys.argv) > 1 else "appf65536,book65536,cfg2_16384").split(",")
var = (sys.argv[2] if len(sys.argv) > 2 else "d,4:3:0:512").split(",")
book = np.frombuffer(open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read(), dtype=np.uint8)


def make(name):
    if name.startswith("cfg2_"):
        n, blk = int(name[5:]), 4 << 20
        src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
        amd.DeviceBatch.gen_blocks(src, blk, blk, n, first_idx=1 << 24, win=4096)
    else:
        kind = name.rstrip("0123456789"); n = int(name[len(kind):]); blk = 65536
        src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
        if kind == "appf":
            amd.DeviceBatch.gen_blocks(src, blk, blk, n)
        elif kind == "book":
            bdev = torch.from_numpy(book.copy()).to(dev)
            offs = torch.arange(n, dtype=torch.int64, device=dev) * 7919 % (len(book) - blk)
            ar = torch.arange(blk, dtype=torch.int64, device=dev)
            for c0 in range(0, n, 1024):
                c1 = min(n, c0 + 1024)
                src[c0 * blk:c1 * blk] = bdev[(offs[c0:c1, None] + ar[None, :]).reshape(-1)]
        else:
            b = np.frombuffer(open(os.path.join(ROOT, "tests/golden/%s_65536.bin" % kind), "rb").read(), dtype=np.uint8)
            src = torch.from_numpy(b.copy()).to(dev).repeat(n)
    return n, blk, src


for name in wl:
    n, blk, src = make(name)
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
    print("%s: %d x %d B, ratio %.3f" % (name, n, blk, n * blk / int(clen.sum().item())), flush=True)
    for v in var:
        if v == "d":
            lanes, pipe, stage, ring = 0, -1, -1, 0
        else:
            lanes, pipe, stage, ring = (int(x) for x in v.split(":"))
        amd.set_option("decode_lanes", lanes); amd.set_option("decode_pipe", pipe); amd.set_option("decode_stage", stage); amd.set_option("decode_ring", ring)
        back.zero_()
        best = 1e30
        try:
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); amd.DeviceBatch.decompress_safe(comp, co, clen, back, so, sl, dlen); b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
            ok = bool(torch.equal(back, src)) and bool(torch.equal(dlen, sl))
            print("  %-14s %9.3f ms  %8.1f GB/s  ok=%s" % (v, best, n * blk / best / 1e6, ok), flush=True)
        except Exception as e:
            print("  %-14s failed: %r" % (v, e), flush=True)
    del src, comp, back
    torch.cuda.empty_cache()
amd.set_option("decode_lanes", 0); amd.set_option("decode_pipe", -1); amd.set_option("decode_stage", -1); amd.set_option("decode_ring", 0)
