#!/usr/bin/env python3
"""The density route's crossover: for batches of more than 16 blocks per CU, the lane-group default of the batch size against the wave
kernel (decode_route_short 0 / 1 = never / always) on workloads of different sequence density, with the density decode_route_kernel
measures (sampled hops per 256 bytes of stream).  route_sweep.py [sizes] [workloads]
workloads: book, pic, geo, appf (litmax 38), lit16, lit8, lit4, lit2 (App. F generator with shorter literal runs, win 65535), cfg2_N (N x 4 MiB)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
amd = importlib.import_module("lz4-java_amd")
dev = torch.device("cuda:0")
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "6144,16384,65536").split(",")]
wls = (sys.argv[2] if len(sys.argv) > 2 else "book,lit2,lit4,lit8,lit16,appf,pic,geo").split(",")
book = np.frombuffer(open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read(), dtype=np.uint8)


def make(kind, n, blk):
    src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
    if kind == "book":
        bdev = torch.from_numpy(book.copy()).to(dev)
        offs = torch.arange(n, dtype=torch.int64, device=dev) * 7919 % (len(book) - blk)
        ar = torch.arange(blk, dtype=torch.int64, device=dev)
        for c0 in range(0, n, 1024):
            c1 = min(n, c0 + 1024)
            src[c0 * blk:c1 * blk] = bdev[(offs[c0:c1, None] + ar[None, :]).reshape(-1)]
    elif kind in ("pic", "geo"):
        b = np.frombuffer(open(os.path.join(ROOT, "tests/golden/%s_65536.bin" % kind), "rb").read(), dtype=np.uint8)
        src = torch.from_numpy(b.copy()).to(dev).repeat(n)
    else:
        near = kind.endswith("w4k")             # appfw4k, lit8w4k: the same generator with a 4 KB match window
        base = kind[:-3] if near else kind
        lm = 38 if base in ("appf", "cfg2") else int(base[3:])
        amd.DeviceBatch.gen_blocks(src, blk, blk, n, first_idx=3 << 24, litmax=lm, win=4096 if (kind == "cfg2" or near) else 65535)
    return src


for kind in wls:
    for n in sizes:
        blk = 65536
        k = kind
        if kind.startswith("cfg2_"):
            k, n, blk = "cfg2", int(kind[5:]), 4 << 20
        src = make(k, n, blk)
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
        def run(lanes, pipe, stage, ring):
            amd.set_option("decode_lanes", lanes); amd.set_option("decode_pipe", pipe); amd.set_option("decode_stage", stage); amd.set_option("decode_ring", ring)
            back.zero_()
            best = 1e30
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); amd.DeviceBatch.decompress_safe(comp, co, clen, back, so, sl, dlen); b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
            return best, bool(torch.equal(back, src))
        staged = run(4, 0, 1, 0) if n >= 40960 else None
        deep = run(8, 2, 0, 0)
        wave = run(64, 5, 0, 8192)
        ringl = run(4, 3, 0, 2048) if kind.startswith("cfg2_") else None   # (big blocks: the ring loop too)
        dflt = run(0, -1, -1, 0)
        r = amd.last_decode_route()
        gb = lambda t: n * blk / t / 1e6
        print("%-9s n %6d ratio %5.2f  sampled %5.1f seq/256B %6.1f B/seq near %3.0f %%  %s deep %8.3f ms %6.1f GB/s | wave %8.3f ms %6.1f GB/s | default -> route %d %8.3f ms %6.1f GB/s  ok=%s" % (
            kind, n, n * blk / int(clen.sum().item()), 256.0 * r[1] / max(r[2], 1), r[6] / max(r[5], 1), 100.0 * r[4] / max(r[5], 1),
            ("staged %8.3f ms %6.1f GB/s |" % (staged[0], gb(staged[0]))) if staged else "", deep[0], gb(deep[0]), wave[0], gb(wave[0]), r[0], dflt[0], gb(dflt[0]),
            deep[1] and wave[1] and dflt[1] and (staged is None or staged[1])) + ((" | ring %8.3f ms %6.1f GB/s" % (ringl[0], gb(ringl[0]))) if ringl else ""), flush=True)
        del src, comp, back
        torch.cuda.empty_cache()
        if kind.startswith("cfg2_"):
            break
