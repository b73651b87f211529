#!/usr/bin/env python3
"""Does it pay to run the decompress of one part of the batch under the compress of the next part (two streams)?
usage: overlap_probe.py [n_blocks=65536] [parts=2,4,8] ; env DS=0 runs the decoder without its LDS staging (compress takes all LDS of a CU)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
parts_list = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
dev = torch.device("cuda:0"); blk = 65536; cap = amd.maxCompressedLength(blk)
src = torch.empty(n * blk, dtype=torch.uint8, device=dev); amd.DeviceBatch.gen_blocks(src, blk, blk, n)
so = torch.arange(n, dtype=torch.int64, device=dev) * blk; sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
comp = torch.empty(n * cap, dtype=torch.uint8, device=dev); co = torch.arange(n, dtype=torch.int64, device=dev) * cap
cc = torch.full((n,), cap, dtype=torch.int32, device=dev); clen = torch.zeros(n, dtype=torch.int32, device=dev)
back = torch.zeros(n * blk, dtype=torch.uint8, device=dev); dlen = torch.zeros(n, dtype=torch.int32, device=dev)
if os.environ.get("DS"): amd.set_option("decode_stage", int(os.environ["DS"]))
if os.environ.get("DL"): amd.set_option("decode_lanes", int(os.environ["DL"]))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for parts in parts_list:
    m = n // parts
    for rep in range(3):
        back.zero_()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        evs = []
        for p in range(parts):
            sl_ = slice(p * m, (p + 1) * m)
            with torch.cuda.stream(s1):
                amd.DeviceBatch.compress_fast(src, so[sl_], sl[sl_], comp, co[sl_], cc[sl_], clen[sl_])
                e = torch.cuda.Event(); e.record(s1); evs.append(e)
            with torch.cuda.stream(s2):
                s2.wait_event(evs[p])
                amd.DeviceBatch.decompress_safe(comp, co[sl_], clen[sl_], back, so[sl_], sl[sl_], dlen[sl_])
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b)
    print("parts %d: %.2f ms per step -> %.1f GB/s round trip  ok=%s" % (parts, t, n * blk / t / 1e6, bool(torch.equal(back, src))), flush=True)
