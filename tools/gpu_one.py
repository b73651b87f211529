#!/usr/bin/env python3
"""Runs ONE workload shape a few times (for rocprofv3 --pmc passes): gpu_one.py <n_blocks> <reps> [lanes] [data]
data: synth (App. F, default) | book1 | zeros | random"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
n = int(sys.argv[1]); reps = int(sys.argv[2]); lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
data = sys.argv[4] if len(sys.argv) > 4 else "synth"
dev = torch.device("cuda:0"); blk = 65536; cap = amd.maxCompressedLength(blk)
if data == "synth":
    src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
    amd.DeviceBatch.gen_blocks(src, blk, blk, n)
else:
    import numpy as np
    if data == "book1":
        b = open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read()[:blk]
    elif data in ("geo", "pic"):
        b = open(os.path.join(ROOT, "tests/golden/%s_65536.bin" % data), "rb").read()
    elif data == "zeros":
        b = bytes(blk)
    else:
        b = os.urandom(blk)
    src = torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(dev).repeat(n)
so = torch.arange(n, dtype=torch.int64, device=dev) * blk
sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
comp = torch.empty(n * cap, dtype=torch.uint8, device=dev)
co = torch.arange(n, dtype=torch.int64, device=dev) * cap
cc = torch.full((n,), cap, dtype=torch.int32, device=dev)
clen = torch.zeros(n, dtype=torch.int32, device=dev)
back = torch.zeros(n * blk, dtype=torch.uint8, device=dev)
dlen = torch.zeros(n, dtype=torch.int32, device=dev)
if lanes:
    amd.set_option("decode_lanes", lanes)
if os.environ.get("CC"):
    amd.set_option("compress_core", int(os.environ["CC"]))
if os.environ.get("CS"):
    amd.set_option("compress_switch", int(os.environ["CS"]))
if os.environ.get("DS"):
    amd.set_option("decode_stage", int(os.environ["DS"]))
if os.environ.get("DP"):
    amd.set_option("decode_pipe", int(os.environ["DP"]))
for _ in range(reps):
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a.record(); amd.DeviceBatch.compress_fast(src, so, sl, comp, co, cc, clen); b.record()
    amd.DeviceBatch.decompress_safe(comp, co, clen, back, so, sl, dlen); c.record()
    torch.cuda.synchronize()
    print("compress %.3f ms (%.1f GB/s)  decode %.3f ms (%.1f GB/s) ok=%s" % (a.elapsed_time(b), n * blk / a.elapsed_time(b) / 1e6,
          b.elapsed_time(c), n * blk / b.elapsed_time(c) / 1e6, bool(torch.equal(back, src))), flush=True)
