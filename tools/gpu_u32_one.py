#!/usr/bin/env python3
"""ONE big-block compress workload a few times (for rocprofv3 --pmc passes): gpu_u32_one.py <n_blocks> <compress_pack 0|1> [block_bytes] [synth|book] [core]
book: every block is made of 70000-byte slices of Calgary book1 (English text; a slice is longer than the match window, so a block is text throughout)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
nb = int(sys.argv[1]); pack = int(sys.argv[2]); b = int(sys.argv[3]) if len(sys.argv) > 3 else 4 << 20
data = sys.argv[4] if len(sys.argv) > 4 else "synth"; core = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda:0")
s = torch.empty(nb * b, dtype=torch.uint8, device=dev)
if data == "synth":
    amd.DeviceBatch.gen_blocks(s, b, b, nb, first_idx=1 << 24, litmax=int(os.environ.get("U32_LITMAX", "38")), win=int(os.environ.get("U32_WIN", "4096")))
else:
    import numpy as np
    book = np.frombuffer(open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read(), dtype=np.uint8)
    bdev = torch.from_numpy(book.copy()).to(dev)
    sl_ = 70000; per = (b + sl_ - 1) // sl_
    ar = torch.arange(sl_, dtype=torch.int64, device=dev)
    for i in range(nb):
        offs = (torch.arange(per, dtype=torch.int64, device=dev) * 70000 + i * 7919) % (len(book) - sl_)   # (the same text comes back no closer than 130000 bytes)
        s[i * b:(i + 1) * b] = bdev[(offs[:, None] + ar[None, :]).reshape(-1)][:b]
cap = amd.maxCompressedLength(b)
c = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
so = torch.arange(nb, dtype=torch.int64, device=dev) * b; sl = torch.full((nb,), b, dtype=torch.int32, device=dev)
do = torch.arange(nb, dtype=torch.int64, device=dev) * cap; dc = torch.full((nb,), cap, dtype=torch.int32, device=dev)
res = torch.empty(nb, dtype=torch.int32, device=dev)
amd.set_option("compress_core", core); amd.set_option("compress_pack", pack)
for _ in range(3):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); amd.DeviceBatch.compress_fast(s, so, sl, c, do, dc, res); e.record(); torch.cuda.synchronize()
print("%s core %d pack %d: %d x %d B: %.2f ms, %.1f GB/s, ratio %.3f" % (data, core, pack, nb, b, a.elapsed_time(e), nb * b / a.elapsed_time(e) / 1e6, nb * b / float(res.sum().item())))
