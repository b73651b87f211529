#!/usr/bin/env python3
"""The two REAL-TEXT legs of the bench line by themselves (bench.py `real_book1`: 65536 x 64 KiB slices of Calgary book1, fast compress +
safe decompress; `real_book1_4MiB`: 2560 x 4 MiB blocks of the same text, fast compress), built exactly as bench.py builds them, three
launches each -- so that PMC passes of THIS command hold only those launches and profiles/traffic.json can carry the HBM bytes of
the text legs (round-4 verdict: both carried "traffic": null).  usage: gpu_text_legs.py   (tools/traffic_passes.sh <dir> 2 text)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
amd = importlib.import_module("lz4-java_amd")
dev = torch.device("cuda:0")
u8, i64, i32 = torch.uint8, torch.int64, torch.int32
book = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "book1_200000.bin"), "rb").read(), dtype=np.uint8)
bdev = torch.from_numpy(book.copy()).to(dev)


def batch(n, blk, cap):
    return dict(so=torch.arange(n, dtype=i64, device=dev) * blk, sl=torch.full((n,), blk, dtype=i32, device=dev),
                co=torch.arange(n, dtype=i64, device=dev) * cap, cc=torch.full((n,), cap, dtype=i32, device=dev),
                clen=torch.zeros(n, dtype=i32, device=dev), dlen=torch.zeros(n, dtype=i32, device=dev))


# ---- real_book1: every block = 64 KiB of book1 from a different offset (bench.py) ----
n, blk = 65536, 65536
cap = amd.maxCompressedLength(blk)
src = torch.empty(n * blk, dtype=u8, device=dev)
offs = torch.arange(n, dtype=i64, device=dev) * 7919 % (len(book) - blk)
ar = torch.arange(blk, dtype=i64, device=dev)
for c0 in range(0, n, 1024):
    c1 = min(n, c0 + 1024)
    src[c0 * blk:c1 * blk] = bdev[(offs[c0:c1, None] + ar[None, :]).reshape(-1)]
del ar
comp = torch.empty(n * cap, dtype=u8, device=dev)
back = torch.zeros(n * blk, dtype=u8, device=dev)
B = batch(n, blk, cap)
for _ in range(3):
    amd.DeviceBatch.compress_fast(src, B["so"], B["sl"], comp, B["co"], B["cc"], B["clen"])
for _ in range(3):
    amd.DeviceBatch.decompress_safe(comp, B["co"], B["clen"], back, B["so"], B["sl"], B["dlen"])
torch.cuda.synchronize()
print("real_book1: ratio %.3f  round trip %s" % (n * blk / int(B["clen"].sum().item()), bool(torch.equal(back, src))), flush=True)
del src, comp, back, B
torch.cuda.empty_cache()
# ---- real_book1_4MiB: blocks made of 70000-byte slices of book1 (bench.py) ----
bT, nT = 4 << 20, 2560
capT = amd.maxCompressedLength(bT)
sT = torch.empty(nT * bT, dtype=u8, device=dev)
slc = 70000
per = (bT + slc - 1) // slc
arT = torch.arange(slc, dtype=i64, device=dev)
blkT = torch.arange(nT, dtype=i64, device=dev) * 7919
sTv = sT.view(nT, bT)
for j in range(per):
    wdt = min(slc, bT - j * slc)
    oT = (blkT + j * 70000) % (len(book) - slc)
    sTv[:, j * slc:j * slc + wdt] = bdev[oT[:, None] + arT[None, :wdt]]
cT = torch.empty(nT * capT, dtype=u8, device=dev)
BT = batch(nT, bT, capT)
for _ in range(3):
    amd.DeviceBatch.compress_fast(sT, BT["so"], BT["sl"], cT, BT["co"], BT["cc"], BT["clen"])
torch.cuda.synchronize()
print("real_book1_4MiB: ratio %.3f" % (nT * bT / int(BT["clen"].sum().item())), flush=True)
