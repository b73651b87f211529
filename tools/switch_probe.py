#!/usr/bin/env python3
"""Where is the crossover between the lean core and the window-parallel core?  Blocks of words drawn from dictionaries of different
sizes / lengths (8192 copies of one 64 KiB block each): bytes per LZ4 sequence against the time of compress_core 3 / 1 / 5.
usage: switch_probe.py [n_blocks=8192]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
amd = importlib.import_module("lz4-java_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0"); blk = 65536; cap = amd.maxCompressedLength(blk)
rng = np.random.default_rng(7)
def words_block(vocab, wmin, wmax, noise):
    ws = [rng.integers(97, 123, size=rng.integers(wmin, wmax + 1), dtype=np.uint8).tobytes() for _ in range(vocab)]
    p = 1.0 / np.arange(1, vocab + 1); p /= p.sum()
    out = bytearray()
    while len(out) < blk:
        out += ws[rng.choice(vocab, p=p)]
        if rng.random() < noise: out += rng.integers(0, 256, size=rng.integers(1, 6), dtype=np.uint8).tobytes()
        else: out += b" "
    return bytes(out[:blk])
cases = [("book1", open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read()[:blk])]
for vocab, wmin, wmax, noise in ((200, 3, 8, 0.0), (2000, 4, 10, 0.0), (2000, 6, 14, 0.1), (500, 8, 20, 0.2), (300, 12, 30, 0.3), (200, 20, 40, 0.4), (100, 30, 60, 0.5)):
    cases.append(("words v%d %d-%d noise %.1f" % (vocab, wmin, wmax, noise), words_block(vocab, wmin, wmax, noise)))
so = torch.arange(n, dtype=torch.int64, device=dev) * blk; sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
comp = torch.empty(n * cap, dtype=torch.uint8, device=dev); co = torch.arange(n, dtype=torch.int64, device=dev) * cap
cc = torch.full((n,), cap, dtype=torch.int32, device=dev); clen = torch.zeros(n, dtype=torch.int32, device=dev)
def seqs(c):   # count the sequences of one compressed block
    i, k = 0, 0
    while i < len(c):
        t = c[i]; i += 1; l = t >> 4
        if l == 15:
            while True:
                b = c[i]; i += 1; l += b
                if b != 255: break
        i += l
        if i >= len(c): break
        i += 2; m = t & 15
        if m == 15:
            while True:
                b = c[i]; i += 1
                if b != 255: break
        k += 1
    return k + 1
SW = (6, 7, 8, 9, 10, 12, 16, 20)
print("%-30s %9s %8s | %9s %9s | core 5 with compress_switch = %s  (ms per %d blocks)" % ("data", "bytes/seq", "ratio", "core 3", "core 1", SW, n))
book = np.frombuffer(open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read(), dtype=np.uint8)
cases.insert(1, ("book1 slices (bench)", None))
for name, b in cases:
    if b is None:
        bdev = torch.from_numpy(book.copy()).to(dev)
        offs = torch.arange(n, dtype=torch.int64, device=dev) * 7919 % (len(book) - blk)
        src = bdev[(offs[:, None] + torch.arange(blk, dtype=torch.int64, device=dev)[None, :]).reshape(-1)]
    else:
        src = torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(dev).repeat(n)
    t = {}
    def run():
        for _ in range(2):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); amd.DeviceBatch.compress_fast(src, so, sl, comp, co, cc, clen); e.record(); torch.cuda.synchronize()
        return a.elapsed_time(e)
    for core in (3, 1):
        amd.set_option("compress_core", core)
        t[core] = run()
    amd.set_option("compress_core", 5)
    ts = []
    for sw in SW:
        amd.set_option("compress_switch", sw)
        ts.append(run())
    amd.set_option("compress_switch", 16)
    c0 = comp[:int(clen[0])].cpu().numpy().tobytes()
    print("%-30s %9.1f %8.2f | %9.2f %9.2f | %s" % (name, blk / seqs(c0), blk / len(c0), t[3], t[1], " ".join("%6.2f" % x for x in ts)), flush=True)
