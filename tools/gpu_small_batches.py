#!/usr/bin/env python3
"""Many small fast-compress launches (1..48 blocks of random sizes) against the reference library: shakes the finder/writer
hand-over of the default core at its edges (fewer blocks than wavefronts, blocks of a few bytes, single-batch blocks).
usage: gpu_small_batches.py [launches=400] [seed=1]"""
import importlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
amd = importlib.import_module("lz4-java_amd")
from oracle import oracle as O
ref = O.ref()
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
book = open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read()
tot = 0
for it in range(launches):
    nb = rng.choice([1, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 16, 33, 48])
    blocks = []
    for _ in range(nb):
        n = rng.choice([0, 1, 12, 13, 14, 20, 64, 200, 1000, 5000, 20000, 65536, 65546, 65547, 70000]) if rng.random() < 0.6 else rng.randrange(0, 3000)
        t = rng.randrange(4)
        if t == 0: v = rng.randbytes(n)
        elif t == 1: o = rng.randrange(len(book) - n); v = book[o:o + n]
        elif t == 2: v = O.gen_block(n, rng.randrange(1 << 20), litmax=rng.choice([2, 38, 200]), win=rng.choice([8, 300, 65535]))
        else: v = bytes(rng.randrange(3) for _ in range(min(n, 8000)))
        blocks.append(v)
    caps = [ref.compress_bound(len(v)) if rng.random() < 0.7 else rng.randrange(0, ref.compress_bound(len(v)) + 1) for v in blocks]
    src = b"".join(blocks); so, sl, do, p, q = [], [], [], 0, 0
    for b, c in zip(blocks, caps):
        so.append(p); sl.append(len(b)); do.append(q); p += len(b); q += c
    dst = bytearray(max(q, 1))
    out = amd.LZ4HIPBatch.compress(src, so, sl, dst, do, caps)
    for i, (r, o) in enumerate(zip(out, do)):
        er, eb = ref.compress_fast_raw(blocks[i], caps[i])
        if r != er or (er > 0 and bytes(dst[o:o + r]) != eb[:er]):
            print("MISMATCH launch %d block %d len %d cap %d: %d vs %d" % (it, i, len(blocks[i]), caps[i], r, er)); sys.exit(1)
    tot += nb
print("small batches: %d launches, %d blocks, all bit-exact" % (launches, tot))
