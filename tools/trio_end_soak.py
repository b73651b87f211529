#!/usr/bin/env python3
"""Soak of the trio loop's end-of-block rules in the lane simulator (tests/hostsim: three host threads per block): trio_end_soak.py <seed> <cases> [trio|pair|par|wave] --
streams of 2.5 .. 30 KB of five kinds, intact / with capacities off by -607 .. +700 / damaged in their last 400 bytes / truncated there / extended / damaged anywhere, safe decoder
against the reference library, bounded fast decoder against the C restatement, every ring, both stream rings, five slot alignments."""
import sys, random, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_hostsim as T
from oracle import oracle as O
sim = T.load_sim(); ref = O.ref()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
book = open("tests/golden/book1_200000.bin", "rb").read()
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
loop = {"trio": "trio", "pair": "pair", "par": True, "wave": False}[sys.argv[3] if len(sys.argv) > 3 else "trio"]   # which loop of the wave family
t0 = time.time(); bad = 0; stats = {}
for it in range(n_cases):
    kind = rng.randrange(5)
    n = rng.randrange(2500, 30000)
    if kind == 0: v = O.gen_block(n, rng.randrange(1 << 20), litmax=rng.choice([2, 8, 38, 300]), win=rng.choice([40, 4096, 65535]))
    elif kind == 1: a = rng.randrange(len(book) - n); v = book[a:a + n]
    elif kind == 2: v = bytes(rng.randrange(4) for _ in range(n))
    elif kind == 3: v = (rng.randbytes(rng.randrange(1, 300)) * (n // 3 + 1))[:n]
    else: v = O.gen_block(n, rng.randrange(1 << 20))[: n // 2] + rng.randbytes(n - n // 2)
    c = bytearray(ref.compress_fast(v))
    mode = rng.randrange(7); cap = len(v)
    if mode == 1: cap = max(0, len(v) + rng.choice([-1, 1, -3, 5, -12, 12, -33, 33, -64, 64, -100, 100, -300, 300, -607, 700]))
    elif mode == 2:                       # damage near the end of the stream
        for _ in range(rng.randrange(1, 4)): c[len(c) - 1 - rng.randrange(min(len(c), 400))] = rng.randrange(256)
    elif mode == 3: c = c[:len(c) - rng.randrange(1, min(len(c) - 1, 400))]
    elif mode == 4: c = c + rng.randbytes(rng.randrange(1, 40))
    elif mode == 5:                       # damage anywhere
        for _ in range(rng.randrange(1, 3)): c[rng.randrange(len(c))] = rng.randrange(256)
    c = bytes(c)
    flag = T.wave_flag(rng.choice([13, 14, 16]), rng.random() < 0.5, loop)
    shift = rng.choice([0, 1, 5, 64, 131])
    r2, d2 = ref.decompress_safe_raw(c, cap)
    r1, d1 = T.sim_decode(sim, c, cap, 1, flag, shift=shift)
    ok = r1 == r2 and (r2 < 0 or d1[:r2] == d2[:r2])
    scap = max(len(c) + rng.choice([0, 0, 0, 3, 16, -1]), 0)
    r3, d3 = O.decompress_fast_bounded(c, scap, cap)
    r4, d4 = T.sim_decode(sim, c + bytes(max(0, scap - len(c))), cap, 0, flag, src_size=scap, shift=shift)
    ok = ok and r3 == r4 and (r3 < 0 or d3[:cap] == d4[:cap])
    stats[mode] = stats.get(mode, 0) + 1
    if not ok:
        bad += 1; print("MISMATCH", it, kind, mode, len(v), cap, r1, r2, r3, r4)
        if bad > 5: break
print("cases", n_cases, "by mode", stats, "bad", bad, "%.1fs" % (time.time() - t0))
