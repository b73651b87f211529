#!/usr/bin/env python3
"""Extended CPU fuzz of the lean fast-compress core in the lock-step lane simulator (random LDS-atomic lane orders) against the
reference library: inputs chosen to make lanes of one step share buckets / fingerprints (small alphabets, periodic data with
noise, short-window synthetic blocks).  usage: hostsim_fuzz_v2.py [cases=4000] [seed=1]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import test_hostsim as T
from oracle import oracle as O
import conftest
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sim = T.load_sim()
ref = O.ref()
bad = 0
for i in range(cases):
    t = rng.randrange(6)
    n = rng.choice([70, 200, 500, 1000, 3000, 9000, 20000, 65536]) if rng.random() < 0.8 else rng.randrange(13, 3000)
    if t == 0:
        a = rng.randrange(2, 9); v = bytes(rng.randrange(a) for _ in range(min(n, 20000)))
    elif t == 1:
        p = rng.randbytes(rng.randrange(1, 90)); v = bytearray((p * (n // len(p) + 1))[:n])
        for _ in range(rng.randrange(0, max(1, n // 50))):
            v[rng.randrange(len(v))] = rng.randrange(256)
        v = bytes(v)
    elif t == 2:
        v = O.gen_block(n, rng.randrange(100000), litmax=rng.choice([1, 2, 4, 8, 38]), win=rng.choice([4, 8, 16, 32, 64, 300, 65535]))
    elif t == 3:   # words from a tiny dictionary
        words = [rng.randbytes(rng.randrange(2, 9)) for _ in range(rng.randrange(2, 12))]
        b = bytearray()
        while len(b) < n:
            b += rng.choice(words)
        v = bytes(b[:n])
    elif t == 4:
        v = rng.randbytes(min(n, 20000))
    else:
        a = rng.randbytes(rng.randrange(4, 40)); b = bytearray()
        while len(b) < n:
            b += a if rng.random() < 0.7 else rng.randbytes(rng.randrange(1, 6))
        v = bytes(b[:n])
    full = ref.compress_bound(len(v))
    want = ref.compress_fast_raw(v, full)
    for _ in range(2):
        seed = rng.getrandbits(63) | 1
        r, b, st = T.sim_compress(sim, v, full, seed=seed, v2=True)
        r2, b2, _ = T.sim_compress(sim, v, full, seed=seed, raw=True)
        if r != want[0] or b != want[1] or r2 != want[0] or b2 != want[1]:
            bad += 1
            print("MISMATCH case %d type %d n %d seed %d" % (i, t, len(v), seed), flush=True)
print("hostsim lean-core fuzz: %d cases x 2 lane orders, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
