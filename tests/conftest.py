"""Shared fixtures.  `-m "not gpu"`: oracle vs golden vectors, host logic, algorithm cores in the
lock-step simulator, C-ABI export check.  `-m gpu`: parity of the HIP engine (through the C ABI)
against the oracle.  The oracle (oracle/) is used here ONLY as the checker."""
import hashlib
import importlib
import json
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def amd():
    """the product package (directory name has a hyphen -> importlib)"""
    return importlib.import_module("lz4-java_amd")


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="session")
def ref(O):
    """the reference liblz4 1.9.3 itself (oracle/_ref) -- the strongest checker available"""
    return O.ref()


@pytest.fixture(scope="session")
def port(O):
    return O.port()


@pytest.fixture(scope="session")
def golden():
    return json.load(open(os.path.join(GOLD, "golden.json")))


@pytest.fixture(scope="session")
def corpus(O):
    """name -> bytes for every golden input that can be rebuilt without /root/reference"""
    book1 = open(os.path.join(GOLD, "book1_200000.bin"), "rb").read()
    geo = open(os.path.join(GOLD, "geo_65536.bin"), "rb").read()
    pic = open(os.path.join(GOLD, "pic_65536.bin"), "rb").read()
    return {
        "book1[:65536]": book1[:65536], "geo[:65536]": geo, "pic[:65536]": pic,
        "book1[:65546]": book1[:65546], "book1[:65547]": book1[:65547], "book1[:200000]": book1,
        "zeros65536": bytes(65536), "len12": book1[:12], "len13": book1[:13], "len0": b"",
        "selftest": b"abcd      abcdefghij", "readme": b"12345345234572",
        "gen_block(65536,0)": O.gen_block(65536, 0), "gen_block(65536,1)": O.gen_block(65536, 1),
        "gen_block(1<<20,3,win=4096)": O.gen_block(1 << 20, 3, win=4096),
    }


def sha(b):
    return hashlib.sha256(b).hexdigest()


def rnd_inputs(O, corpus, seed, count, max_n=70000):
    """deterministic mixed bag of inputs: random, text, image, low-entropy, periodic, synthetic"""
    rng = random.Random(seed)
    book1, pic, geo = corpus["book1[:200000]"], corpus["pic[:65536]"], corpus["geo[:65536]"]
    sizes = [0, 1, 5, 12, 13, 14, 15, 20, 40, 63, 64, 65, 70, 100, 127, 128, 129, 200, 500, 1000, 3000, 9000, 30000,
             65535, 65536, 65546, 65547, 65548, 70000]
    out = []
    for _ in range(count):
        t = rng.randrange(7)
        n = rng.choice(sizes) if rng.random() < 0.7 else rng.randrange(0, 4000)
        n = min(n, max_n)
        if t == 0:
            v = rng.randbytes(min(n, 20000))
        elif t == 1:
            o = rng.randrange(len(book1) - n); v = book1[o:o + n]
        elif t == 2:
            n = min(n, len(pic)); o = rng.randrange(len(pic) - n + 1); v = pic[o:o + n]
        elif t == 3:
            v = bytes(rng.randrange(2) for _ in range(min(n, 30000)))
        elif t == 4:
            p = rng.randbytes(rng.randrange(1, 70)); v = (p * (n // len(p) + 1))[:n]
        elif t == 5:
            n = min(n, len(geo)); o = rng.randrange(len(geo) - n + 1); v = geo[o:o + n]
        else:
            v = O.gen_block(n, rng.randrange(1000), litmax=rng.choice([2, 4, 38, 200]), win=rng.choice([4, 8, 64, 300, 4096, 65535]))
        out.append(v)
    return out


def deep_decoder_cases(ref, O, corpus, rng, lz4_seq):
    """streams long enough for the decoder's deep interior loop (it needs 2 KB of stream ahead): compressed real and synthetic
    blocks, hand-assembled streams mixing one-step sequences with long literal runs, long matches, offsets shorter than a step and
    offsets that reach into the sequences still waiting in a slot; then the same streams corrupted, truncated, with wrong
    capacities.  -> (valid [(stream, size)], all cases [(stream, capacity)]); used by the CPU simulator test and the GPU test"""
    valid = []
    for v in (corpus["book1[:200000]"][:70000], corpus["geo[:65536]"], corpus["pic[:65536]"], O.gen_block(65536, 1), O.gen_block(300000, 2, win=4096),
              O.gen_block(65536, 3, litmax=4, win=64), O.gen_block(100000, 4, litmax=70, win=300), bytes(50000) + rng.randbytes(3000) + bytes(40000),
              (b"abcdefghijklmnopqrstuvwxyz" * 3 + rng.randbytes(11)) * 900):
        valid.append((ref.compress_fast(v), len(v)))
    for trial in range(12):   # hand-assembled: every kind of sequence next to every other
        c, n = bytearray(), 0
        for _ in range(rng.randrange(300, 900)):
            kind = rng.random()
            if kind < 0.70: lit, ml = rng.randrange(0, 65), rng.randrange(4, 65)
            elif kind < 0.80: lit, ml = rng.randrange(65, 600), rng.randrange(4, 65)
            elif kind < 0.90: lit, ml = rng.randrange(0, 40), rng.randrange(65, 1200)
            else: lit, ml = rng.randrange(0, 20), rng.randrange(4, 30)
            hi = n + lit
            if hi == 0: lit, hi = 1, 1
            off = rng.choice([rng.randrange(1, min(hi, 64) + 1), rng.randrange(1, min(hi, 300) + 1), rng.randrange(1, min(hi, 65535) + 1)])
            c += lz4_seq(lit, ml, off, rng); n += lit + ml
        last = rng.randrange(5, 40)
        c += bytes([last << 4 if last < 15 else 0xF0]) + (bytes([last - 15]) if last >= 15 else b"") + rng.randbytes(last); n += last
        valid.append((bytes(c), n))
    cases = [(c, n) for c, n in valid]
    for c, n in valid:   # malformed: flipped bytes, truncation, wrong capacities
        for _ in range(3):
            b = bytearray(c)
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(len(b))] = rng.randrange(256)
            cases.append((bytes(b), n))
        cases.append((c[:rng.randrange(len(c) // 2, len(c))], n))
        cases.append((c, n - rng.randrange(1, 700)))
        cases.append((c, n + rng.randrange(1, 100)))
    return valid, cases


def lz4_seq(lit, ml, off, rng):
    """one LZ4 sequence: `lit` random literals, then a match of `ml` >= 4 bytes at distance `off` (hand-assembled)"""
    def ext(v):
        out = bytearray()
        while v >= 255:
            out.append(255); v -= 255
        out.append(v)
        return bytes(out)
    tok = (min(lit, 15) << 4) | min(ml - 4, 15)
    s = bytes([tok]) + (ext(lit - 15) if lit >= 15 else b"") + rng.randbytes(lit) + bytes([off & 255, off >> 8])
    return s + (ext(ml - 4 - 15) if ml - 4 >= 15 else b"")


def wild_piece_stream(kw, rng):
    """A hand-assembled valid block that aims at one hazard of the parallel wave loop: a sequence only the one-sequence step takes
    (a match that overlaps its own output) writes wave-wide pieces -- up to a step (256 bytes) beyond its end -- into the output ring,
    i.e. over the ring bytes that hold the output KW bytes earlier; a short trip right behind it whose matches lie KW - 16 .. KW - 300
    bytes back must not take those bytes from the ring.  (Found by the fuzz test in round 5: the trip's rule for "the ring still holds
    the source" looked at the trip's own output only.)"""
    seq = lz4_seq
    c, n = bytearray(), 0
    def add(lit, ml, off):
        nonlocal n
        c.extend(seq(lit, ml, off, rng)); n += lit + ml
    while n < kw + 600:                                   # history: short simple sequences, sources near by
        lit = rng.randrange(0, 20); add(lit, rng.randrange(4, 20), rng.randrange(1, min(n + lit, 2000) + 1) if n + lit else 1) if n + lit else add(5, 4, 3)
    for d in list(range(16, 320, 7)) * 2:
        add(rng.randrange(3, 9), rng.randrange(6, 12), rng.randrange(1, 5))      # overlaps its own output: the one-sequence step
        # a trip of little output (the rule compares a source's position with the END of everything the trip has found: 3 bytes of
        # stream for 4 of output keep that near): sources just inside the ring's reach
        for _ in range(rng.choice([3, 40])):
            lit = rng.choice([0, 0, 0, 1])
            add(lit, rng.randrange(4, 6), kw - d - rng.randrange(0, 4))
        for _ in range(rng.randrange(0, 3)):
            lit = rng.randrange(0, 20); add(lit, rng.randrange(4, 20), rng.randrange(1, 2000))
    for _ in range(400):                                  # far from the end of the stream: the loop stays in charge throughout
        lit = rng.randrange(0, 20); add(lit, rng.randrange(4, 20), rng.randrange(1, 2000))
    c.extend(bytes([0xF0, 1]) + rng.randbytes(16))        # last literals (the last match starts more than 12 bytes in front of the end)
    return bytes(c), n + 16


def ring_edge_stream(rng, n_target=90000):
    """A hand-assembled valid block of short sequences whose match distances cluster around the sizes of the wave loops' output rings
    (4 .. 64 KB, +- 320 bytes) -- "the ring still holds it" against "it comes from flushed memory" is decided there --, mixed with
    matches that overlap their own output (the one-sequence step and its wave-wide pieces), long literal runs and long matches (the
    slow copies, trips that end early) and plain sequences.  Returns (stream, decoded size)."""
    c, n = bytearray(), 0
    def add(lit, ml, off):
        nonlocal n
        off = max(1, min(off, n + lit, 65535))
        c.extend(lz4_seq(lit, ml, off, rng)); n += lit + ml
    add(12, 4, 5)
    while n < n_target:
        k = rng.random()
        if k < 0.55:
            if rng.random() < 0.5:
                off = rng.choice([4096, 8192, 16384, 32768, 65536]) + rng.randrange(-320, 321)
            else:
                off = rng.randrange(1, 2000)
            for _ in range(rng.choice([1, 1, 2, 5, 30])):
                add(rng.randrange(0, 7), rng.randrange(4, 9), off + rng.randrange(-3, 4))
        elif k < 0.65:
            add(rng.randrange(0, 10), rng.randrange(4, 41), rng.randrange(1, 9))
        elif k < 0.70:
            add(rng.randrange(15, 300), rng.randrange(4, 20), rng.randrange(1, 65536))
        elif k < 0.75:
            add(rng.randrange(0, 10), rng.randrange(20, 600), rng.randrange(1, 65536))
        else:
            add(rng.randrange(0, 31), rng.randrange(4, 31), rng.randrange(1, 65536))
    c.extend(bytes([0xF0, 1]) + rng.randbytes(16))
    return bytes(c), n + 16
