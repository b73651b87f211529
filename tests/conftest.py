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
