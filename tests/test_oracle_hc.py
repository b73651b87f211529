"""Pins the HC restatement (oracle/lz4hc_oracle.c, LZ4_compress_HC levels 1..12: hash-chain strategy and optimal parser)
against the golden table and against the reference library itself (levels, level clamps, limited output)."""
import os
import random

import pytest

from conftest import rnd_inputs, sha


def test_hc_golden_table(port, golden, corpus):
    for name, data in corpus.items():
        g = golden["inputs"][name]
        c = port.compress_hc(data, 9)
        assert (len(c), sha(c)) == (g["hc9_size"], g["hc9_sha256"]), name
        assert port.decompress_safe(c, len(data)) == data
        for lvl in (10, 12):   # optimal parser (lz4-java levels 10 and 12..17)
            c = port.compress_hc(data, lvl)
            assert (len(c), sha(c)) == (g["hc%d_size" % lvl], g["hc%d_sha256" % lvl]), (name, lvl)
            assert port.decompress_safe(c, len(data)) == data


def test_hc_full_calgary(port, golden):
    cal = "/root/reference/src/test-resources/calgary/"
    if not os.path.isdir(cal):
        pytest.skip("reference tree not on this box")
    for name in ("book1", "geo", "pic"):
        data = open(cal + name, "rb").read()
        c = port.compress_hc(data, 9)
        assert (len(c), sha(c)) == (golden["inputs"][name]["hc9_size"], golden["inputs"][name]["hc9_sha256"])


def test_hc_levels_and_clamps_vs_reference(port, ref, corpus):
    data = corpus["book1[:200000]"][:40000]
    pic = corpus["pic[:65536]"][:30000]
    for lvl in range(1, 10):
        assert port.compress_hc(data, lvl) == ref.compress_hc(data, lvl), lvl
        assert port.compress_hc(pic, lvl) == ref.compress_hc(pic, lvl), lvl
    for lvl in (0, -5):                                   # liblz4: level < 1 -> 9
        assert port.compress_hc(data, lvl) == ref.compress_hc(data, 9)
    assert port.compress_hc(data, 1) == port.compress_hc(data, 2)
    for lvl in (10, 11, 12):
        assert port.compress_hc(data, lvl) == ref.compress_hc(data, lvl), lvl
        assert port.compress_hc(pic, lvl) == ref.compress_hc(pic, lvl), lvl
    assert port.compress_hc(data, 17) == ref.compress_hc(data, 12)   # liblz4: level > 12 -> 12 (lz4-java 13..17)


def test_hc_optimal_parser_fuzz_vs_reference(port, ref, O, corpus):
    """levels 10..12 (LZ4HC_compress_optimal: chain swap + pattern analysis in every search, price table over 4096 positions)"""
    rng = random.Random(29)
    for v in rnd_inputs(O, corpus, 52, 220):
        lvl = rng.choice([10, 11, 12])
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_hc_raw(v, lvl, full)
        for cap in (full, max(0, er + rng.choice([-1, 0, 1, -9, 9])), rng.randrange(0, full + 1)):
            a = ref.compress_hc_raw(v, lvl, cap)
            b = port.compress_hc_raw(v, lvl, cap)
            assert a[0] == b[0] and (a[0] <= 0 or a[1] == b[1]), (len(v), lvl, cap, a[0], b[0])
    for period in (1, 2, 3, 4, 5, 8):
        p = rng.randbytes(period)
        for n in (5000, 70000):
            v = bytearray((p * (n // period + 1))[:n])
            for _ in range(n // 3000):
                v[rng.randrange(n)] ^= 0x55
            v = bytes(v)
            for lvl in (10, 12):
                assert port.compress_hc(v, lvl) == ref.compress_hc(v, lvl), (period, n, lvl)


def test_hc_fuzz_vs_reference(port, ref, O, corpus):
    rng = random.Random(19)
    for v in rnd_inputs(O, corpus, 51, 500):
        lvl = rng.choice([1, 3, 4, 6, 9, 9, 9, 0])
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_hc_raw(v, lvl, full)
        for cap in (full, max(0, er + rng.choice([-1, 0, 1, -9, 9])), rng.randrange(0, full + 1)):
            a = ref.compress_hc_raw(v, lvl, cap)
            b = port.compress_hc_raw(v, lvl, cap)
            assert a[0] == b[0] and (a[0] <= 0 or a[1] == b[1]), (len(v), lvl, cap, a[0], b[0])
    # pattern-analysis heavy inputs (level 9 only): long runs, short periods, runs broken by noise
    for period in (1, 2, 3, 4, 5, 8):
        p = rng.randbytes(period)
        for n in (5000, 70000, 200000):
            v = bytearray((p * (n // period + 1))[:n])
            for _ in range(n // 3000):
                v[rng.randrange(n)] ^= 0x55
            v = bytes(v)
            assert port.compress_hc(v, 9) == ref.compress_hc(v, 9), (period, n)
