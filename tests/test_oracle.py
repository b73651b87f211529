"""Pins the oracle: (1) the C restatement (oracle/lz4_oracle.c) against every golden vector produced
by the reference's own library (tests/golden/golden.json), (2) the restatement
against the reference library itself on fuzzed / malformed inputs when that library is on this box."""
import hashlib
import os
import random

import pytest

from conftest import rnd_inputs, sha


def test_golden_table_port(port, golden, corpus):
    for name, data in corpus.items():
        g = golden["inputs"][name]
        assert len(data) == g["n"] and hashlib.md5(data).hexdigest() == g["md5"], name
        c = port.compress_fast(data)
        assert (len(c), sha(c)) == (g["fast_size"], g["fast_sha256"]), name
        assert "%08x" % port.xxh32(data, 0) == g["xxh32_seed0"], name
        assert "%016x" % port.xxh64(data, 0) == g["xxh64_seed0"], name
        assert "%08x" % port.xxh32(data, 0x9747b28c) == g["xxh32_seed9747b28c"], name
        assert "%016x" % port.xxh64(data, 0x9747b28c) == g["xxh64_seed9747b28c"], name
        assert port.decompress_safe(c, len(data)) == data
        r, d = port.decompress_fast_raw(c, len(data))
        assert r == len(c) and d == data


def test_golden_table_full_files(port, golden):
    cal = "/root/reference/src/test-resources/calgary/"
    if not os.path.isdir(cal):
        pytest.skip("reference tree not on this box")
    for name in ("book1", "geo", "pic"):
        data = open(cal + name, "rb").read()
        g = golden["inputs"][name]
        c = port.compress_fast(data)
        assert (len(c), sha(c)) == (g["fast_size"], g["fast_sha256"])
        assert "%08x" % port.xxh32(data) == g["xxh32_seed0"] and "%016x" % port.xxh64(data) == g["xxh64_seed0"]


def test_known_answers(port, golden):
    # SURVEY.md App. D / App. E literals
    assert "%08x" % port.xxh32(b"", 0) == "02cc5d05" and "%08x" % port.xxh32(b"a", 0) == "550d7456"
    assert "%016x" % port.xxh64(b"", 0) == "ef46db3751d8e999" and "%016x" % port.xxh64(b"a", 0) == "d24ec4f1a98c6e5b"
    assert "%08x" % port.xxh32(b"12345345234572", 0x9747b28c) == "1e34488c"
    assert port.compress_fast(b"abcd      abcdefghij").hex() == "5161626364200100a06162636465666768696a"  # LZ4Factory.java:205
    assert port.compress_fast(b"12345345234572").hex() == "e03132333435333435323334353732"
    for n, b in golden["compress_bound"].items():
        assert port.compress_bound(int(n)) == b


def test_malformed_vectors_port(port, golden):
    for v in golden["malformed"]:
        vec = bytes.fromhex(v["hex"])
        r, d = port.decompress_safe_raw(vec, v["safe_cap"])
        assert r == v["safe_ret"]
        if r >= 0:
            assert d[:r].hex() == v["safe_out_hex"]
        assert port.decompress_fast_raw(vec, v["fast_len"])[0] == v["fast_ret"]


def test_gen_block_fingerprint(O, golden):
    b = O.gen_block(65536, 0)
    assert sha(b) == golden["gen_block_65536_0_sha256"] == "c1d73891b1a07b6f3fc298f21b379a7eb181e6da8e5572e49fd0158108fd881a"
    assert b[:16].hex() == "865f8989f63c2c1e1b0573f51aaba202"


def test_port_equals_reference_compress(port, ref, O, corpus):
    rng = random.Random(7)
    for v in rnd_inputs(O, corpus, 11, 400):
        full = ref.compress_bound(len(v))
        er, eb = ref.compress_fast_raw(v, full)
        caps = [full, max(0, er + rng.choice([-1, 0, 1, -7, 9])), rng.randrange(0, full + 1)]
        for cap in caps:
            a = ref.compress_fast_raw(v, cap)
            b = port.compress_fast_raw(v, cap)
            assert a[0] == b[0] and (a[0] <= 0 or a[1] == b[1]), (len(v), cap)


def test_port_equals_reference_decode_fuzz(port, ref, O, corpus):
    """return codes AND output of LZ4_decompress_safe on valid, truncated, extended, bit-flipped and
    random streams, with exact and wrong capacities (LZ4Test.java:188-255 properties)"""
    rng = random.Random(5)
    accepted = 0
    for v in rnd_inputs(O, corpus, 13, 3000, max_n=9000):
        c = bytearray(ref.compress_fast(v))
        mode, cap = rng.randrange(6), len(v)
        if mode == 1:
            cap = max(0, len(v) + rng.choice([-1, 1, -5, 5, -12, 12, -33, 33, 64, 100]))
        elif mode == 2 and c:
            for _ in range(rng.randrange(1, 4)):
                c[rng.randrange(len(c))] = rng.randrange(256)
        elif mode == 3 and len(c) > 1:
            c = c[:rng.randrange(1, len(c))]
        elif mode == 4:
            c = c + rng.randbytes(rng.randrange(1, 20))
        elif mode == 5:
            c, cap = bytearray(rng.randbytes(rng.randrange(1, 40))), rng.randrange(0, 200)
        c = bytes(c)
        r1, d1 = port.decompress_safe_raw(c, cap)
        r2, d2 = ref.decompress_safe_raw(c, cap)
        assert r1 == r2, (mode, len(v), cap, c[:24].hex())
        if r2 >= 0:
            accepted += 1
            assert d1[:r2] == d2[:r2]
        if mode == 0:
            assert port.decompress_fast_raw(c, cap) == ref.decompress_fast_raw(c, cap)
    assert accepted > 500


def test_issue12_regression_blob(port, ref):
    """LZ4Test.testRoundtripIssue12 (LZ4Test.java:487-541): the input that broke an early lz4-java; bytes [9:] as there.
    The restatement must agree with the reference library on it (fast, HC 9, HC 12) and every stream must decode back."""
    data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "issue12.bin"), "rb").read()[9:]
    assert len(data) == 1510
    c = port.compress_fast(data)
    assert c == ref.compress_fast(data) and port.decompress_safe(c, len(data)) == data
    for level in (1, 9, 12):
        h = port.compress_hc(data, level)
        assert h == ref.compress_hc(data, level) and ref.decompress_safe(h, len(data)) == data


def test_fast_decoder_contract_vectors(O):
    """tests/golden/fast_decode_contract.json (the bounded fast decoder's pinned behaviour, include/lz4hip.h): the C restatement
    reproduces every vector"""
    import hashlib, json, os
    from conftest import GOLD
    cases = json.load(open(os.path.join(GOLD, "fast_decode_contract.json")))["cases"]
    assert len(cases) >= 600
    for e in cases:
        r, d = O.decompress_fast_bounded(bytes.fromhex(e["hex"]), e["src_cap"], e["dst_len"])
        assert r == e["ret"]
        if r >= 0:
            assert hashlib.sha256(d[:e["dst_len"]]).hexdigest() == e["sha256"]


def test_cpu_bench_harness_one_block_per_thread():
    """oracle/cpu_bench (the CPU leg of bench.py): with as many threads as blocks every share is ONE block, repeated passes and
    stealing make several workers ask for the same block, and the harness must still hand a block to one worker at a time
    (two decoders on one output buffer read each other's wild-copy scratch).  Port library only: runs without oracle/_ref."""
    import json
    import subprocess
    odir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    exe = os.path.join(odir, "cpu_bench_test")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-o", exe, os.path.join(odir, "cpu_bench.c"), "-ldl", "-lpthread"])
    try:
        port = os.path.join(odir, "liblz4oracle.so")
        for n, blk in ((4, 65536), (3, 1 << 20)):
            out = subprocess.check_output([exe, "port", port, port, str(n), str(blk), str(n), "2", "0", "38", "4096"],
                                          env=dict(os.environ, CPU_BENCH_MIN_MS="60"), timeout=120)
            r = json.loads(out)
            assert r["ok"] is True and r["n_blocks"] == n and min(r["passes"]) >= 2
    finally:
        os.remove(exe)
