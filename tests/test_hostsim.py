"""The ALGORITHMS of the gfx950 kernels, executed on the CPU: tests/hostsim compiles the product's
cores (lz4-java_amd/csrc/lz4_fast_core.h, lz4_decode_core.h) against lock-step lane simulators and
this file checks them bit-for-bit against the oracle.  (The kernels proper are checked on the GPU by
test_gpu_*.py; this catches algorithmic regressions before any GPU time is spent.)"""
import ctypes as C
import os
import random
import subprocess

import pytest

from conftest import ROOT, rnd_inputs

_u8p = C.POINTER(C.c_uint8)


def load_sim():
    d = os.path.join(ROOT, "tests", "hostsim")
    # (LZ4HIP_SIM_FLAGS: a developer / soak build of the simulator with other macros -- e.g. the wave loop's instances switching every few windows --
    # into a library of its own: <name>:<flags>)
    extra = os.environ.get("LZ4HIP_SIM_FLAGS", "")
    so = os.path.join(d, "libhostsim%s.so" % (("_" + extra.split(":", 1)[0]) if extra else ""))
    srcs = [os.path.join(d, f) for f in ("hostsim.cpp", "wave_host.h", "group_host.h")] + \
           [os.path.join(ROOT, "lz4-java_amd", "csrc", f) for f in ("lz4_fast_core.h", "lz4_fast_ms_core.h", "lz4_fast_v2_core.h", "lz4_decode_core.h", "lz4_decode_deep.h", "lz4_decode_ring.h", "lz4_decode_wave.h", "lz4_decode_pair.h", "lz4_decode_trio.h", "lz4_hc_core.h")]
    srcs.append(os.path.join(ROOT, "lz4-java_amd", "csrc", "mail_ring.h"))
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread"] + (extra.split(":", 1)[1].split() if extra else []) + ["-o", so, os.path.join(d, "hostsim.cpp")])
    l = C.CDLL(so)
    l.sim_compress_fast.restype = C.c_int
    l.sim_compress_fast.argtypes = [C.c_char_p, C.c_int, _u8p, C.c_int, C.POINTER(C.c_uint64), C.c_uint64]
    l.sim_compress_fast_ms.restype = C.c_int
    l.sim_compress_fast_ms.argtypes = [C.c_char_p, C.c_int, _u8p, C.c_int, C.POINTER(C.c_uint64), C.c_uint64]
    l.sim_compress_fast_v2.restype = C.c_int
    l.sim_compress_fast_v2.argtypes = [C.c_char_p, C.c_int, _u8p, C.c_int, C.POINTER(C.c_uint64), C.c_uint64]
    l.sim_compress_fast_v2raw.restype = C.c_int
    l.sim_compress_fast_v2raw.argtypes = [C.c_char_p, C.c_int, _u8p, C.c_int, C.POINTER(C.c_uint64), C.c_uint64]
    l.sim_compress_fast_probe.restype = C.c_int
    l.sim_compress_fast_probe.argtypes = [C.c_char_p, C.c_int, _u8p, C.c_int, C.c_uint32]
    l.sim_decompress.restype = C.c_int
    l.sim_decompress.argtypes = [C.c_char_p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int]
    l.sim_mail_ring.restype = C.c_int
    l.sim_mail_ring.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), _u8p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint64]
    return l


@pytest.fixture(scope="module")
def sim():
    return load_sim()


def sim_compress(sim, v, cap, seed=0, ms=False, v2=False, raw=False):
    out = (C.c_uint8 * max(cap, 1))()
    st = (C.c_uint64 * 4)()
    f = sim.sim_compress_fast_v2raw if raw else sim.sim_compress_fast_v2 if v2 else (sim.sim_compress_fast_ms if ms else sim.sim_compress_fast)
    r = f(bytes(v), len(v), out, cap, st, seed)
    return r, bytes(out[:max(r, 0)]), list(st)


def sim_decode(sim, c, cap, safe, gl, src_size=None, shift=0):
    """shift: the destination slot starts that many bytes into its buffer (address alignment of the slot)"""
    out = (C.c_uint8 * (max(cap, 1) + shift))()
    C.memset(out, 0xA5, max(cap, 1) + shift)
    dst = (C.c_uint8 * max(cap, 1)).from_buffer(out, shift)
    r = sim.sim_decompress(bytes(c), len(c) if src_size is None else src_size, dst, cap, safe, gl)
    assert bytes(out[:shift]) == b"\xA5" * shift
    return r, bytes(out[shift:shift + cap])


def test_cores_on_issue12_blob(sim, ref):
    """LZ4Test.testRoundtripIssue12 (LZ4Test.java:487-541), bytes [9:], through both compress cores and the decoder"""
    import os
    data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "issue12.bin"), "rb").read()[9:]
    want = ref.compress_fast(data)
    for ms in (False, True):
        r, b, _ = sim_compress(sim, data, ref.compress_bound(len(data)), ms=ms)
        assert b == want, ms
    for gl in (4, 64, 8 | 0x100, 4 | 0x200):
        r, d = sim_decode(sim, want, len(data), 1, gl)
        assert r == len(data) and d == data


def test_compress_core_golden(sim, ref, corpus):
    slow = 0
    for name, v in corpus.items():
        cap = ref.compress_bound(len(v))
        r, b, st = sim_compress(sim, v, cap)
        assert b == ref.compress_fast(v), name
        slow += st[1]
    assert slow > 0  # the collision-resolution path was exercised


def test_compress_core_fuzz(sim, ref, O, corpus):
    """random inputs x {full, tight, random} capacities x random LDS-atomic lane orders"""
    rng = random.Random(3)
    for v in rnd_inputs(O, corpus, 21, 500):
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_fast_raw(v, full)
        for cap in (full, max(0, er + rng.choice([-1, 0, 1, 2, -5, 5, -20, 20])), rng.randrange(0, full + 1)):
            a = ref.compress_fast_raw(v, cap)
            r, b, _ = sim_compress(sim, v, cap, seed=rng.getrandbits(63) | 1)
            assert r == a[0] and (r <= 0 or b == a[1]), (len(v), cap, r, a[0])


def test_compress_ms_core_golden(sim, ref, corpus):
    """window-parallel core (lz4_fast_ms_core.h): all sequences of a 64-position window per step"""
    slow = steps = seqs = 0
    for name, v in corpus.items():
        cap = ref.compress_bound(len(v))
        r, b, st = sim_compress(sim, v, cap, ms=True)
        assert b == ref.compress_fast(v), name
        steps += st[0]; slow += st[1]; seqs += st[3]
    assert slow > 0           # the roll-back path was exercised
    assert seqs > 2 * steps   # and the point of the design: several sequences per step


def test_compress_ms_core_fuzz(sim, ref, O, corpus):
    rng = random.Random(4)
    for v in rnd_inputs(O, corpus, 22, 500):
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_fast_raw(v, full)
        for cap in (full, max(0, er + rng.choice([-1, 0, 1, 2, -5, 5, -20, 20])), rng.randrange(0, full + 1)):
            a = ref.compress_fast_raw(v, cap)
            r, b, _ = sim_compress(sim, v, cap, seed=rng.getrandbits(63) | 1, ms=True)
            assert r == a[0] and (r <= 0 or b == a[1]), (len(v), cap, r, a[0])


def test_compress_v2_core_golden(sim, ref, corpus):
    """lean core (lz4_fast_v2_core.h): minimal finder loop, exact generic step for everything else, sequences parked in lanes and
    written 64 at a time"""
    slow = steps = seqs = 0
    for name, v in corpus.items():
        cap = ref.compress_bound(len(v))
        r, b, st = sim_compress(sim, v, cap, v2=True)
        assert b == ref.compress_fast(v), name
        steps += st[0]; slow += st[1]; seqs += st[3]
    assert slow > 0           # the undo + exact-path replay was exercised
    assert seqs > steps // 2  # and most sequences come from the lean loop


def test_compress_v2_core_fuzz(sim, ref, O, corpus):
    rng = random.Random(5)
    for v in rnd_inputs(O, corpus, 23, 500):
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_fast_raw(v, full)
        for cap in (full, max(0, er + rng.choice([-1, 0, 1, 2, -5, 5, -20, 20])), rng.randrange(0, full + 1)):
            a = ref.compress_fast_raw(v, cap)
            seed = rng.getrandbits(63) | 1
            r, b, _ = sim_compress(sim, v, cap, seed=seed, v2=True)
            assert r == a[0] and (r <= 0 or b == a[1]), (len(v), cap, r, a[0], seed & 2)


def test_compress_v2_raw_parking(sim, ref, O, corpus):
    """the lean core parking bare hits, liblz4's backward extension done for 64 parked hits at once when they are written (what
    the writer wavefront of the default GPU kernel does): same bytes, full and tight capacities"""
    rng = random.Random(6)
    for name, v in corpus.items():
        r, b, _ = sim_compress(sim, v, ref.compress_bound(len(v)), raw=True)
        assert b == ref.compress_fast(v), name
    for v in rnd_inputs(O, corpus, 29, 500):
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_fast_raw(v, full)
        for cap in (full, max(0, er + rng.choice([-1, 0, 1, 2, -5, 5, -20, 20])), rng.randrange(0, full + 1)):
            a = ref.compress_fast_raw(v, cap)
            seed = rng.getrandbits(63) | 1
            r, b, _ = sim_compress(sim, v, cap, seed=seed, raw=True)
            assert r == a[0] and (r <= 0 or b == a[1]), (len(v), cap, r, a[0])


def test_compress_v2_packed_entries(sim, ref, O, corpus):
    """byU32 blocks of at most 4 MiB with the compact table entries of the eight-chain kernel ({position 22 bits, fingerprint 10
    bits}, csrc/lz4_fast_core.h PK): a narrower fingerprint only adds tentative hits that the candidate bytes rule out -- same bytes
    as the reference on synthetic / text / periodic / random blocks, copies at distances around 65535, full and tight capacities"""
    sim.sim_compress_fast_v2pk.restype = C.c_int
    sim.sim_compress_fast_v2pk.argtypes = [C.c_char_p, C.c_int, _u8p, C.c_int, C.POINTER(C.c_uint64), C.c_uint64]
    rng = random.Random(44)
    book = corpus["book1[:200000]"]
    unit = rng.randbytes(900)
    inputs = [O.gen_block(65547, 1), O.gen_block(200000, 2, win=4096), O.gen_block(150000, 3, litmax=4, win=64), book[:150000],
              (book[:70000] + rng.randbytes(500)) * 2, (unit + rng.randbytes(65535 - 900)) * 3, (unit + rng.randbytes(65536 - 900)) * 2 + unit,
              bytes(rng.randrange(2) for _ in range(30000)) * 3, rng.randbytes(70000), (rng.randbytes(41) * 2000)[:80000], O.gen_block(1 << 20, 5, win=65535)]
    slow = 0
    for k, v in enumerate(inputs):
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_fast_raw(v, full)
        for cap in (full, max(0, er + rng.choice([-1, 0, 1, -7]))):
            a = ref.compress_fast_raw(v, cap)
            out = (C.c_uint8 * max(cap, 1))()
            st = (C.c_uint64 * 4)()
            r = sim.sim_compress_fast_v2pk(bytes(v), len(v), out, cap, st, rng.getrandbits(63) | 1)
            assert r == a[0] and (r <= 0 or bytes(out[:r]) == a[1]), (k, len(v), cap, r, a[0])
            slow += st[1]
    assert slow > 0
    assert sim.sim_compress_fast_v2pk(bytes(100), 100, (C.c_uint8 * 200)(), 200, None, 0) == -3


def test_v2_density_probe_routes_blocks(sim, ref, O, corpus):
    """the lean core leaves a block to the window-parallel core exactly when its sequences 32..95 cover fewer than dense64
    bytes (decided when 128 sequences are parked); otherwise it finishes it with the usual bytes"""
    sim.sim_compress_fast_v2_probe.restype = C.c_int
    sim.sim_compress_fast_v2_probe.argtypes = [C.c_char_p, C.c_int, _u8p, C.c_int, C.c_uint32]
    routed = kept = 0
    for name, v in list(corpus.items()) + [("rnd%d" % i, x) for i, x in enumerate(rnd_inputs(O, corpus, 25, 120))]:
        cap = ref.compress_bound(len(v))
        want = ref.compress_fast(v)
        ends = lz4_sequence_ends(want)
        for dense64 in (64 * 8, 64 * 20, 64 * 40):
            out = (C.c_uint8 * max(cap, 1))()
            r = sim.sim_compress_fast_v2_probe(bytes(v), len(v), out, cap, dense64)
            dense = len(ends) >= 128 and ends[95] - ends[31] < dense64
            if dense:
                assert r == -2, (name, dense64, r)
                routed += 1
            else:
                assert r == len(want) and bytes(out[:r]) == want, (name, dense64, r)
                kept += 1
    assert routed > 0 and kept > 0


def lz4_sequence_ends(c):
    """input position after each sequence of an LZ4 block (independent walk of the format)"""
    i = pos = 0
    ends = []
    while i < len(c):
        t = c[i]; i += 1
        l = t >> 4
        if l == 15:
            while True:
                b = c[i]; i += 1; l += b
                if b != 255:
                    break
        i += l; pos += l
        if i >= len(c):
            break
        i += 2
        m = t & 15
        if m == 15:
            while True:
                b = c[i]; i += 1; m += b
                if b != 255:
                    break
        pos += m + 4
        ends.append(pos)
    return ends


def test_density_probe_routes_blocks(sim, ref, O, corpus):
    """adaptive two-pass scheme, pass 1: the one-sequence-per-step core leaves a block to the window-parallel core exactly
    when its sequences 32..95 cover fewer than dense64 bytes; otherwise it finishes it with the usual bytes"""
    routed = {}
    for name, v in list(corpus.items()) + [("rnd%d" % i, x) for i, x in enumerate(rnd_inputs(O, corpus, 24, 120))]:
        cap = ref.compress_bound(len(v))
        expect = ref.compress_fast(v)
        ends = lz4_sequence_ends(expect)
        for dense64 in (64 * 20, 64 * 40):
            out = (C.c_uint8 * max(cap, 1))()
            r = sim.sim_compress_fast_probe(bytes(v), len(v), out, cap, dense64)
            dense = len(ends) >= 96 and ends[95] - ends[31] < dense64
            if dense:
                assert r == -2, (name, dense64)
            else:
                assert r == len(expect) and bytes(out[:r]) == expect, (name, dense64)
            routed[(name, dense64)] = dense
    assert routed[("book1[:65536]", 1280)] and not routed[("gen_block(65536,0)", 1280)] and routed[("gen_block(65536,0)", 2560)]


def test_decode_core_fuzz(sim, ref, O, corpus):
    rng = random.Random(9)
    for v in rnd_inputs(O, corpus, 31, 2500, max_n=20000):
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
        gl = rng.choice([4, 8, 16, 32, 64]) | rng.choice([0, 0x100, 0x200])   # bit 8: the pipelined interior loop, bit 9: output staging
        if rng.random() < 0.25:
            gl = rng.choice([4, 8, 16]) | 0x400                                  # bit 10: the deep interior loop
        elif rng.random() < 0.33:
            gl = rng.choice([1, 4, 8, 16])
            gl |= 0x800 | (rng.choice([8, 9, 10] if gl == 1 else [9, 10, 11, 12]) << 12)   # (the output ring holds at least eight steps)   # bit 11: the ring loop; bits 12..15: log2 of its output ring
        r2, d2 = ref.decompress_safe_raw(c, cap)
        r1, d1 = sim_decode(sim, c, cap, 1, gl)
        assert r1 == r2 and (r2 < 0 or d1[:r2] == d2[:r2]), ("safe", mode, gl, len(v), cap, r1, r2)
        # fast decoder: bounded-input semantics are defined by the oracle port
        scap = max(len(c) + rng.choice([0, 0, 0, 3, 16, -1]), 0)
        r3, d3 = O.decompress_fast_bounded(c, scap, cap)
        padded = c + bytes(max(0, scap - len(c)))
        r4, d4 = sim_decode(sim, padded, cap, 0, gl, src_size=scap)
        assert r3 == r4 and (r3 < 0 or d3[:cap] == d4[:cap]), ("fast", mode, gl, len(v), cap, scap, r3, r4)
        if mode == 0 and scap >= len(c):
            assert ref.decompress_fast_raw(c, cap)[0] == r4


def _lz4_seq(lit, ml, off, rng):
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


def test_staged_decoder_flush_past_position(sim, ref):
    """round-1 advisor finding: in the staged interior loop st_lits may flush whole lines PAST the sequence start; a sequence
    that then leaves the loop (match length >= 274: second extension byte) made st_flush_all run with fl > op -- a 4 GiB
    out-of-slot copy.  Valid blocks: a few short sequences, then literals 65..269 + a match of >= 274 bytes, at every
    destination alignment the staging cares about."""
    rng = random.Random(1234)
    for trial in range(400):
        c = bytearray()
        n = 0
        for _ in range(rng.randrange(2, 13)):
            lit, ml = rng.randrange(1, 60), rng.randrange(4, 100)
            c += _lz4_seq(lit, ml, rng.randrange(1, n + lit + 1), rng); n += lit + ml
        lit, ml = rng.randrange(65, 270), rng.randrange(274, 900)
        c += _lz4_seq(lit, ml, rng.randrange(1, n + lit + 1), rng); n += lit + ml
        for _ in range(rng.randrange(8, 40)):     # enough tail for the interior loop to be entered at all (>= 606 bytes from the end)
            lit, ml = rng.randrange(1, 60), rng.randrange(4, 100)
            c += _lz4_seq(lit, ml, rng.randrange(1, n + lit + 1), rng); n += lit + ml
        last = rng.randrange(5, 40)
        c += bytes([last << 4 if last < 15 else 0xF0]) + (bytes([last - 15]) if last >= 15 else b"") + rng.randbytes(last); n += last
        c = bytes(c)
        want_r, want = ref.decompress_safe_raw(c, n)
        assert want_r == n
        for gl in (4 | 0x200, 8 | 0x200, 16 | 0x200, 64 | 0x200):
            r, d = sim_decode(sim, c, n, 1, gl)
            assert r == n and d == want, (trial, gl, r)


def test_deep_decoder_loop(sim, ref, O, corpus):
    """The deep interior loop (csrc/lz4_decode_deep.h: stream staged in an LDS ring, three match sources in flight, whole-step
    unconditional loads / stores) in the lock-step simulator, groups of 4 / 8 / 16 lanes: real and synthetic blocks whose streams
    are long enough for the loop to run (it needs 2 KB of stream ahead), hand-assembled streams that mix one-step sequences with
    long literal runs, long matches, offsets shorter than a step and offsets that reach into the sequences still waiting in a
    slot -- and the same streams corrupted / truncated: return codes and bytes against the reference library (safe) and the C
    restatement's bounded fast decoder.  Every access outside the block's slots or the ring counts as a failure."""
    rng = random.Random(4242)
    from conftest import deep_decoder_cases
    valid, cases = deep_decoder_cases(ref, O, corpus, rng, _lz4_seq)
    sim.sim_deep_trips.restype = C.c_ulonglong
    trips0 = sim.sim_deep_trips()
    for k, (c, cap) in enumerate(cases):
        want_r, want = ref.decompress_safe_raw(c, cap)
        for gl in ((4, 8, 16) if k < len(valid) else (rng.choice([4, 8, 16]),)):
            r, d = sim_decode(sim, c, cap, 1, gl | 0x400)
            assert r == want_r and (want_r < 0 or d[:want_r] == want[:want_r]), ("safe", k, gl, len(c), cap, r, want_r)
            scap = len(c) + rng.choice([0, 0, 5, 64])
            r3, d3 = O.decompress_fast_bounded(c, scap, cap)
            r4, d4 = sim_decode(sim, c + bytes(scap - len(c)), cap, 0, gl | 0x400, src_size=scap)
            assert r3 == r4 and (r3 < 0 or d3[:cap] == d4[:cap]), ("fast", k, gl, len(c), cap, scap, r3, r4)
    assert sim.sim_deep_trips() - trips0 > 500000   # (the loop under test did the work)


def test_ring_decoder_loop(sim, ref, O, corpus):
    """The ring loop (csrc/lz4_decode_ring.h: stream AND recent output in LDS rings, near matches copied inside the ring -- incl. the
    ones that overlap their own output --, far ones pipelined through slots, output leaving as address-aligned 64-byte steps) in the
    lock-step simulator: groups of 4 / 8 / 16 lanes, output rings of 512 / 1024 / 4096 bytes, destination slots at every kind of
    address alignment; the long-stream cases of the deep loop's test (real and synthetic blocks, hand-assembled mixes of every kind
    of sequence, the same streams corrupted / truncated / with wrong capacities) plus short-offset text-like data that lives in the
    ring: return codes and bytes against the reference library (safe) and the C restatement's bounded fast decoder.  Every access
    outside the block's slots or its LDS bytes counts as a failure."""
    rng = random.Random(777)
    from conftest import deep_decoder_cases
    valid, cases = deep_decoder_cases(ref, O, corpus, rng, _lz4_seq)
    extra = [corpus["book1[:200000]"][1000:40000], O.gen_block(50000, 7, litmax=3, win=40), O.gen_block(50000, 8, litmax=20, win=500),
             O.gen_block(80000, 9, win=4096), bytes(rng.randrange(4) for _ in range(30000)), (rng.randbytes(37) * 3000)[:70000]]
    for v in extra:
        c = ref.compress_fast(v)
        valid.append((c, len(v))); cases.insert(len(valid) - 1, (c, len(v)))
    sim.sim_ring_trips.restype = C.c_ulonglong
    trips0 = sim.sim_ring_trips()
    for k, (c, cap) in enumerate(cases):
        want_r, want = ref.decompress_safe_raw(c, cap)
        full = k < len(valid)
        for gl, rl in (((4, 9), (8, 12), (16, 12), (8, 9), (4, 10), (1, 8), (1, 9)) if full else ((rng.choice([1, 4, 8, 16]), rng.choice([9, 10, 12])),)):
            flag = gl | 0x800 | (rl << 12)
            shift = rng.choice([0, 0, 1, 7, 16, 33, 63, 64, 100])
            r, d = sim_decode(sim, c, cap, 1, flag, shift=shift)
            assert r == want_r and (want_r < 0 or d[:want_r] == want[:want_r]), ("safe", k, gl, rl, shift, len(c), cap, r, want_r)
            scap = len(c) + rng.choice([0, 0, 5, 64])
            r3, d3 = O.decompress_fast_bounded(c, scap, cap)
            r4, d4 = sim_decode(sim, c + bytes(scap - len(c)), cap, 0, flag, src_size=scap, shift=shift)
            assert r3 == r4 and (r3 < 0 or d3[:cap] == d4[:cap]), ("fast", k, gl, rl, len(c), cap, scap, r3, r4)
    assert sim.sim_ring_trips() - trips0 > 500000   # (the loop under test did the work)


def wave_flag(log, ks1k=False, par=False):
    """sim_decompress flag of the wave loop: 64 lanes, bit 16, bits 17..21 = log2 of the output ring, bit 22 = 1 KB stream ring,
    bit 23 = the parallel loop (several sequences of the block per trip); par == "pair": bit 24 as well = the PAIR loop
    (csrc/lz4_decode_pair.h: a parser and a copier wavefront per block -- two host threads over one block of "LDS"); par == "trio":
    bit 25 = the TRIO loop (csrc/lz4_decode_trio.h: scanner, planner, copier -- three host threads)"""
    return 64 | 0x10000 | (log << 17) | (0x400000 if ks1k else 0) | (0x800000 if par else 0) | (0x1000000 if par == "pair" else 0) | (0x2000000 if par == "trio" else 0)


@pytest.mark.parametrize("par", [False, True, "pair", "trio"])
def test_wave_decoder_loop(sim, ref, O, corpus, par):
    """The wave loop (csrc/lz4_decode_wave.h: one wavefront per block, stream ring + an output ring of 4 .. 64 KB in LDS, wave-uniform
    parse, pieces of up to 252 bytes stored as aligned dwords, far sources from flushed memory, byte-exact entry / exit flushes) in
    the lock-step simulator: the long-stream cases of the deep and ring loops' tests (real and synthetic blocks, hand-assembled
    mixes of every kind of sequence -- literal runs over 252 bytes, matches that overlap their own output, long matches, offsets
    beyond every ring size --, the same streams corrupted / truncated / with wrong capacities) plus text-like and run-heavy data,
    at every kind of address alignment of the destination slot: return codes and bytes against the reference library (safe) and
    the C restatement's bounded fast decoder.  Every access outside the block's slots or the wavefront's LDS bytes, every
    unaligned store by lanes 1.., every mirror store the device's wave-uniform pre-test would have skipped counts as a failure.
    par: the PARALLEL loop -- every sequence that starts in a 256-byte window of the stream per trip (speculative lane-parallel
    discovery, scalar walk, records by lane shuffle, prefix-summed output positions, the dependency rule, exact lane-per-run
    copies), with the one-sequence step for what a trip cannot start with."""
    rng = random.Random(20250924)
    from conftest import deep_decoder_cases
    valid, cases = deep_decoder_cases(ref, O, corpus, rng, _lz4_seq)
    extra = [corpus["book1[:200000]"][1000:40000], O.gen_block(50000, 7, litmax=3, win=40), O.gen_block(50000, 8, litmax=20, win=500),
             O.gen_block(150000, 9, win=4096), O.gen_block(65536, 10), O.gen_block(200000, 11, litmax=300, win=65535),
             bytes(rng.randrange(4) for _ in range(30000)), (rng.randbytes(37) * 3000)[:70000],
             (rng.randbytes(255) * 400)[:90000], (rng.randbytes(3) * 40000)[:100000]]
    for v in extra:
        c = ref.compress_fast(v)
        valid.append((c, len(v))); cases.insert(len(valid) - 1, (c, len(v)))
    st = (C.c_ulonglong * 4)()
    sim.sim_wave_stats(st)
    trips0, far0, mir0 = st[0], st[2], st[3]
    ps0 = (C.c_ulonglong * 3)()
    sim.sim_wave_par_stats(ps0)
    for k, (c, cap) in enumerate(cases):
        want_r, want = ref.decompress_safe_raw(c, cap)
        full = k < len(valid)
        for log, ks1k in (((12, False), (13, True), (14, False), (16, False)) if full else ((rng.choice([12, 13, 14, 15, 16]), rng.random() < 0.3),)):
            flag = wave_flag(log, ks1k, par)
            shift = rng.choice([0, 0, 1, 2, 3, 7, 16, 33, 63, 64, 100, 255, 256, 257])
            r, d = sim_decode(sim, c, cap, 1, flag, shift=shift)
            assert r == want_r and (want_r < 0 or d[:want_r] == want[:want_r]), ("safe", k, log, ks1k, shift, len(c), cap, r, want_r)
            scap = len(c) + rng.choice([0, 0, 5, 64])
            r3, d3 = O.decompress_fast_bounded(c, scap, cap)
            r4, d4 = sim_decode(sim, c + bytes(scap - len(c)), cap, 0, flag, src_size=scap, shift=shift)
            assert r3 == r4 and (r3 < 0 or d3[:cap] == d4[:cap]), ("fast", k, log, ks1k, len(c), cap, scap, r3, r4)
    sim.sim_wave_stats(st)
    if not par:
        assert st[0] - trips0 > 500000 and st[2] - far0 > 1000 and st[3] - mir0 > 1000   # the loop did the work: pieces, far sources, ring wraps
    else:
        ps = (C.c_ulonglong * 3)()
        sim.sim_wave_par_stats(ps)
        trips, seqs = ps[0] - ps0[0], ps[1] - ps0[1]
        assert trips > (50000 if par in (True, "trio") else 100000) and seqs > 1.2 * trips, (trips, seqs)   # (the one-wavefront loop and the trio copy self-overlapping matches inside their passes: up to a third fewer passes on this corpus)
        if par is True or par == "trio":   # windows full of sequences take the walk by pointer doubling (group_dev.h vwalk_par): the backend runs it NEXT TO the plain walk and any difference fails the decode
            sim.sim_walk_par_calls.restype = C.c_ulonglong
            assert sim.sim_walk_par_calls() > 2000, sim.sim_walk_par_calls()   # trips did the work (this corpus is mostly irregular streams: App. F data runs 9-12 sequences per trip, text 18)
        if par is True:   # ... and go on in the loop's SHORT instance, whose passes pick the form of their copy rounds by their longest run (the backend checks every round's lanes against the form)
            sim.sim_short_rounds.restype = C.c_ulonglong
            assert sim.sim_short_rounds() > 2000, sim.sim_short_rounds()


@pytest.mark.parametrize("par", [False, True, "pair", "trio"])
def test_wave_decoder_small_and_fuzz(sim, ref, O, corpus, par):
    """the wave loop's instantiation of decode_block on everything the other decoders' fuzz test sees (short, empty, damaged and
    random streams: mostly the exact tiers with 64 lanes, the wave loop where a stream is long enough)"""
    rng = random.Random(99)
    for v in rnd_inputs(O, corpus, 57, 1200, max_n=40000):
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
        flag = wave_flag(rng.choice([12, 13, 16]), rng.random() < 0.5, par)
        shift = rng.choice([0, 1, 5, 64, 131])
        r2, d2 = ref.decompress_safe_raw(c, cap)
        r1, d1 = sim_decode(sim, c, cap, 1, flag, shift=shift)
        assert r1 == r2 and (r2 < 0 or d1[:r2] == d2[:r2]), ("safe", mode, len(v), cap, r1, r2)
        scap = max(len(c) + rng.choice([0, 0, 0, 3, 16, -1]), 0)
        r3, d3 = O.decompress_fast_bounded(c, scap, cap)
        r4, d4 = sim_decode(sim, c + bytes(max(0, scap - len(c))), cap, 0, flag, src_size=scap, shift=shift)
        assert r3 == r4 and (r3 < 0 or d3[:cap] == d4[:cap]), ("fast", mode, len(v), cap, scap, r3, r4)


def test_wave_par_trip_behind_a_one_sequence_step(sim, ref, O):
    """conftest.wild_piece_stream: the parallel wave loop right behind a sequence of its one-sequence step, sources at the edge of what
    the ring still holds -- every ring size, both stream rings"""
    from conftest import wild_piece_stream
    rng = random.Random(505)
    for log in (12, 13, 14, 15, 16):
        for rep in range(3):
            c, n = wild_piece_stream(1 << log, rng)
            want_r, want = ref.decompress_safe_raw(c, n)
            assert want_r == n
            for lg in sorted({log, 13, 16}):
                for ks1k in (False, True):
                    for par in (True, "pair", "trio"):
                        r, d = sim_decode(sim, c, n, 1, wave_flag(lg, ks1k, par), shift=rng.choice([0, 3, 64, 131]))
                        assert r == n and d[:n] == want, (log, rep, lg, ks1k, par, r, n, next((i for i in range(min(r, n)) if d[i] != want[i]), None))


def test_wave_loops_ring_edge_streams(sim, ref):
    """conftest.ring_edge_stream through both wave loops at every ring size: distances at the edge of what a ring holds, one-sequence
    steps between short trips, slow copies, early trip ends"""
    from conftest import ring_edge_stream
    rng = random.Random(7117)
    for rep in range(6):
        c, n = ring_edge_stream(rng, rng.choice([20000, 90000, 150000]))
        want_r, want = ref.decompress_safe_raw(c, n)
        assert want_r == n
        for log in (12, 13, 14, 15, 16):
            for par in (True, "pair", "trio", False) if log in (13, 16) else (True, "pair", "trio"):
                r, d = sim_decode(sim, c, n, 1, wave_flag(log, rng.random() < 0.5, par), shift=rng.choice([0, 3, 64, 131, 255]))
                assert r == n and d[:n] == want, (rep, log, par, r, n, next((i for i in range(min(max(r, 0), n)) if d[i] != want[i]), None))


@pytest.mark.parametrize("loop", ["trio", "par"])
def test_trio_end_of_block_rules(sim, loop):
    """The trio loop's planner -- and, with the same rule, the one-wavefront parallel loop (csrc/lz4_decode_wave.h) -- ends the loop by liblz4's OWN fast-loop conditions, sequence by sequence (literals that end 32 bytes in front
    of the stream's end, a match that ends more than 64 bytes in front of the output's end: csrc/lz4_decode_trio.h) instead of the other
    interior loops' blanket 306 / 606-byte margin -- so which tier of liblz4 meets a defect near a block's end must come out the same.
    A slice of tools/trio_end_soak.py (70000 cases without a mismatch when the rule went in): streams of five kinds, capacities off by
    -607 .. +700, damage / truncation in the last 400 stream bytes, extension, damage anywhere; safe decoder against the reference
    library's codes and bytes, bounded fast decoder against the C restatement."""
    import subprocess, sys
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "trio_end_soak.py"), "20260930", "1200", loop], timeout=600).decode()
    assert "bad 0" in out and "cases 1200" in out, out[-400:]


def test_wave_loop_switches_instances_on_any_data():
    """The parallel wave loop exists in two instances (csrc/lz4_decode_wave.h: plain, and SHORT -- copy rounds in the form their longest run allows);
    decode_block moves a block from the first to the second where the loop leaves it, the second starting with empty rings.  A build of the simulator
    in which that happens on every kind of data and at once (full from 4 starts per window on, after 1 window), a slice of the end-of-block soak on
    it: the reference library's codes and bytes."""
    import subprocess, sys
    env = dict(os.environ, LZ4HIP_SIM_FLAGS="switchy:-DLZ4HIP_SHORT_AFTER=1u -DLZ4HIP_SHORT_MIN_T=4u")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "trio_end_soak.py"), "20260931", "800", "par"], timeout=900, env=env).decode()
    assert "bad 0" in out and "cases 800" in out, out[-400:]


def test_decode_core_malformed_vectors(sim, golden):
    for v in golden["malformed"]:
        vec = bytes.fromhex(v["hex"])
        for gl in (4, 16, 64, 8 | 0x100, 64 | 0x100, 4 | 0x200, 16 | 0x200, 8 | 0x400):
            r, d = sim_decode(sim, vec, v["safe_cap"], 1, gl)
            assert r == v["safe_ret"]
            if r >= 0:
                assert d[:r].hex() == v["safe_out_hex"]


def sim_hc(sim, v, level, cap, seed=0):
    sim.sim_compress_hc.restype = C.c_int
    sim.sim_compress_hc.argtypes = [C.c_char_p, C.c_int, _u8p, C.c_int, C.c_int, C.c_uint64]
    out = (C.c_uint8 * max(cap, 1))()
    r = sim.sim_compress_hc(bytes(v), len(v), out, cap, level, seed)
    return r, bytes(out[:max(r, 0)])


def test_hc_core_golden(sim, ref, golden, corpus):
    from conftest import sha
    for name, v in corpus.items():
        r, b = sim_hc(sim, v, 9, ref.compress_bound(len(v)))
        assert (r, sha(b)) == (golden["inputs"][name]["hc9_size"], golden["inputs"][name]["hc9_sha256"]), name


def test_hc_core_fuzz(sim, ref, O, corpus):
    """delta[] builder + lazy parse vs liblz4: levels 1..9, level clamps, limited output, pattern-heavy inputs,
    random LDS-atomic orders"""
    rng = random.Random(23)
    for v in rnd_inputs(O, corpus, 61, 300):
        lvl = rng.choice([1, 2, 3, 5, 8, 9, 9, 9, 0, -2])
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_hc_raw(v, lvl, full)
        for cap in (full, max(0, er + rng.choice([-1, 0, 1, -9, 9])), rng.randrange(0, full + 1)):
            a = ref.compress_hc_raw(v, lvl, cap)
            r, b = sim_hc(sim, v, lvl, cap, seed=rng.getrandbits(63) | 1)
            assert r == a[0] and (r <= 0 or b == a[1]), (len(v), lvl, cap, r, a[0])
    for period in (1, 2, 3, 4, 7):
        p = rng.randbytes(period)
        for n in (3000, 70000):
            v = bytearray((p * (n // period + 1))[:n])
            for _ in range(n // 2500):
                v[rng.randrange(n)] ^= 0x33
            v = bytes(v)
            assert sim_hc(sim, v, 9, ref.compress_bound(n))[1] == ref.compress_hc(v, 9), (period, n)


def test_hc_core_optimal_parser(sim, ref, golden, corpus, O):
    """levels 10..12 (lz4-java 10..17): delta[] builder + optimal parser (price table, chain swap, pattern analysis) vs liblz4"""
    from conftest import sha
    for name, v in corpus.items():
        if len(v) > 300000:
            continue   # (the 1 MiB input takes the lock-step simulator minutes at level 12; the GPU suite covers it)
        for lvl in (10, 12):
            r, b = sim_hc(sim, v, lvl, ref.compress_bound(len(v)))
            assert (r, sha(b)) == (golden["inputs"][name]["hc%d_size" % lvl], golden["inputs"][name]["hc%d_sha256" % lvl]), (name, lvl)
    rng = random.Random(31)
    for v in rnd_inputs(O, corpus, 62, 120):
        lvl = rng.choice([10, 11, 12, 17])
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_hc_raw(v, lvl, full)
        for cap in (full, max(0, er + rng.choice([-1, 0, 1, -9, 9]))):
            a = ref.compress_hc_raw(v, lvl, cap)
            r, b = sim_hc(sim, v, lvl, cap, seed=rng.getrandbits(63) | 1)
            assert r == a[0] and (r <= 0 or b == a[1]), (len(v), lvl, cap, r, a[0])
    for period in (1, 2, 4, 7):
        p = rng.randbytes(period)
        n = 20000
        v = bytearray((p * (n // period + 1))[:n])
        for _ in range(8):
            v[rng.randrange(n)] ^= 0x33
        v = bytes(v)
        for lvl in (10, 12):
            assert sim_hc(sim, v, lvl, ref.compress_bound(n))[1] == ref.compress_hc(v, lvl), (period, lvl)




def run_mail_ring(sim, blocks, caps, pairs, dense64, seed, writers=None):
    """writers: None = a writer per finder (sim_mail_ring); k = `pairs` finders on k writers that serve two rings each (sim_mail_ring_shared)"""
    n = len(blocks)
    src = b"".join(blocks)
    so, do, p, q = [], [], 0, 0
    for b, c in zip(blocks, caps):
        so.append(p); do.append(q); p += len(b); q += c
    dst = (C.c_uint8 * max(q, 1))()
    out = (C.c_int32 * n)(*([-99] * n))
    routed = (C.c_uint32 * n)()
    nr = C.c_uint32(0)
    if writers is None:
        rc = sim.sim_mail_ring(src, (C.c_uint64 * n)(*so), (C.c_int32 * n)(*[len(b) for b in blocks]), dst, (C.c_uint64 * n)(*do),
                               (C.c_int32 * n)(*caps), out, n, pairs, dense64, routed if dense64 else None, C.byref(nr), seed)
    else:
        sim.sim_mail_ring_shared.restype = C.c_int
        sim.sim_mail_ring_shared.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), _u8p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint64]
        rc = sim.sim_mail_ring_shared(src, (C.c_uint64 * n)(*so), (C.c_int32 * n)(*[len(b) for b in blocks]), dst, (C.c_uint64 * n)(*do),
                                      (C.c_int32 * n)(*caps), out, n, pairs, writers, dense64, routed if dense64 else None, C.byref(nr), seed)
    assert rc == 0, "a simulated wavefront touched memory outside its block / slot"
    res = [(out[i], bytes(dst[do[i]:do[i] + max(out[i], 0)])) for i in range(n)]
    return res, sorted(routed[:nr.value])


def test_mail_ring_two_threads(sim, ref, O, corpus):
    """The finder / writer hand-over of the default compress kernel (csrc/mail_ring.h: MailOutT + mail_writer_t, the source the
    GPU runs) with HOST threads -- a finder and a writer thread per pair, two or three pairs on one block queue, a ring of two
    slots (every third post wraps and meets back-pressure), random naps around every publish: blocks of every size class incl.
    empty, shorter than a batch, several batches, byU32; full and tight capacities; all bytes against the reference library."""
    rng = random.Random(7)
    book = corpus["book1[:200000]"]
    blocks = [b"", b"a" * 5, b"b" * 12, b"c" * 13, bytes(rng.randrange(2) for _ in range(500)), O.gen_block(1000, 3), O.gen_block(20000, 4),
              O.gen_block(65536, 5), O.gen_block(65536, 6, litmax=4, win=64), rng.randbytes(9000), book[1000:31000], O.gen_block(70000, 8, win=4096),
              bytes(30000), O.gen_block(65546, 9), corpus["pic[:65536]"][:40000], O.gen_block(3000, 10, litmax=2, win=8)] * 2
    rng.shuffle(blocks)
    for pairs, seed in ((1, 1), (2, 2), (3, 3), (2, 4)):
        caps = []
        for v in blocks:
            full = ref.compress_bound(len(v))
            er, _ = ref.compress_fast_raw(v, full)
            caps.append(rng.choice([full, full, max(0, er - 1), er, er + 3, rng.randrange(0, full + 1)]))
        res, routed = run_mail_ring(sim, blocks, caps, pairs, 0, seed)
        assert routed == []
        for v, cap, (r, c) in zip(blocks, caps, res):
            er, eb = ref.compress_fast_raw(v, cap)
            assert r == er and (er <= 0 or c == eb[:er]), (pairs, seed, len(v), cap, r, er)


def test_mail_ring_writers_shared_by_two_finders(sim, ref, O, corpus):
    """The ten-chain kernel of packed byU32 blocks has ten finders and six writers: a writer serves the rings of TWO finders
    (csrc/mail_ring.h mail_writer2_t).  Host threads again -- 2 finders on 1 writer, 3 on 2, 5 on 3 -- over blocks of every size class
    (those of 65547 bytes .. 4 MiB with the packed table entries), full and tight capacities, with and without the density probe."""
    rng = random.Random(17)
    book = corpus["book1[:200000]"]
    blocks = [b"", b"c" * 13, O.gen_block(1000, 3), O.gen_block(70000, 4, win=4096), O.gen_block(65547, 5), O.gen_block(200000, 6),
              book[:90000], O.gen_block(65536, 7), (rng.randbytes(300) + bytes(65300)) * 2, O.gen_block(120000, 8, litmax=4, win=64),
              rng.randbytes(70000), O.gen_block(3000, 10, litmax=2, win=8), O.gen_block(300000, 11, win=65535)] * 2
    rng.shuffle(blocks)
    for finders, writers, seed in ((2, 1, 1), (3, 2, 2), (5, 3, 3), (2, 2, 4)):
        caps = []
        for v in blocks:
            full = ref.compress_bound(len(v))
            er, _ = ref.compress_fast_raw(v, full)
            caps.append(rng.choice([full, full, max(0, er - 1), er, er + 3]))
        res, routed = run_mail_ring(sim, blocks, caps, finders, 0, seed, writers=writers)
        assert routed == []
        for v, cap, (r, c) in zip(blocks, caps, res):
            er, eb = ref.compress_fast_raw(v, cap)
            assert r == er and (er <= 0 or c == eb[:er]), (finders, writers, seed, len(v), cap, r, er)
    caps = [ref.compress_bound(len(v)) for v in blocks]
    res, routed = run_mail_ring(sim, blocks, caps, 3, 64 * 20, 9, writers=2)
    assert routed
    for i, (v, (r, c)) in enumerate(zip(blocks, res)):
        assert (r == -2) if i in routed else (c == ref.compress_fast(v)), (i, len(v))


def test_mail_ring_abort_after_routed_block(sim, ref, O, corpus):
    """with the density probe on, blocks of short sequences are ABORT messages (the writer forgets them mid-block, after it may
    already have written batches of theirs) and land in the routed list; the blocks before and behind them on the same pair
    come out right, and EXIT ends every writer"""
    rng = random.Random(9)
    book = corpus["book1[:200000]"]
    dense = [book[i * 3000:i * 3000 + 65536] for i in range(4)] + [bytes(rng.randrange(3) for _ in range(40000))]
    sparse = [O.gen_block(65536, 20 + i) for i in range(4)] + [O.gen_block(20000, 30), b"", O.gen_block(200, 31)]
    blocks = []
    for i in range(max(len(dense), len(sparse))):
        blocks += dense[i:i + 1] + sparse[i:i + 1]
    caps = [ref.compress_bound(len(v)) for v in blocks]
    for pairs, seed in ((1, 5), (2, 6)):
        res, routed = run_mail_ring(sim, blocks, caps, pairs, 64 * 20, seed)
        assert routed, "no block was routed: the ABORT path was not exercised"
        for i, (v, (r, c)) in enumerate(zip(blocks, res)):
            if i in routed:
                assert r == -2
            else:
                assert c == ref.compress_fast(v), (pairs, i, len(v))
        assert any(any(v is d for d in dense) for i, v in enumerate(blocks) if i in routed)


# ---------------------------------------------------------------------------------------------------------------------
# The hand-scheduled match-finder loops (csrc/lz4_fast_v2_asm.h, lz4_fast_v2_asm32.h, lz4_fast_v2_asm_body.inc) on the CPU:
# their TEXT, extracted from the preprocessed device headers, is run by an interpreter of its instructions (tests/hostsim/asm_emu.h)
# inside the lean core's lock-step simulation, wherever the GPU build runs the assembled loop.
# ---------------------------------------------------------------------------------------------------------------------
def build_asmsim(tag="", defs=()):
    """the interpreter library; tag / defs: a variant -- the loops' text AND the C++ around it built with extra -D flags"""
    d = os.path.join(ROOT, "tests", "hostsim")
    inc, so = os.path.join(d, "lean_asm_text%s.inc" % tag), os.path.join(d, "libhostsim_asm%s.so" % tag)
    csrc = os.path.join(ROOT, "lz4-java_amd", "csrc")
    srcs = [os.path.join(d, f) for f in ("hostsim_asm.cpp", "asm_emu.h", "wave_host.h", "gen_asm_text.py")] + \
           [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".inc"))]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        rc = subprocess.call([os.sys.executable, os.path.join(d, "gen_asm_text.py"), inc] + list(defs))
        if rc == 3:
            pytest.skip("no hipcc here: the loops' text cannot be extracted")
        assert rc == 0
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", '-DLZ4HIP_ASM_TEXT_INC="%s"' % os.path.basename(inc)] + list(defs) +
                              ["-o", so, os.path.join(d, "hostsim_asm.cpp")])
    l = C.CDLL(so)
    l.sim_asm_compress.restype = C.c_int
    l.sim_asm_compress.argtypes = [C.c_char_p, C.c_int, _u8p, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_uint64]
    return l


@pytest.fixture(scope="module")
def asmsim():
    return build_asmsim()


def asm_compress(asmsim, v, cap, kind=0, seed=1):
    out = (C.c_uint8 * max(cap, 1))()
    st = (C.c_uint64 * 3)()
    r = asmsim.sim_asm_compress(bytes(v), len(v), out, cap, kind, st, seed)
    return r, bytes(out[:max(r, 0)]), list(st)


def test_asm_loop_byu16_golden_and_fuzz(asmsim, ref, O, corpus):
    """byU16 blocks through the lean core WITH the hand-scheduled loop (interpreted): golden corpus, the regression inputs that once
    broke it on the GPU, 400 fuzz inputs x {full, tight, random} capacities, each run with its own order of the LDS atomics' lanes
    and its own garbage in the registers the loop does not set: return values and bytes against the reference library"""
    import glob
    parked = hits = 0
    for name, v in corpus.items():
        if len(v) >= 65547:
            continue
        r, b, st = asm_compress(asmsim, v, ref.compress_bound(len(v)))
        assert b == ref.compress_fast(v), name
        parked += st[2]
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "regress", "*.bin"))):
        v = open(f, "rb").read()
        r, b, st = asm_compress(asmsim, v, ref.compress_bound(len(v)), seed=5)
        assert b == ref.compress_fast(v), f
    rng = random.Random(71)
    for v in rnd_inputs(O, corpus, 81, 400):
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_fast_raw(v, full)
        for cap in (full, max(0, er + rng.choice([-1, 0, 1, 2, -5, 5])), rng.randrange(0, full + 1)):
            a = ref.compress_fast_raw(v, cap)
            r, b, st = asm_compress(asmsim, v, cap, seed=rng.getrandbits(62) | 1)
            assert r == a[0] and (r <= 0 or b == a[1]), (len(v), cap, r, a[0])
            parked += st[2]
    assert parked > 20000   # (the loop under test found the hits)


def test_asm_loop_byu32_both_entry_kinds(asmsim, ref, O, corpus):
    """blocks of 65547 bytes and more: the loop with packed 32-bit entries (blocks up to 4 MiB: kind 0) and with 64-bit entries
    (kind 1) -- synthetic blocks of every literal / window mix, text, periodic and random data, copies at distances around 65535"""
    rng = random.Random(72)
    book = corpus["book1[:200000]"]
    unit = rng.randbytes(700)
    inputs = [O.gen_block(65547, 1), O.gen_block(200000, 2, win=4096), O.gen_block(150000, 3, litmax=4, win=64), book[:120000], O.gen_block(100000, 4, litmax=200, win=300),
              (unit + rng.randbytes(65535 - 700)) * 2 + unit, (unit + rng.randbytes(65536 - 700)) * 2 + unit, bytes(rng.randrange(2) for _ in range(30000)) * 3,
              rng.randbytes(70000), (rng.randbytes(41) * 2000)[:80000], O.gen_block(300000, 5, win=65535), corpus["pic[:65536]"] * 2]
    parked = 0
    for k, v in enumerate(inputs):
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_fast_raw(v, full)
        for kind in (0, 1):
            for cap in (full, max(0, er + rng.choice([-1, 0, 1, -7]))):
                a = ref.compress_fast_raw(v, cap)
                r, b, st = asm_compress(asmsim, v, cap, kind=kind, seed=rng.getrandbits(62) | 1)
                assert r == a[0] and (r <= 0 or b == a[1]), (k, len(v), kind, cap, r, a[0])
                parked += st[2]
    assert parked > 20000


def test_asm_loop_retry_path_with_narrow_fingerprints(ref, O, corpus):
    """LZ4HIP_V2_RETRY = 1 (not the product's default yet: it has not been timed on the GPU): a FALSE hit -- fingerprints agree, the
    candidate's bytes differ -- is handled inside the hand-scheduled loop (the probes behind it come out of the lookups already made,
    which therefore start one position earlier; a retry commit has its own undo record) instead of leaving the loop.  Built with
    6-bit byU16 fingerprints so that a fifth of the searches meet a false hit (the packed byU32 entries have ten bits anyway): the
    loops' text in the interpreter against the reference library, fuzz inputs x {full, tight} capacities, both table kinds."""
    sim = build_asmsim("_retry", ("-DLZ4HIP_V2_RETRY=1", "-DLZ4HIP_FP_BITS=6"))
    rng = random.Random(73)
    inputs = [O.gen_block(65536, 0), O.gen_block(65536, 1), corpus["geo[:65536]"], corpus["book1[:200000]"][:65536], corpus["pic[:65536]"],
              O.gen_block(200000, 2, win=4096), O.gen_block(150000, 3, litmax=4, win=64), corpus["book1[:200000]"][:120000], O.gen_block(300000, 5, win=65535)]
    inputs += rnd_inputs(O, corpus, 85, 300)
    parked = calls = 0
    for k, v in enumerate(inputs):
        full = ref.compress_bound(len(v))
        er, _ = ref.compress_fast_raw(v, full)
        for cap in (full, max(0, er + rng.choice([-1, 0, 1, -6]))):
            a = ref.compress_fast_raw(v, cap)
            r, b, st = asm_compress(sim, v, cap, seed=rng.getrandbits(62) | 1)
            assert r == a[0] and (r <= 0 or b == a[1]), (k, len(v), cap, r, a[0])
            calls += st[0]; parked += st[2]
    assert parked > 50000


def test_wave_walk_asm_text_in_the_interpreter(asmsim):
    """The hand-written walk of the parallel wave decoder (lz4-java_amd/csrc/group_dev.h vwalk: four hops per end test, position 255
    its own successor, the count = the first lane that holds 255) -- its TEXT, extracted from the preprocessed device header, run by
    the ISA interpreter against the plain definition of the walk (and thereby against its C++ twin of the lane simulator, which hops
    in the same groups): random next-position tables of every density, chains that fill all 64 lanes, chains that end at once."""
    asmsim.sim_wave_walk_asm.restype = C.c_int
    asmsim.sim_wave_walk_asm.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    rng = random.Random(31337)
    for trial in range(3000):
        # a window's next-position bytes: a start at p is followed by one at p + 3 .. p + gap (255 beyond position 250), and -- the
        # property the grouped hops rely on -- nothing that starts at 248 .. 255 has a successor inside the window
        gap = rng.choice([3, 4, 6, 10, 30, 80, 300])
        nxt = [255] * 256
        for p_ in range(256):
            q = p_ + rng.randrange(3, gap + 1)
            nxt[p_] = q if q <= 250 else 255
        if trial % 50 == 0:
            nxt[0] = 255                                            # a window with a single start
        if trial % 50 == 1:
            for p_ in range(0, 250, 3): nxt[p_] = p_ + 3 if p_ + 3 <= 250 else 255   # 84 starts: the lanes run out
        nx = (C.c_uint32 * 64)(*[nxt[4 * l] | (nxt[4 * l + 1] << 8) | (nxt[4 * l + 2] << 16) | (nxt[4 * l + 3] << 24) for l in range(64)])
        posv, t = (C.c_uint32 * 64)(), C.c_uint32(0)
        assert asmsim.sim_wave_walk_asm(nx, 0, posv, C.byref(t)) == 0
        want, s_ = [], 0
        while True:
            want.append(s_)
            s_ = nxt[s_]
            if s_ == 255 or len(want) == 64:
                break
        assert t.value == len(want) and list(posv[:t.value]) == want, (trial, gap, t.value, len(want), list(posv[:8]), want[:8])
