"""Parity at BASELINE.json scale, against the reference library (oracle/_ref), for every config -- the round-1 review asked for
thousands of checked blocks, not three.  All through the C ABI on device-resident batches (what bench.py times)."""
import concurrent.futures as cf
import hashlib
import json
import os
import random

import pytest

from conftest import GOLD

pytestmark = pytest.mark.gpu


def _batch(torch, dev, n, blk, cap):
    i64, i32 = torch.int64, torch.int32
    return dict(so=torch.arange(n, dtype=i64, device=dev) * blk, sl=torch.full((n,), blk, dtype=i32, device=dev),
                co=torch.arange(n, dtype=i64, device=dev) * cap, cc=torch.full((n,), cap, dtype=i32, device=dev),
                clen=torch.zeros(n, dtype=i32, device=dev), dlen=torch.zeros(n, dtype=i32, device=dev))


def _pool(fn, items, threads=32):
    with cf.ThreadPoolExecutor(threads) as ex:   # (the reference library is called through ctypes: the GIL is released in the call)
        return list(ex.map(fn, items))


def test_cfg2_65536_blocks_compressed_bytes_vs_reference(amd, ref):
    """BASELINE configs[1]: the full 65536 x 64 KiB batch; the compressed bytes of 4096 randomly chosen blocks (and the sizes of
    all of them through the round trip) against LZ4_compress_default of the reference library"""
    import torch
    dev = torch.device("cuda:0")
    n, blk = 65536, 65536
    cap = amd.maxCompressedLength(blk)
    src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
    amd.DeviceBatch.gen_blocks(src, blk, blk, n)
    comp = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    B = _batch(torch, dev, n, blk, cap)
    amd.DeviceBatch.compress_fast(src, B["so"], B["sl"], comp, B["co"], B["cc"], B["clen"])
    torch.cuda.synchronize()
    clen = B["clen"].cpu().tolist()
    assert min(clen) > 0
    rng = random.Random(2)
    pick = sorted(rng.sample(range(n), 4096))
    idx = torch.tensor(pick, device=dev)
    hs = src.view(n, blk)[idx].cpu().numpy()
    hc = comp.view(n, cap)[idx].cpu().numpy()

    def one(k):
        want = ref.compress_fast(hs[k].tobytes())
        return len(want) == clen[pick[k]] and hc[k][:len(want)].tobytes() == want
    bad = [pick[k] for k, ok in enumerate(_pool(one, range(len(pick)))) if not ok]
    assert not bad, bad[:10]
    # and every block decodes back through both decoders
    back = torch.zeros(n * blk, dtype=torch.uint8, device=dev)
    amd.DeviceBatch.decompress_safe(comp, B["co"], B["clen"], back, B["so"], B["sl"], B["dlen"])
    assert torch.equal(back, src) and torch.equal(B["dlen"], B["sl"])
    back.zero_()
    amd.DeviceBatch.decompress_fast(comp, B["co"], B["cc"], back, B["so"], B["sl"], B["dlen"])
    assert torch.equal(back, src) and torch.equal(B["dlen"], B["clen"])


def test_cfg3_reference_compressed_4MiB_blocks_every_decoder_variant(amd, ref, O):
    """BASELINE configs[2]: 64 blocks of 4 MiB compressed BY THE REFERENCE LIBRARY, decoded by every variant of the HIP decoder
    (lanes x plain / pipelined / staged), safe and fast"""
    import numpy as np
    import torch
    dev = torch.device("cuda:0")
    n, blk = 64, 4 << 20
    dsrc = torch.empty(n * blk, dtype=torch.uint8, device=dev)
    amd.DeviceBatch.gen_blocks(dsrc, blk, blk, n, first_idx=1 << 24, win=4096)
    host = dsrc.cpu().numpy()
    streams = _pool(lambda i: ref.compress_fast(host[i * blk:(i + 1) * blk].tobytes()), range(n))
    # the engine's own bytes for the same blocks are the reference's
    cap = amd.maxCompressedLength(blk)
    comp = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    B = _batch(torch, dev, n, blk, cap)
    amd.DeviceBatch.compress_fast(dsrc, B["so"], B["sl"], comp, B["co"], B["cc"], B["clen"])
    got = comp.view(n, cap).cpu().numpy()
    for i, (s, l) in enumerate(zip(streams, B["clen"].cpu().tolist())):
        assert l == len(s) and got[i][:l].tobytes() == s, i
    # reference-compressed streams, packed back to back at odd offsets
    offs, p = [], 1
    for s in streams:
        offs.append(p); p += len(s) + 3
    packed = np.zeros(p + 64, dtype=np.uint8)
    for o, s in zip(offs, streams):
        packed[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    dcomp = torch.from_numpy(packed).to(dev)
    co = torch.tensor(offs, dtype=torch.int64, device=dev)
    cl = torch.tensor([len(s) for s in streams], dtype=torch.int32, device=dev)
    back = torch.empty(n * blk, dtype=torch.uint8, device=dev)
    try:
        for lanes, pipe, stage, ring in ((0, -1, -1, 0), (4, 0, 0, 0), (4, 1, 0, 0), (8, 1, 0, 0), (16, 0, 0, 0), (16, 1, 0, 0), (32, 1, 0, 0), (64, 0, 0, 0), (64, 1, 0, 0), (4, 0, 1, 0), (8, 0, 1, 0), (16, 0, 1, 0), (64, 0, 1, 0),
                                         (4, 2, 0, 0), (8, 2, 0, 0), (16, 2, 0, 0),
                                         (4, 3, 0, 2048), (8, 3, 0, 2048), (8, 3, 0, 4096), (16, 3, 0, 4096),          # the ring loop; (4, 3, 2048) is what a routed batch of 12288..40959 such blocks gets
                                         (64, 4, 0, 0), (64, 4, 0, 8192), (64, 4, 0, 16384), (64, 4, 0, 32768),      # the wave loop: a wavefront per block
                                         (64, 5, 0, 0), (64, 5, 0, 8192), (64, 5, 0, 16384), (64, 5, 0, 32768),      # ... several sequences of the block per trip
                                         (64, 7, 0, 0), (64, 7, 0, 16384), (64, 7, 0, 32768),                        # the pair loop: two wavefronts per block
                                         (64, 8, 0, 0), (64, 8, 0, 8192), (64, 8, 0, 16384), (64, 8, 0, 32768)):          # the trio loop: three
            amd.set_option("decode_lanes", lanes); amd.set_option("decode_pipe", pipe); amd.set_option("decode_stage", stage); amd.set_option("decode_ring", ring)
            back.zero_()
            amd.DeviceBatch.decompress_safe(dcomp, co, cl, back, B["so"], B["sl"], B["dlen"])
            assert torch.equal(back, dsrc) and torch.equal(B["dlen"], B["sl"]), (lanes, pipe, stage, ring)
            back.zero_()
            amd.DeviceBatch.decompress_fast(dcomp, co, cl + 3, back, B["so"], B["sl"], B["dlen"])
            assert torch.equal(back, dsrc) and torch.equal(B["dlen"], cl), (lanes, pipe, stage, ring)
    finally:
        amd.set_option("decode_lanes", 0); amd.set_option("decode_pipe", -1); amd.set_option("decode_stage", -1); amd.set_option("decode_ring", 0)


def test_routed_batch_of_big_blocks_vs_reference(amd, ref):
    """Every decode knob at its default and a batch of 16384..40959 blocks whose compressed size averages >= 512 KiB: the launch is
    routed ON THE DEVICE (decode_route_kernel) to the ring loop with 4 lanes and a 2 KiB ring -- the instantiation behind the
    configs[2] number of the bench line.  16384 blocks of 1.25 MiB; a sample of the streams is the reference library's own bytes
    (asserted), a second sample is damaged (flipped bytes, truncated, capacity too small): sizes, bytes and error codes against
    LZ4_decompress_safe (LZ4JNI.c:216), and the untouched blocks against their source."""
    import numpy as np
    import torch
    dev = torch.device("cuda:0")
    n, blk = 16384, 1310720
    cap = amd.maxCompressedLength(blk)
    src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
    amd.DeviceBatch.gen_blocks(src, blk, blk, n, first_idx=5 << 24, win=4096)
    comp = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    B = _batch(torch, dev, n, blk, cap)
    amd.DeviceBatch.compress_fast(src, B["so"], B["sl"], comp, B["co"], B["cc"], B["clen"])
    torch.cuda.synchronize()
    clen = B["clen"].cpu().numpy().copy()
    assert clen.min() > 0 and clen.mean() >= (512 << 10), clen.mean()      # (the route's threshold)
    rng = random.Random(12288)
    sample = rng.sample(range(n), 40)
    good, bad = sample[:20], sample[20:]
    hsrc = {i: src[i * blk:(i + 1) * blk].cpu().numpy().tobytes() for i in sample}
    streams = {i: comp[i * cap:i * cap + int(clen[i])].cpu().numpy().tobytes() for i in sample}
    for i in good:
        assert streams[i] == ref.compress_fast(hsrc[i]), i                 # these streams ARE the reference library's bytes
    caps = np.full(n, blk, dtype=np.int32)
    for k, i in enumerate(bad):
        c = bytearray(streams[i])
        if k % 3 == 0:
            for _ in range(rng.randrange(1, 4)):
                c[rng.randrange(len(c))] = rng.randrange(256)
            comp[i * cap:i * cap + len(c)] = torch.from_numpy(np.frombuffer(bytes(c), dtype=np.uint8).copy()).to(dev)
        elif k % 3 == 1:
            clen[i] = rng.randrange(len(c) // 3, len(c)); c = c[:clen[i]]
        else:
            caps[i] = blk - rng.randrange(1, 5000)
        streams[i] = bytes(c)
    cl = torch.from_numpy(clen).to(dev)
    dc = torch.from_numpy(caps).to(dev)
    back = torch.zeros(n * blk, dtype=torch.uint8, device=dev)
    amd.DeviceBatch.decompress_safe(comp, B["co"], cl, back, B["so"], dc, B["dlen"])
    torch.cuda.synchronize()
    assert amd.last_decode_route()[0] == 1, amd.last_decode_route()       # the ring loop it was
    dlen = B["dlen"].cpu().numpy()
    for i in bad:
        er, ed = ref.decompress_safe_raw(streams[i], int(caps[i]))
        assert int(dlen[i]) == er, (i, int(dlen[i]), er)
        if er >= 0:
            assert back[i * blk:i * blk + er].cpu().numpy().tobytes() == ed[:er], i
    ok = np.ones(n, dtype=bool); ok[bad] = False
    assert (dlen[ok] == blk).all()
    okt = torch.from_numpy(ok).to(dev)
    assert torch.equal(back.view(n, blk)[okt], src.view(n, blk)[okt])


def test_batches_routed_by_sequence_density_vs_reference(amd, ref, corpus):
    """More than 16 blocks per CU and every knob at its default: decode_route_kernel samples the MIDDLE of 32 streams (the wave loop's
    speculative walk from an arbitrary byte: it falls in with the true token chain within a few sequences) and sends batches of SHORT
    sequences with near sources -- text: ~6 output bytes per sequence, most offsets within 6 KB -- to the wave kernel, everything else
    (App. F 34 bytes per sequence, a bitmap 100+, geophysical data 40; synthetic streams as dense as text of 13 bytes per sequence) to a
    lane-group loop: the deep loop below 40960 blocks and, from there on, for near sources (the bitmap), else the 4-lane staged loop.  Batches of
    up to 32 blocks per CU (two rounds of the wave kernel's 16 wavefronts per CU) go to the wave kernel whatever they hold, unless their streams are
    mostly literals (the geophysical data).  For text, App. F, bitmap and geophysical batches below and above 40960 blocks: the route
    taken, sizes and bytes against the source, a sample of damaged / reference-compressed streams against LZ4_decompress_safe
    (LZ4JNI.c:216); and the same bytes with the wave route opened wide (decode_route_short 255) and closed (0)."""
    import numpy as np
    import torch
    dev = torch.device("cuda:0")
    blk = 65536
    cap = amd.maxCompressedLength(blk)
    book = np.frombuffer(corpus["book1[:200000]"], dtype=np.uint8)
    rng = random.Random(4406)

    def make(kind, n):
        src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
        if kind == "appf":
            amd.DeviceBatch.gen_blocks(src, blk, blk, n, first_idx=9 << 24)
        elif kind == "lit2":      # as dense as text (one or two literals per match) with match offsets anywhere in 64 KB: NOT for the wave kernel's 8 KB ring
            amd.DeviceBatch.gen_blocks(src, blk, blk, n, first_idx=9 << 24, litmax=2)
        elif kind == "book":
            bdev = torch.from_numpy(book.copy()).to(dev)
            offs = torch.arange(n, dtype=torch.int64, device=dev) * 7919 % (len(book) - blk)
            ar = torch.arange(blk, dtype=torch.int64, device=dev)
            for c0 in range(0, n, 1024):
                c1 = min(n, c0 + 1024)
                src[c0 * blk:c1 * blk] = bdev[(offs[c0:c1, None] + ar[None, :]).reshape(-1)]
        else:
            b = np.frombuffer(corpus[kind + "[:65536]"], dtype=np.uint8)
            src = torch.from_numpy(b.copy()).to(dev).repeat(n)
        return src

    try:
        # (up to 32 blocks per CU -- 8192 -- the wave kernel takes every kind of data but streams that are mostly literals: geo, ratio 1.07)
        for kind, n, want_route in (("book", 6144, 2), ("appf", 6144, 2), ("lit2", 6144, 2), ("pic", 5000, 2), ("geo", 5000, 0), ("appf", 10240, 0), ("book", 45056, 2), ("appf", 45056, 0), ("pic", 45056, 3), ("book", 20000, 2)):
            src = make(kind, n)
            comp = torch.zeros(n * cap, dtype=torch.uint8, device=dev)   # (zeros behind every stream: what the fast decoder's sampler may look at is defined)
            B = _batch(torch, dev, n, blk, cap)
            amd.DeviceBatch.compress_fast(src, B["so"], B["sl"], comp, B["co"], B["cc"], B["clen"])
            torch.cuda.synchronize()
            clen = B["clen"].cpu().numpy().copy()
            # the FAST decoder is given the slots' capacity, not the streams' lengths (LZ4_decompress_fast, LZ4JNI.c:169): the sampler must not
            # take the middle of THAT for the middle of the stream (it did: the headline's decompress_fast went to the wave kernel, 710 -> 442 GB/s)
            back = torch.zeros(n * blk, dtype=torch.uint8, device=dev)
            amd.DeviceBatch.decompress_fast(comp, B["co"], B["cc"], back, B["so"], B["sl"], B["dlen"])
            torch.cuda.synchronize()
            # (a fifth of the output size into the slot; the bitmap's streams -- ratio 14 -- end before that: zeros, offset 0, nothing is routed)
            # (geo: a fifth of the output size into ITS streams lies in the file's header region, which is not mostly literals -- either route)
            assert amd.last_decode_route()[0] in ((0,) if kind == "pic" else (0, 2) if kind == "geo" else (want_route,)), (kind, n, "fast", amd.last_decode_route())
            assert torch.equal(back, src) and torch.equal(B["dlen"], B["clen"]), (kind, n, "fast")
            del back
            sample = rng.sample(range(n), 24)
            good, bad = sample[:8], sample[8:]
            streams = {i: comp[i * cap:i * cap + int(clen[i])].cpu().numpy().tobytes() for i in sample}
            for i in good:
                assert streams[i] == ref.compress_fast(src[i * blk:(i + 1) * blk].cpu().numpy().tobytes()), (kind, i)   # the reference library's own bytes
            caps = np.full(n, blk, dtype=np.int32)
            for k, i in enumerate(bad):
                c = bytearray(streams[i])
                if k % 3 == 0:
                    for _ in range(rng.randrange(1, 4)):
                        c[rng.randrange(len(c))] = rng.randrange(256)
                    comp[i * cap:i * cap + len(c)] = torch.from_numpy(np.frombuffer(bytes(c), dtype=np.uint8).copy()).to(dev)
                elif k % 3 == 1:
                    clen[i] = rng.randrange(len(c) // 3, len(c)); c = c[:clen[i]]
                else:
                    caps[i] = blk - rng.randrange(1, 5000)
                streams[i] = bytes(c)
            cl = torch.from_numpy(clen).to(dev)
            dc = torch.from_numpy(caps).to(dev)
            want = {i: ref.decompress_safe_raw(streams[i], int(caps[i])) for i in bad}
            ok = np.ones(n, dtype=bool); ok[bad] = False
            okt = torch.from_numpy(ok).to(dev)
            for short in (-1, 255, 0):                        # the default (8 bytes per sequence), "every sequence is short", "none is"
                if short >= 0:
                    amd.set_option("decode_route_short", short)
                back = torch.zeros(n * blk, dtype=torch.uint8, device=dev)
                amd.DeviceBatch.decompress_safe(comp, B["co"], cl, back, B["so"], dc, B["dlen"])
                torch.cuda.synchronize()
                route = amd.last_decode_route()
                near = 2 * route[4] >= route[5]
                expect = want_route if short < 0 else (2 if (near or (n <= 8192 and kind != "geo")) else 0) if short == 255 else (3 if near and n >= 40960 else 0)
                assert route[0] == expect, (kind, n, short, route)
                dense = short
                dlen = B["dlen"].cpu().numpy()
                for i in bad:
                    er, ed = want[i]
                    assert int(dlen[i]) == er, (kind, n, dense, i, int(dlen[i]), er)
                    if er >= 0:
                        assert back[i * blk:i * blk + er].cpu().numpy().tobytes() == ed[:er], (kind, n, dense, i)
                assert (dlen[ok] == blk).all()
                assert torch.equal(back.view(n, blk)[okt], src.view(n, blk)[okt]), (kind, n, dense)
                del back
            amd.set_option("decode_route_short", 8)
            del src, comp
            torch.cuda.empty_cache()
    finally:
        amd.set_option("decode_route_short", 8)


def test_cfg4_256_blocks_hc9_every_block_vs_reference(amd, ref):
    """BASELINE configs[3] shape: 256 x 1 MiB blocks at HC level 9, EVERY block's bytes against LZ4_compress_HC of the reference"""
    import torch
    dev = torch.device("cuda:0")
    n, blk = 256, 1 << 20
    cap = amd.maxCompressedLength(blk)
    src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
    amd.DeviceBatch.gen_blocks(src, blk, blk, n, first_idx=2 << 24, win=4096)
    comp = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    B = _batch(torch, dev, n, blk, cap)
    amd.DeviceBatch.compress_hc(src, B["so"], B["sl"], comp, B["co"], B["cc"], B["clen"], 9)
    torch.cuda.synchronize()
    host = src.cpu().numpy()
    got = comp.view(n, cap).cpu().numpy()
    clen = B["clen"].cpu().tolist()

    def one(i):
        want = ref.compress_hc(host[i * blk:(i + 1) * blk].tobytes(), 9)
        return len(want) == clen[i] and got[i][:len(want)].tobytes() == want
    bad = [i for i, ok in enumerate(_pool(one, range(n), threads=64)) if not ok]
    assert not bad, bad[:10]


def test_cfg5_65536_hashes_vs_reference(amd, ref):
    """BASELINE configs[4] shape: XXH32 and XXH64 of 65536 x 4 KiB buffers, EVERY hash against the reference library, two seeds"""
    import torch
    dev = torch.device("cuda:0")
    n, blk = 65536, 4096
    src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
    amd.DeviceBatch.gen_blocks(src, 65536, 65536, n * blk // 65536, first_idx=3 << 24)
    off = torch.arange(n, dtype=torch.int64, device=dev) * blk
    ln = torch.full((n,), blk, dtype=torch.int32, device=dev)
    host = src.cpu().numpy().tobytes()
    for seed in (0, 0x9747b28c):
        h32 = torch.zeros(n, dtype=torch.int32, device=dev)
        h64 = torch.zeros(n, dtype=torch.int64, device=dev)
        amd.DeviceBatch.xxh32(src, off, ln, seed, h32)
        amd.DeviceBatch.xxh64(src, off, ln, seed, h64)
        a32, a64 = h32.cpu().tolist(), h64.cpu().tolist()

        def chunk(c):
            return all((a32[i] & 0xFFFFFFFF) == ref.xxh32(host[i * blk:(i + 1) * blk], seed) and
                       (a64[i] & 0xFFFFFFFFFFFFFFFF) == ref.xxh64(host[i * blk:(i + 1) * blk], seed) for i in range(c, c + 1024))
        assert all(_pool(chunk, range(0, n, 1024)))


def test_fast_decoder_contract_pinned(amd):
    """include/lz4hip.h "fast decoder contract": return code and output of lz4hip_decompress_fast on 600 valid / truncated /
    corrupted / random streams equal the committed vectors (tests/golden/fast_decode_contract.json; the valid ones were checked
    against the reference library when the file was generated)"""
    cases = json.load(open(os.path.join(GOLD, "fast_decode_contract.json")))["cases"]
    src, so, sl, do, dl, p, q = bytearray(), [], [], [], [], 0, 0
    for e in cases:
        c = bytes.fromhex(e["hex"])
        c = c[:e["src_cap"]] + bytes(max(0, e["src_cap"] - len(c)))
        so.append(p); sl.append(e["src_cap"]); src += c + b"\x77"; p += len(c) + 1
        do.append(q); dl.append(e["dst_len"]); q += e["dst_len"] + 1
    dst = bytearray(b"\xA5" * (q + 1))
    out = amd.LZ4HIPBatch.decompressFast(bytes(src), so, sl, dst, do, dl)
    for e, r, o in zip(cases, out, do):
        assert r == e["ret"], (e["hex"][:40], e["src_cap"], e["dst_len"], r, e["ret"])
        if r >= 0:
            assert hashlib.sha256(bytes(dst[o:o + e["dst_len"]])).hexdigest() == e["sha256"]
        assert dst[o + e["dst_len"]] == 0xA5   # nothing past the slot


def test_mixed_64k_blocks_every_core_vs_reference(amd, ref, O, corpus):
    """4000 blocks of 20 KiB .. 65546 bytes, each a patchwork of text, image, geophysical data, random bytes, zero runs, short
    repeated patterns and App. F pieces (every kind of step the finder meets: short and long matches, matches over 65 and 256
    bytes, dense and sparse hits, long literal runs), compressed in ONE batch by each core -- 3 = the lean finder (the hand-scheduled
    loop) forced on ALL of them incl. the text-like ones, 1 = the window-parallel core, 5 = the adaptive pairing -- and EVERY
    block's bytes compared with LZ4_compress_default of the reference library"""
    import numpy as np
    import torch
    dev = torch.device("cuda:0")
    rng = random.Random(1234)
    book, pic, geo = corpus["book1[:200000]"], corpus["pic[:65536]"], corpus["geo[:65536]"]
    syn = [O.gen_block(65536, 900 + i, litmax=lm, win=w) for i, (lm, w) in enumerate([(38, 65535), (4, 300), (200, 65535), (38, 64), (12, 4096)])]

    def piece(k):
        t = rng.randrange(8)
        if t == 0: o = rng.randrange(len(book) - k); return book[o:o + k]
        if t == 1: o = rng.randrange(len(pic) - min(k, 60000)); return pic[o:o + k]
        if t == 2: o = rng.randrange(len(geo) - min(k, 60000)); return geo[o:o + k]
        if t == 3: return rng.randbytes(k)
        if t == 4: return bytes(k)
        if t == 5: p = rng.randbytes(rng.randrange(1, 40)); return (p * (k // len(p) + 1))[:k]
        if t == 6: s = syn[rng.randrange(len(syn))]; o = rng.randrange(len(s) - min(k, 60000)); return s[o:o + k]
        return bytes(rng.randrange(3) for _ in range(min(k, 3000)))
    blocks = []
    for _ in range(4000):
        n = rng.choice([65536, 65536, 65546, rng.randrange(20000, 65536)])
        b = bytearray()
        while len(b) < n:
            b += piece(rng.choice([50, 300, 2000, 9000, 30000]))
        blocks.append(bytes(b[:n]))
    want = _pool(ref.compress_fast, blocks)
    slot = 65546
    cap = amd.maxCompressedLength(slot)
    n = len(blocks)
    host = np.zeros(n * slot, dtype=np.uint8)
    for i, b in enumerate(blocks):
        host[i * slot:i * slot + len(b)] = np.frombuffer(b, dtype=np.uint8)
    src = torch.from_numpy(host).to(dev)
    B = _batch(torch, dev, n, slot, cap)
    B["sl"] = torch.tensor([len(b) for b in blocks], dtype=torch.int32, device=dev)
    comp = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    for core in (3, 1, 5):
        amd.set_option("compress_core", core)
        try:
            comp.zero_()
            amd.DeviceBatch.compress_fast(src, B["so"], B["sl"], comp, B["co"], B["cc"], B["clen"])
            torch.cuda.synchronize()
        finally:
            amd.set_option("compress_core", 5)
        clen = B["clen"].cpu().tolist()
        hc = comp.view(n, cap).cpu().numpy()
        bad = [i for i in range(n) if clen[i] != len(want[i]) or hc[i][:clen[i]].tobytes() != want[i]]
        assert not bad, (core, bad[:10], [len(blocks[i]) for i in bad[:10]])


def test_host_batch_many_chunks_ragged_concurrent_callers(amd, ref, O):
    """The host-pointer batch API over SEVERAL staging chunks (round 3: pack | GPU | finisher threads on rotating buffer sets,
    csrc/api.cpp host_shard): ~330 MB of ragged blocks (0 .. 64 KiB, some incompressible, some with capacities one byte short)
    from two concurrent callers; every size and a sample of the compressed bytes against the reference library, every block back
    through the safe decoder's host path."""
    import numpy as np
    rng = random.Random(5)
    base = [O.gen_block(65536, s) for s in range(24)] + [rng.randbytes(65536) for _ in range(4)] + [bytes(65536)]
    n = 5200
    lens = [rng.choice([65536, 65536, 65536, rng.randrange(0, 65537), rng.randrange(13, 2000)]) for _ in range(n)]
    srcs = [base[i % len(base)][:ln] for i, ln in enumerate(lens)]
    src = b"".join(srcs)
    so = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    want = _pool(lambda v: len(ref.compress_fast(v)), srcs)
    caps = [amd.maxCompressedLength(ln) if i % 7 else max(0, want[i] - 1) for i, ln in enumerate(lens)]   # every 7th: one byte short -> 0
    do = np.concatenate([[0], np.cumsum(caps)[:-1]]).astype(np.uint64)

    def caller(_):
        dst = bytearray(int(sum(caps)) + 1)
        out = amd.LZ4HIPBatch.compress(src, so, np.array(lens, dtype=np.int32), dst, do, np.array(caps, dtype=np.int32))
        return dst, out
    with cf.ThreadPoolExecutor(2) as ex:
        results = list(ex.map(caller, range(2)))
    for dst, out in results:
        for i in range(n):
            exp = want[i] if caps[i] >= want[i] else 0
            assert out[i] == exp, (i, lens[i], caps[i], int(out[i]), exp)
        for i in rng.sample(range(n), 300):
            if out[i] > 0:
                assert bytes(dst[int(do[i]):int(do[i]) + int(out[i])]) == ref.compress_fast(srcs[i]), i
    dst, out = results[0]
    ok = [i for i in range(n) if out[i] > 0]
    back = bytearray(len(src) + 1)
    got = amd.LZ4HIPBatch.decompressSafe(dst, do[ok], np.array([out[i] for i in ok], dtype=np.int32), back, so[ok],
                                     np.array([lens[i] for i in ok], dtype=np.int32))
    assert list(got) == [lens[i] for i in ok]
    for i in ok:
        assert bytes(back[int(so[i]):int(so[i]) + lens[i]]) == srcs[i], i


def test_batches_that_are_not_a_multiple_of_a_workgroup_per_cu_vs_reference(amd, ref):
    """The trio and wave kernels give a workgroup W blocks; a batch that is not a multiple of W x CUs is SPREAD -- wavefront k of workgroup i takes
    block i + k * grid, every CU gets ceil(n / CUs) blocks (kernels.hip wave_spread) -- where that puts fewer blocks on a CU than packing W neighbours
    does.  Batch sizes on both sides of every W (trio 1 / 2 / 4 / 5, wave 8 / 16) with every knob at its default, safe and fast decoders: sizes and
    bytes against the source, a sample of streams against the reference's own bytes, a damaged sample against LZ4_decompress_safe (LZ4JNI.c:216)."""
    import numpy as np
    import torch
    dev = torch.device("cuda:0")
    blk = 65536
    cap = amd.maxCompressedLength(blk)
    rng = random.Random(4711)
    for n in (255, 257, 300, 511, 700, 1023, 1100, 1279, 1300, 1800, 2047, 2300, 3000, 4095):
        src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
        amd.DeviceBatch.gen_blocks(src, blk, blk, n, first_idx=(11 << 24) + n)
        comp = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
        B = _batch(torch, dev, n, blk, cap)
        amd.DeviceBatch.compress_fast(src, B["so"], B["sl"], comp, B["co"], B["cc"], B["clen"])
        torch.cuda.synchronize()
        clen = B["clen"].cpu().numpy().copy()
        back = torch.zeros(n * blk, dtype=torch.uint8, device=dev)
        amd.DeviceBatch.decompress_fast(comp, B["co"], B["cc"], back, B["so"], B["sl"], B["dlen"])
        torch.cuda.synchronize()
        assert torch.equal(back, src) and torch.equal(B["dlen"], B["clen"]), (n, "fast")
        sample = rng.sample(range(n), 12)
        good, bad = sample[:4], sample[4:]
        streams = {i: comp[i * cap:i * cap + int(clen[i])].cpu().numpy().tobytes() for i in sample}
        for i in good:
            assert streams[i] == ref.compress_fast(src[i * blk:(i + 1) * blk].cpu().numpy().tobytes()), (n, i)
        caps = np.full(n, blk, dtype=np.int32)
        for k, i in enumerate(bad):
            c = bytearray(streams[i])
            if k % 2 == 0:
                c[rng.randrange(len(c))] ^= 1 + rng.randrange(255)
                comp[i * cap:i * cap + len(c)] = torch.from_numpy(np.frombuffer(bytes(c), dtype=np.uint8).copy()).to(dev)
            else:
                caps[i] = blk - rng.randrange(1, 3000)
            streams[i] = bytes(c)
        want = {i: ref.decompress_safe_raw(streams[i], int(caps[i])) for i in bad}
        ok = np.ones(n, dtype=bool); ok[bad] = False
        okt = torch.from_numpy(ok).to(dev)
        back = torch.zeros(n * blk, dtype=torch.uint8, device=dev)
        amd.DeviceBatch.decompress_safe(comp, B["co"], torch.from_numpy(clen).to(dev), back, B["so"], torch.from_numpy(caps).to(dev), B["dlen"])
        torch.cuda.synchronize()
        dlen = B["dlen"].cpu().numpy()
        for i in bad:
            er, ed = want[i]
            assert int(dlen[i]) == er, (n, i, int(dlen[i]), er)
            if er >= 0:
                assert back[i * blk:i * blk + er].cpu().numpy().tobytes() == ed[:er], (n, i)
        assert (dlen[ok] == blk).all(), n
        assert torch.equal(back.view(n, blk)[okt], src.view(n, blk)[okt]), n
        del src, comp, back
