"""Shared cases for the stream/container twins (lz4-java_amd/streams.py).  The same case functions run with the
oracle-backed engine on CPU (tests/test_streams_host.py: container logic only) and with the GPU engine
(tests/test_gpu_streams.py: the product path).  TEST INFRASTRUCTURE: OracleEngine wraps oracle/ and exists only so
the container logic can be checked without a GPU."""
import io
import os
import shutil
import struct

import pytest
import subprocess
import tempfile

LZ4_CLI = shutil.which("lz4") or ("/opt/conda/bin/lz4" if os.path.exists("/opt/conda/bin/lz4") else None)


class OracleEngine:
    """same interface as streams.HIPEngine, served by the CPU oracle (tests only)"""

    def __init__(self, port, O, hcLevel=None):
        self.port, self.O, self.hcLevel = port, O, hcLevel

    def compress(self, src, srcOff, srcLen, dst, dstOff, dstCap):
        out = []
        for so, sl, do, dc in zip(srcOff, srcLen, dstOff, dstCap):
            blk = bytes(src[so:so + sl])
            r, b = (self.port.compress_fast_raw(blk, dc) if self.hcLevel is None
                    else self.port.compress_hc_raw(blk, self.hcLevel, dc))
            if r > 0:
                dst[do:do + r] = b[:r]
            out.append(r)
        return out

    def decompressSafe(self, src, srcOff, srcLen, dst, dstOff, dstCap):
        out = []
        for so, sl, do, dc in zip(srcOff, srcLen, dstOff, dstCap):
            r, b = self.port.decompress_safe_raw(bytes(src[so:so + sl]), dc)
            if r > 0:
                dst[do:do + r] = b[:r]
            out.append(r)
        return out

    def decompressFast(self, src, srcOff, srcCap, dst, dstOff, dstLen):
        out = []
        for so, sc, do, dl in zip(srcOff, srcCap, dstOff, dstLen):
            r, b = self.O.decompress_fast_bounded(bytes(src[so:so + sc]), sc, dl)
            if r >= 0:
                dst[do:do + dl] = b[:dl]
            out.append(r)
        return out

    def xxh32(self, buf, off, length, seed=0):
        return [self.port.xxh32(bytes(buf[o:o + n]), seed) for o, n in zip(off, length)]

    def newStreamingHash32(self, seed):
        port = self.port

        class H:  # the oracle has no streaming state: hash the concatenation
            def __init__(self):
                self.data = bytearray()

            def update(self, buf, off, length):
                self.data += bytes(buf[off:off + length])

            def getValue(self):
                return port.xxh32(bytes(self.data), seed)

            def reset(self):
                self.data = bytearray()

            def close(self):
                pass
        return H()


class OracleDeviceEngine(OracleEngine):
    """OracleEngine + `containerDecode`: a host restatement of the device read path (kernels.hip container_walk_kernel / _raw_kernel /
    _verdict_kernel: the walk's rules in their order, the first failing block in the readers' order of checks), so that the CPU suite
    runs the readers' DEVICE-PATH logic (chunking, hand-back of unconsumed bytes, stop reasons -> exceptions) that
    tests/test_gpu_streams.py runs on the GPU.  Test infrastructure only."""
    CR_END, CR_MORE, CR_TRUNCATED, CR_BLOCK_TOO_BIG, CR_BLOCK_CHECKSUM, CR_DECODE, CR_CORRUPT, CR_SLOTS = range(8)

    def __init__(self, port, O, hcLevel=None):
        super().__init__(port, O, hcLevel)
        self.calls = 0

    def containerDecode(self, kind, body, maxBlock, nMax, blockChecksum=False):
        self.calls += 1
        body = bytes(body)
        n, p, why, blocks = len(body), 0, self.CR_SLOTS, []
        u32 = lambda o: struct.unpack_from("<I", body, o)[0]
        i32 = lambda o: struct.unpack_from("<i", body, o)[0]
        while len(blocks) < nMax:
            if p == n:
                why = self.CR_MORE; break
            if kind == 0:
                if p + 4 > n:
                    why = self.CR_TRUNCATED; break
                word = u32(p); size = word & 0x7FFFFFFF
                if size == 0:
                    p += 4; why = self.CR_END; break
                if size > maxBlock:
                    why = self.CR_BLOCK_TOO_BIG; break
                need = 4 + size + (4 if blockChecksum else 0)
                if p + need > n:
                    why = self.CR_TRUNCATED; break
                blocks.append(dict(raw=bool(word & 0x80000000), pay=body[p + 4:p + 4 + size], cap=maxBlock,
                                   stored=u32(p + 4 + size) if blockChecksum else 0, end=p + need))
                p += need
            else:
                if p + 21 > n:
                    why = self.CR_TRUNCATED; break
                token = body[p + 8]; method = token & 0xF0; level = 10 + (token & 0x0F)
                clen, olen, check = i32(p + 9), i32(p + 13), u32(p + 17)
                bad = body[p:p + 8] != b"LZ4Block" or method not in (0x10, 0x20) or olen > (1 << level) or olen < 0 or clen < 0 or \
                    (olen == 0 and clen != 0) or (olen != 0 and clen == 0) or (method == 0x10 and olen != clen)
                if bad:
                    why = self.CR_CORRUPT; break
                if olen == 0:
                    if check != 0:
                        why = self.CR_CORRUPT; break
                    p += 21; why = self.CR_END; break
                if olen > maxBlock:
                    why = self.CR_BLOCK_TOO_BIG; break
                if p + 21 + clen > n:
                    why = self.CR_TRUNCATED; break
                if method == 0x20 and olen > 255 * clen + 64:   # (cannot decode: clen bytes of LZ4 are at most 255 x clen bytes; the device says so without sizing a slot by it)
                    why = self.CR_CORRUPT; break
                blocks.append(dict(raw=method == 0x10, pay=body[p + 21:p + 21 + clen], cap=olen, stored=check, end=p + 21 + clen))
                p += 21 + clen
        if len(blocks) == nMax and p == n:
            why = self.CR_MORE
        out, sizes, code = bytearray(), [], 0
        for k, b in enumerate(blocks):
            if kind == 0:
                if blockChecksum and self.port.xxh32(b["pay"], 0) != b["stored"]:
                    return bytes(out), sizes, (blocks[k - 1]["end"] if k else 0), self.CR_BLOCK_CHECKSUM, 0
                if b["raw"]:
                    dec = b["pay"]
                else:
                    r, d = self.port.decompress_safe_raw(b["pay"], b["cap"])
                    if r < 0:
                        return bytes(out), sizes, (blocks[k - 1]["end"] if k else 0), self.CR_DECODE, r
                    dec = d[:r]
            else:
                if b["raw"]:
                    dec = b["pay"]
                else:
                    r, d = self.O.decompress_fast_bounded(b["pay"], len(b["pay"]), b["cap"])
                    if r != len(b["pay"]):
                        return bytes(out), sizes, (blocks[k - 1]["end"] if k else 0), self.CR_CORRUPT, 0
                    dec = d[:b["cap"]]
                if (self.port.xxh32(dec, 0x9747b28c) & 0x0FFFFFFF) != b["stored"]:
                    return bytes(out), sizes, (blocks[k - 1]["end"] if k else 0), self.CR_CORRUPT, 0
            out += dec; sizes.append(len(dec))
        return bytes(out), sizes, p, why, code


class _NoSeek:
    """a pipe-like input: read() only, and it remembers how much was taken"""

    def __init__(self, b):
        self.b, self.pos = bytes(b), 0

    def read(self, n=-1):
        n = len(self.b) - self.pos if n is None or n < 0 else n
        c = self.b[self.pos:self.pos + n]
        self.pos += len(c)
        return c

    def seekable(self):
        return False


def case_device_read_path_advisor_findings(S, engine, data):
    """round-4 advisor: (1) an LZ4Block header in the middle of a long stream whose compressedLen exceeds the reader's chunk made the
    device read path spin (no progress, no exception); (2) with readSingleFrame / stopOnEmptyBlock the device path left `inp` far
    behind the container's end.  `engine` has containerDecode (the GPU engine, or OracleDeviceEngine on the CPU suite)."""
    B = S.FLG.Bits
    # (1) LZ4Block: blocks of 1 KiB, tiny batches, a compressedLen of 0x7FFFFFF0 / one flipped top bit in block 5's header
    d = data[:40000]
    bs = block_stream_bytes(S, d, engine, 1024, chunk=7000, batchBlocks=4)
    heads = [i for i in range(len(bs) - 8) if bs[i:i + 8] == b"LZ4Block"]
    for newlen in (0x7FFFFFF0, None):
        bad = bytearray(bs)
        h = heads[5]
        if newlen is None:
            bad[h + 12] ^= 0x40          # top byte of compressedLen: + 1 GiB
        else:
            bad[h + 9:h + 13] = struct.pack("<i", newlen)
        rd = S.LZ4BlockInputStream(io.BytesIO(bytes(bad) + bytes(300000)), engine=engine, batchBlocks=4)
        got = bytearray()
        with pytest.raises((S.IOException, EOFError)):
            for _ in range(10000):        # (bounded: the defect was an endless loop)
                c = rd.read(1 << 16)
                if not c:
                    break
                got += c
            else:
                raise AssertionError("the reader makes no progress")
        assert bytes(got) == d[:len(got)] and len(got) >= 4 * 1024
    # (2a) a single frame followed by other bytes: the input is left right behind the frame
    fr = frame_bytes(S, data[:150000], engine, S.BLOCKSIZE.SIZE_64KB, (B.BLOCK_INDEPENDENCE, B.CONTENT_CHECKSUM))
    tail = b"TRAILER-" * 5000
    inp = io.BytesIO(fr + tail)
    one = S.LZ4FrameInputStream(inp, readSingleFrame=True, engine=engine, batchBlocks=8)
    assert one.read() == data[:150000]
    assert inp.read() == tail
    pipe = _NoSeek(fr + tail)             # not seekable: the host walk, which takes exactly the frame
    one = S.LZ4FrameInputStream(pipe, readSingleFrame=True, engine=engine, batchBlocks=8)
    assert one.read() == data[:150000]
    assert pipe.read() == tail
    # (2b) an LZ4Block stream that stops at its empty block (the default), embedded in another protocol
    bs2 = block_stream_bytes(S, data[:90000], engine, 1 << 16)
    inp = io.BytesIO(bs2 + tail)
    rd = S.LZ4BlockInputStream(inp, engine=engine)
    assert rd.read() == data[:90000]
    assert inp.read() == tail
    pipe = _NoSeek(bs2 + tail)
    rd = S.LZ4BlockInputStream(pipe, engine=engine)
    assert rd.read() == data[:90000]
    assert pipe.read() == tail


def payload(corpus, O):
    """~1.1 MB mixed: text, synthetic LZ blocks, an incompressible stretch, zeros, a ragged tail"""
    import random
    rng = random.Random(77)
    noise = bytes(rng.getrandbits(8) for _ in range(70000))
    return (corpus["book1[:200000]"] + O.gen_block(300000, 5) + noise + bytes(200000) + corpus["geo[:65536]"]
            + corpus["pic[:65536]"] + corpus["book1[:200000]"][:12345])


def frame_bytes(S, data, engine, blockSize, bits, knownSize=-1, chunk=100000, batchBlocks=3, flush_at=None):
    sink = io.BytesIO()
    f = S.LZ4FrameOutputStream(sink, blockSize, knownSize, *bits, engine=engine, batchBlocks=batchBlocks)
    for i in range(0, len(data), chunk):
        f.write(data[i:i + chunk])
        if flush_at is not None and i // chunk == flush_at:
            f.flush()
    f.close()
    return sink.getvalue()


def parse_frame(buf):
    """independent structural walk of one frame -> (flg, bd, content_size, [(stored_raw, payload, checksum)], content_checksum, end)"""
    assert struct.unpack_from("<I", buf, 0)[0] == 0x184D2204
    flg, bd = buf[4], buf[5]
    p = 6
    csize = None
    if flg & 8:
        csize = struct.unpack_from("<Q", buf, p)[0]
        p += 8
    hc = buf[p]
    p += 1
    blocks = []
    while True:
        w = struct.unpack_from("<I", buf, p)[0]
        p += 4
        if w == 0:
            break
        n = w & 0x7FFFFFFF
        body = buf[p:p + n]
        p += n
        ck = None
        if flg & 16:
            ck = struct.unpack_from("<I", buf, p)[0]
            p += 4
        blocks.append((bool(w >> 31), body, ck))
    cck = None
    if flg & 4:
        cck = struct.unpack_from("<I", buf, p)[0]
        p += 4
    return flg, bd, csize, hc, blocks, cck, p


def cli(args, data):
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "in"), os.path.join(d, "out")
        open(a, "wb").write(data)
        subprocess.run([LZ4_CLI, "-q", "-f"] + args + [a, b], check=True)
        return open(b, "rb").read()


# ---------------------------------------------------------------------------------------------------
# cases (engine-agnostic)
# ---------------------------------------------------------------------------------------------------
def case_frame_layout_and_roundtrip(S, engine, port, data):
    B = S.FLG.Bits
    for blockSize, bits in ((S.BLOCKSIZE.SIZE_64KB, (B.BLOCK_INDEPENDENCE,)),
                            (S.BLOCKSIZE.SIZE_64KB, (B.BLOCK_INDEPENDENCE, B.BLOCK_CHECKSUM, B.CONTENT_CHECKSUM, B.CONTENT_SIZE)),
                            (S.BLOCKSIZE.SIZE_256KB, (B.BLOCK_INDEPENDENCE, B.CONTENT_CHECKSUM)),
                            (S.BLOCKSIZE.SIZE_1MB, (B.BLOCK_INDEPENDENCE, B.BLOCK_CHECKSUM)),
                            (S.BLOCKSIZE.SIZE_4MB, (B.BLOCK_INDEPENDENCE, B.CONTENT_SIZE))):
        known = len(data) if B.CONTENT_SIZE in bits else -1
        fr = frame_bytes(S, data, engine, blockSize, bits, known)
        flg, bd, csize, hc, blocks, cck, end = parse_frame(fr)
        assert end == len(fr)
        assert flg == (1 << 6) | sum(1 << b for b in bits) and bd == blockSize << 4
        desc = fr[4:6] + (struct.pack("<Q", csize) if csize is not None else b"")
        assert hc == (port.xxh32(desc, 0) >> 8) & 0xFF
        assert csize == (len(data) if known >= 0 else None)
        mbs = 1 << (2 * blockSize + 8)
        assert len(blocks) == -(-len(data) // mbs)
        for i, (stored_raw, body, ck) in enumerate(blocks):
            raw = data[i * mbs:(i + 1) * mbs]
            comp = port.compress_fast(raw)
            if len(comp) >= len(raw):            # LZ4FrameOutputStream.java:216-224
                assert stored_raw and body == raw
            else:
                assert not stored_raw and body == comp   # bit-exact liblz4 block inside the container
            if ck is not None:
                assert ck == port.xxh32(body, 0)
        if cck is not None:
            assert cck == port.xxh32(data, 0)
        for bb in (1, 5):
            got = S.LZ4FrameInputStream(io.BytesIO(fr), engine=engine, batchBlocks=bb).read()
            assert got == data


def case_frame_size_sweep(S, engine, data):
    """LZ4FrameIOStreamTest.java:72-90: sizes 0, 1, 1K, 1K+1, 64K, 128K, 1M (+ two ragged ones) x every optional-field
    combination round-trip; the lz4 CLI, when there, decodes every frame and its own output is read back"""
    B = S.FLG.Bits
    combos = [tuple(b for b, on in zip((B.BLOCK_CHECKSUM, B.CONTENT_CHECKSUM, B.CONTENT_SIZE), (k & 1, k & 2, k & 4)) if on) for k in range(8)]
    for size in (0, 1, 1024, 1025, 65536, 131072, 1 << 20, 77777, 300001):
        d = (data * (size // len(data) + 1))[:size]
        for k, extra in enumerate(combos):
            bits = (B.BLOCK_INDEPENDENCE,) + extra
            known = size if B.CONTENT_SIZE in bits else -1
            fr = frame_bytes(S, d, engine, S.BLOCKSIZE.SIZE_64KB if k % 2 else S.BLOCKSIZE.SIZE_256KB, bits, known, chunk=40000, batchBlocks=4)
            rd = S.LZ4FrameInputStream(io.BytesIO(fr), engine=engine, batchBlocks=3)
            assert rd.read() == d, (size, k)
            if LZ4_CLI and k in (0, 7) and size in (0, 1, 1025, 131072):
                assert cli(["-d"], fr) == d, (size, k)
        if LZ4_CLI and size in (0, 1, 1025, 131072):
            assert S.LZ4FrameInputStream(io.BytesIO(cli(["-1"], d)), engine=engine).read() == d, size


def case_frame_known_header_bytes(S, engine):
    # known descriptor checksums: FLG 0x60 / BD 0x70 -> HC 0x73 (what `lz4 --no-frame-crc` writes), BD 0x40 -> 0x82;
    # XXH32("", 0) = 0x02CC5D05 is the content checksum of an empty frame
    sink = io.BytesIO()
    S.LZ4FrameOutputStream(sink, engine=engine).close()
    assert sink.getvalue() == bytes.fromhex("04224d18" "60" "70" "73" "00000000")
    sink = io.BytesIO()
    S.LZ4FrameOutputStream(sink, S.BLOCKSIZE.SIZE_64KB, -1, engine=engine).close()
    assert sink.getvalue() == bytes.fromhex("04224d18" "60" "40" "82" "00000000")
    sink = io.BytesIO()
    S.LZ4FrameOutputStream(sink, S.BLOCKSIZE.SIZE_64KB, -1, S.FLG.Bits.BLOCK_INDEPENDENCE, S.FLG.Bits.CONTENT_CHECKSUM,
                           engine=engine).close()
    v = sink.getvalue()
    assert v[:6] == bytes.fromhex("04224d18" "64" "40") and v[7:] == bytes.fromhex("00000000" "055dcc02")


def case_frame_flush_and_bytewise(S, engine, port, data):
    B = S.FLG.Bits
    small = data[:150000]
    fr = frame_bytes(S, small, engine, S.BLOCKSIZE.SIZE_64KB, (B.BLOCK_INDEPENDENCE,), chunk=50000, flush_at=0)
    _, _, _, _, blocks, _, _ = parse_frame(fr)
    # skip / available / mark on the frame reader (LZ4FrameInputStream.java:361-404)
    rd = S.LZ4FrameInputStream(io.BytesIO(fr), engine=engine, batchBlocks=1)
    assert rd.available() == 0 and rd.skip(0) == 0 and not rd.markSupported()
    k = rd.skip(1234)
    assert k == 1234 and rd.available() == 50000 - 1234 and rd.read(10) == small[1234:1244]
    tot = 1244
    while True:
        k = rd.skip(1 << 20)
        if k == 0:
            break
        tot += k
    assert tot == len(small)
    for op in (lambda: rd.mark(1), rd.reset):
        try:
            op()
            assert False
        except NotImplementedError:
            pass
    # flush() after the first 50000 bytes cuts a short block there (LZ4FrameOutputStream.java:279-286)
    assert [len(b) if r else len(port.decompress_safe(b, 65536)) for r, b, _ in blocks] == [50000, 65536, 150000 - 50000 - 65536]
    rd = S.LZ4FrameInputStream(io.BytesIO(fr), engine=engine, batchBlocks=2)
    got = bytearray()
    while True:
        c = rd.read(7777)
        if not c:
            break
        got += c
    assert bytes(got) == small
    sink = io.BytesIO()
    f = S.LZ4FrameOutputStream(sink, S.BLOCKSIZE.SIZE_64KB, -1, engine=engine)
    for b in small[:300]:
        f.write(b)          # write(int)
    f.close()
    assert S.LZ4FrameInputStream(io.BytesIO(sink.getvalue()), engine=engine).read() == small[:300]
    try:
        f.write(b"x")
        assert False
    except S.IllegalStateException:
        pass


def case_frame_concat_skippable_single(S, engine, data):
    B = S.FLG.Bits
    a = frame_bytes(S, data[:100000], engine, S.BLOCKSIZE.SIZE_64KB, (B.BLOCK_INDEPENDENCE, B.CONTENT_CHECKSUM))
    b = frame_bytes(S, data[100000:260000], engine, S.BLOCKSIZE.SIZE_256KB, (B.BLOCK_INDEPENDENCE, B.BLOCK_CHECKSUM))
    skip = struct.pack("<II", 0x184D2A53, 11) + b"skip me pls"
    empty = frame_bytes(S, b"", engine, S.BLOCKSIZE.SIZE_64KB, (B.BLOCK_INDEPENDENCE,))
    cat = skip + a + skip + empty + b
    assert S.LZ4FrameInputStream(io.BytesIO(cat), engine=engine).read() == data[:260000]
    one = S.LZ4FrameInputStream(io.BytesIO(a + b), readSingleFrame=True, engine=engine)
    assert one.read() == data[:100000]
    f = S.LZ4FrameInputStream(io.BytesIO(frame_bytes(S, data[:5000], engine, S.BLOCKSIZE.SIZE_64KB,
                                                    (B.BLOCK_INDEPENDENCE, B.CONTENT_SIZE), 5000)), engine=engine)
    assert f.isExpectedContentSizeDefined() and f.getExpectedContentSize() == 5000
    assert f.read() == data[:5000]


def _expect(S, buf, msg, engine, delivered=None, **kw):
    rd = S.LZ4FrameInputStream(io.BytesIO(buf), engine=engine, **kw)
    got = bytearray()
    try:
        while True:
            c = rd.read(1 << 20)
            if not c:
                break
            got += c
    except (S.IOException, RuntimeError) as e:
        assert msg in str(e), (msg, str(e))
        if delivered is not None:
            assert bytes(got) == delivered
        return
    raise AssertionError("no error, wanted " + msg)


def case_frame_errors(S, engine, data):
    B = S.FLG.Bits
    d = data[:200000]
    fr = bytearray(frame_bytes(S, d, engine, S.BLOCKSIZE.SIZE_64KB,
                               (B.BLOCK_INDEPENDENCE, B.BLOCK_CHECKSUM, B.CONTENT_CHECKSUM, B.CONTENT_SIZE), len(d)))
    _, _, _, _, blocks, _, _ = parse_frame(bytes(fr))
    _expect(S, b"\x01\x02\x03\x04" + bytes(fr[4:]), S.NOT_SUPPORTED, engine)
    _expect(S, bytes(fr[:3]), S.PREMATURE_EOS, engine)
    _expect(S, b"", S.PREMATURE_EOS, engine)
    bad = bytearray(fr); bad[14] ^= 1
    _expect(S, bytes(bad), S.DESCRIPTOR_HASH_MISMATCH, engine)
    bad = bytearray(fr); bad[4] = 0x40 | 0x1C  # block independence cleared
    _expect(S, bytes(bad), "BLOCK_INDEPENDENCE", engine)
    bad = bytearray(fr); bad[4] = 0x80 | 0x3C
    _expect(S, bytes(bad), "Version 2 is unsupported", engine)
    bad = bytearray(fr); bad[5] |= 1
    _expect(S, bytes(bad), "Reserved fields must be 0", engine)
    # corrupt a byte inside block 2's payload: blocks 0,1 are delivered, then the block checksum trips
    off2 = 15 + sum(4 + len(b) + 4 for _, b, _ in blocks[:2])
    bad = bytearray(fr); bad[off2 + 4 + 10] ^= 0x55
    _expect(S, bytes(bad), S.BLOCK_HASH_MISMATCH, engine, delivered=d[:131072], batchBlocks=8)
    _expect(S, bytes(bad), S.BLOCK_HASH_MISMATCH, engine, delivered=d[:131072], batchBlocks=1)
    bad = bytearray(fr); bad[-1] ^= 1
    _expect(S, bytes(bad), "Content checksum mismatch", engine, delivered=d)
    bad = bytearray(fr); bad[6] ^= 1  # content size; fix the descriptor checksum by brute force
    for hc in range(256):
        bad[14] = hc
        try:
            S.LZ4FrameInputStream(io.BytesIO(bytes(bad)), engine=engine).getExpectedContentSize()
            break
        except S.IOException:
            continue
    _expect(S, bytes(bad), "Size check mismatch", engine, delivered=d)
    _expect(S, bytes(fr[:-6]), S.PREMATURE_EOS, engine, delivered=d)
    _expect(S, bytes(fr[:off2 + 100]), S.PREMATURE_EOS, engine, delivered=d[:131072])
    big = bytearray(fr[:15]) + struct.pack("<I", 65537) + bytes(70000)
    _expect(S, bytes(big), "Block size 65537 exceeded max: 65536", engine)
    # no block checksum: a corrupted LZ4 block is caught by the safe decompressor
    fr2 = bytearray(frame_bytes(S, d, engine, S.BLOCKSIZE.SIZE_64KB, (B.BLOCK_INDEPENDENCE,)))
    _, _, _, _, blocks2, _, _ = parse_frame(bytes(fr2))
    assert not blocks2[1][0]
    off1 = 7 + 4 + len(blocks2[0][1])
    fr2[off1 + 4] = 0xFF  # token: huge literal+match lengths
    fr2[off1 + 5:off1 + 9] = b"\xff\xff\xff\xff"
    _expect(S, bytes(fr2), "Error decoding offset", engine, delivered=d[:65536])
    try:
        S.LZ4FrameOutputStream(io.BytesIO(), S.BLOCKSIZE.SIZE_64KB, -1, B.BLOCK_INDEPENDENCE, B.CONTENT_SIZE, engine=engine)
        assert False
    except ValueError:
        pass
    try:
        S.LZ4FrameOutputStream(io.BytesIO(), S.BLOCKSIZE.SIZE_64KB, -1, B.CONTENT_CHECKSUM, engine=engine)
        assert False
    except RuntimeError as e:
        assert "BLOCK_INDEPENDENCE" in str(e)


def case_frame_cli_interop(S, engine, data):
    """frames from the lz4 1.9.3 CLI decode here; frames written here decode with the CLI; and for matching flags
    the two byte streams are identical (the CLI's frame blocks are liblz4 fast-compressor output as well)"""
    B = S.FLG.Bits
    for args, blockSize, bits in ((["-1", "-B4", "--no-frame-crc"], S.BLOCKSIZE.SIZE_64KB, (B.BLOCK_INDEPENDENCE,)),
                                  (["-1", "-B5"], S.BLOCKSIZE.SIZE_256KB, (B.BLOCK_INDEPENDENCE, B.CONTENT_CHECKSUM)),
                                  (["-1", "-B7", "-BX"], S.BLOCKSIZE.SIZE_4MB,
                                   (B.BLOCK_INDEPENDENCE, B.CONTENT_CHECKSUM, B.BLOCK_CHECKSUM)),
                                  (["-1", "-B6", "--content-size"], S.BLOCKSIZE.SIZE_1MB,
                                   (B.BLOCK_INDEPENDENCE, B.CONTENT_CHECKSUM, B.CONTENT_SIZE))):
        d = data * 5 if blockSize == S.BLOCKSIZE.SIZE_4MB else data  # the CLI lowers the block-size id for small inputs
        theirs = cli(args, d)
        assert S.LZ4FrameInputStream(io.BytesIO(theirs), engine=engine).read() == d
        ours = frame_bytes(S, d, engine, blockSize, bits, len(d) if B.CONTENT_SIZE in bits else -1)
        assert cli(["-d"], ours) == d
        assert ours == theirs
    # block-dependent CLI frames are refused like the reference refuses them
    _expect(S, cli(["-1", "-B4", "-BD"], data), "BLOCK_INDEPENDENCE", engine)


def block_stream_bytes(S, data, engine, blockSize, chunk=70000, batchBlocks=4, syncFlush=False, flush_at=None):
    sink = io.BytesIO()
    f = S.LZ4BlockOutputStream(sink, blockSize, engine=engine, syncFlush=syncFlush, batchBlocks=batchBlocks)
    for i in range(0, len(data), chunk):
        f.write(data[i:i + chunk])
        if flush_at is not None and i // chunk == flush_at:
            f.flush()
    f.close()
    return sink.getvalue()


def case_block_stream(S, engine, port, data):
    d = data[:400000]
    for blockSize in (64, 1000, 1 << 16, 1 << 20):
        dd = d[:5000] if blockSize == 64 else d
        st = block_stream_bytes(S, dd, engine, blockSize)
        level = max(0, (blockSize - 1).bit_length() - 10)
        p, i = 0, 0
        while True:
            assert st[p:p + 8] == b"LZ4Block"
            token = st[p + 8]
            clen, olen, check = struct.unpack_from("<iii", st, p + 9)
            assert token & 0x0F == level
            p += 21
            if olen == 0:
                assert token & 0xF0 == 0x10 and clen == 0 and check == 0 and p == len(st)
                break
            raw = dd[i:i + olen]
            assert olen == min(blockSize, len(dd) - i)
            comp = port.compress_fast(raw)
            if len(comp) >= olen:
                assert token & 0xF0 == 0x10 and st[p:p + clen] == raw
            else:
                assert token & 0xF0 == 0x20 and st[p:p + clen] == comp
            assert check == port.xxh32(raw, 0x9747B28C) & 0x0FFFFFFF
            p += clen
            i += olen
        assert i == len(dd)
        for bb in (1, 7):
            assert S.LZ4BlockInputStream(io.BytesIO(st), engine=engine, batchBlocks=bb).read() == dd
    # caller-supplied Checksum (LZ4BlockStreamingTest.java: Adler32 / CRC32 besides the default): stored per block, verified on read
    import zlib

    class ZChecksum:  # java.util.zip.Adler32 / CRC32 contract over zlib
        def __init__(self, fn, init):
            self.fn, self.init, self.v = fn, init, init

        def reset(self):
            self.v = self.init

        def update(self, buf, off, n):
            self.v = self.fn(bytes(buf[off:off + n]), self.v)

        def getValue(self):
            return self.v & 0xFFFFFFFF
    for fn, init in ((zlib.adler32, 1), (zlib.crc32, 0)):
        sink = io.BytesIO()
        w = S.LZ4BlockOutputStream(sink, 1 << 14, engine=engine, batchBlocks=3, checksum=ZChecksum(fn, init))
        w.write(d[:100000]); w.close()
        st = sink.getvalue()
        olen, check = struct.unpack_from("<iI", st, 13)
        assert olen == 1 << 14 and check == fn(d[:1 << 14], init) & 0xFFFFFFFF
        assert S.LZ4BlockInputStream(io.BytesIO(st), engine=engine, checksum=ZChecksum(fn, init)).read() == d[:100000]
        try:  # the default checksum does not accept it
            S.LZ4BlockInputStream(io.BytesIO(st), engine=engine).read()
            assert False, "a stream written with another checksum must be rejected"
        except S.IOException as e:
            assert str(e) == "Stream is corrupted"
    # skip / available / mark (LZ4BlockStreamingTest.java skip cases; LZ4BlockInputStream.java:134-136, :176-189, :288-302)
    st = block_stream_bytes(S, d[:200000], engine, 1 << 14)
    rd = S.LZ4BlockInputStream(io.BytesIO(st), engine=engine, batchBlocks=2)
    assert rd.available() == 0 and rd.skip(0) == 0 and rd.skip(-5) == 0 and not rd.markSupported()
    got, pos = bytearray(), 0
    while True:
        k = rd.skip(1000)
        if k == 0:
            break
        assert 0 < k <= 1000
        pos += k
        piece = rd.read(777)
        assert piece == d[pos:pos + len(piece)] and rd.available() >= 0
        pos += len(piece)
    assert pos == 200000 and rd.read(10) == b"" and rd.skip(10) == 0
    rd.mark(5)
    try:
        rd.reset()
        assert False
    except S.IOException as e:
        assert str(e) == "mark/reset not supported"
    # syncFlush cuts a block at flush(); concatenated streams need stopOnEmptyBlock=False
    st = block_stream_bytes(S, d[:100000], engine, 1 << 16, chunk=30000, syncFlush=True, flush_at=0)
    assert struct.unpack_from("<i", st, 13)[0] == 30000
    two = st + block_stream_bytes(S, d[100000:150000], engine, 1 << 12)
    assert S.LZ4BlockInputStream(io.BytesIO(two), engine=engine).read() == d[:100000]
    assert S.LZ4BlockInputStream(io.BytesIO(two), stopOnEmptyBlock=False, engine=engine).read() == d[:150000]

    def expect(buf, delivered=None, exc=S.IOException, **kw):
        rd = S.LZ4BlockInputStream(io.BytesIO(buf), engine=engine, **kw)
        got = bytearray()
        try:
            while True:
                c = rd.read(50000)
                if not c:
                    break
                got += c
        except exc as e:
            if delivered is not None:
                assert bytes(got) == delivered
            return str(e)
        raise AssertionError("no error")

    st = bytearray(block_stream_bytes(S, d[:200000], engine, 1 << 16))
    assert expect(bytes(st[:-1]), d[:200000], S.EOFException) == S.PREMATURE_EOS
    bad = bytearray(st); bad[3] ^= 1
    assert expect(bytes(bad)) == "Stream is corrupted"
    clen0 = struct.unpack_from("<i", st, 9)[0]
    second = 21 + clen0
    bad = bytearray(st); bad[second + 8] = 0x30 | 6
    assert expect(bytes(bad), d[:65536]) == "Stream is corrupted"
    bad = bytearray(st); bad[second + 17] ^= 4  # checksum field of block 1
    assert expect(bytes(bad), d[:65536], batchBlocks=8) == "Stream is corrupted"
    bad = bytearray(st); bad[second + 21 + 30] ^= 0x10  # payload of block 1
    assert expect(bytes(bad), d[:65536], batchBlocks=8) == "Stream is corrupted"
    bad = bytearray(st); struct.pack_into("<i", bad, second + 13, 65537)  # originalLen > 1 << level
    assert expect(bytes(bad), d[:65536]) == "Stream is corrupted"
    for bs in (63, (1 << 25) + 1):
        try:
            S.LZ4BlockOutputStream(io.BytesIO(), bs, engine=engine)
            assert False
        except ValueError as e:
            assert "blockSize must be" in str(e)


def case_with_length(S, engine, port, data, hc_engine=None):
    bufs = [data[:0], data[:1], data[:13], data[1000:70000], data[200000:500000], bytes(5000)]
    cw = S.LZ4CompressorWithLength(engine=engine)
    outs = cw.compressMany(bufs)
    for b, o in zip(bufs, outs):
        assert o == struct.pack("<i", len(b)) + port.compress_fast(b)
        assert S.LZ4DecompressorWithLength.getDecompressedLength(o) == len(b)
        assert S.LZ4DecompressorWithLength.getDecompressedLength(b"zz" + o, 2) == len(b)
    assert cw.maxCompressedLength(1000) == 1000 + 1000 // 255 + 16 + 4
    assert cw.compress(b"ab" + bufs[3] + b"cd", 2, len(bufs[3])) == outs[3]
    for fast in (True, False):
        dw = S.LZ4DecompressorWithLength(fast=fast, engine=engine)
        assert dw.decompressMany(outs) == bufs
        assert dw.decompress(b"q" + outs[4], 1) == bufs[4]
        bad = bytearray(outs[3]); bad[4] = 0xFF; bad[5:9] = b"\xff" * 4
        try:
            dw.decompress(bytes(bad))
            assert False
        except Exception as e:
            assert "Error decoding offset" in str(e)
    if hc_engine is not None:
        o = S.LZ4CompressorWithLength(engine=hc_engine).compressMany(bufs[3:5])
        assert [x[4:] for x in o] == [port.compress_hc(b, 9) for b in bufs[3:5]]
        assert S.LZ4DecompressorWithLength(engine=engine).decompressMany(o) == bufs[3:5]
