"""Batched twins of the reference's stream/container classes -- SURVEY.md section 8(f), the callers either side
of the block hot path.  The container logic (headers, flags, checks, error messages) follows the reference
class by class; what changes is the cadence: the reference calls `compressor.compress` /
`decompressor.decompress` once per block (LZ4FrameOutputStream.java:199-235, LZ4BlockOutputStream.java:203-227,
LZ4FrameInputStream.java:258-322, LZ4BlockInputStream.java:191-264), these classes queue `batchBlocks`
independent blocks and hand them to ONE `LZ4HIPBatch` launch (and one xxh32 launch for the per-block checksums).

    LZ4FrameOutputStream / LZ4FrameInputStream     LZ4 Frame format v1.5.1 (magic 0x184D2204), interoperable with
                                                   the `lz4` CLI and the reference's classes of the same name
    LZ4BlockOutputStream / LZ4BlockInputStream     lz4-java's own "LZ4Block" container (21-byte block headers)
    LZ4CompressorWithLength / LZ4DecompressorWithLength   4-byte little-endian length prefix

Every data byte is compressed, decompressed and hashed by liblz4hip through `engine` (default: the GPU batch
engine); nothing here falls back to a CPU codec.  `engine` is injectable exactly like the reference's
constructors take a compressor/decompressor/checksum -- the CPU test-suite injects an oracle-backed engine to
check the container logic without a GPU.
"""
import io
import struct

from . import LZ4Exception, LZ4HIPBatch, maxCompressedLength

_U32 = struct.Struct("<I")
_I32 = struct.Struct("<i")
_U64 = struct.Struct("<Q")


class IOException(IOError):
    """java.io.IOException as thrown by the reference streams"""


class EOFException(IOException):
    pass


class HIPEngine:
    """the batch engine the streams drive: fast or HC compressor + both decompressors + xxh32"""

    def __init__(self, hcLevel=None):
        self.hcLevel = hcLevel

    def compress(self, src, srcOff, srcLen, dst, dstOff, dstCap):
        if self.hcLevel is None:
            return LZ4HIPBatch.compress(src, srcOff, srcLen, dst, dstOff, dstCap)
        return LZ4HIPBatch.compressHC(src, srcOff, srcLen, dst, dstOff, dstCap, self.hcLevel)

    decompressSafe = staticmethod(LZ4HIPBatch.decompressSafe)
    decompressFast = staticmethod(LZ4HIPBatch.decompressFast)
    xxh32 = staticmethod(LZ4HIPBatch.xxh32)

    def containerBlocks(self, kind, data, blockSize, blockChecksum=False):
        """round 3: the container bytes of a batch of blocks are laid out on the device behind the compress launch (raw fallback,
        size scan, headers, payload compaction, checksums) -- no host pass between compress and the copy back.  Engines without
        this method (the CPU suite's oracle-backed one) make the writers assemble on the host, byte for byte the same."""
        # (the C ABI reads level 0 as "fast"; an HC engine built with a level < 1 means lz4-java's default HC level, as
        # LZ4Factory.highCompressor(level) does (LZ4Factory.java:263-270: level < 1 -> 9) and as the host-assembly path behaves)
        lv = 0 if self.hcLevel is None else (9 if self.hcLevel < 1 else self.hcLevel)
        return LZ4HIPBatch.containerBlocks(kind, data, blockSize, blockChecksum, lv)

    def containerDecode(self, kind, body, maxBlock, nMax, blockChecksum=False):
        """round 4: the READ side on the device -- size words / 21-byte headers walked, block checksums verified, blocks decoded or
        copied there (lz4hip_container_decode); engines without this method make the readers walk the headers on the host"""
        return LZ4HIPBatch.containerDecode(kind, body, maxBlock, nMax, blockChecksum)

    @staticmethod
    def newStreamingHash32(seed):
        """content checksum state (LZ4FrameOutputStream.java:116: `XXHashFactory...newStreamingHash32(0)`)"""
        from . import StreamingXXHash32
        return StreamingXXHash32(seed)


def _compress_batch(engine, data, blockSize):
    """compress data[i*blockSize : +blockSize] for all i in one launch -> (dst, bound, [compressedLength])"""
    n = (len(data) + blockSize - 1) // blockSize
    bound = maxCompressedLength(blockSize)
    dst = bytearray(n * bound)
    offs = [i * blockSize for i in range(n)]
    lens = [min(blockSize, len(data) - o) for o in offs]
    sizes = engine.compress(data, offs, lens, dst, [i * bound for i in range(n)], [bound] * n)
    for s in sizes:
        if s <= 0:
            raise LZ4Exception("maxDestLen is too small")  # LZ4JNICompressor.java:39-41
    return dst, bound, lens, sizes


# =================================================================================================
# LZ4 Frame format
# =================================================================================================
class BLOCKSIZE:
    """LZ4FrameOutputStream.java:65-83"""
    SIZE_64KB, SIZE_256KB, SIZE_1MB, SIZE_4MB = 4, 5, 6, 7

    @staticmethod
    def valueOf(indicator):
        if indicator not in (4, 5, 6, 7):
            raise ValueError("Block size must be 4-7. Cannot use value of [%d]" % indicator)
        return indicator

    @staticmethod
    def maximumSize(indicator):  # BD.getBlockMaximumSize, LZ4FrameOutputStream.java:397-399: 2^(2n+8)
        return 1 << (2 * indicator + 8)


class FLG:
    """LZ4FrameOutputStream.java:296-372"""

    class Bits:
        RESERVED_0, RESERVED_1, CONTENT_CHECKSUM, CONTENT_SIZE, BLOCK_CHECKSUM, BLOCK_INDEPENDENCE = 0, 1, 2, 3, 4, 5

    DEFAULT_VERSION = 1

    def __init__(self, version, bits):
        self.version = version
        self.bits = bits & 0x3F
        self._validate()

    @classmethod
    def of(cls, *bits):
        m = 0
        for b in bits:
            m |= 1 << b
        return cls(cls.DEFAULT_VERSION, m)

    @classmethod
    def fromByte(cls, flg):
        return cls((flg >> 6) & 3, flg & 0x3F)

    def toByte(self):
        return self.bits | ((self.version & 3) << 6)

    def isEnabled(self, bit):
        return bool(self.bits >> bit & 1)

    def _validate(self):  # LZ4FrameOutputStream.java:354-367
        if self.isEnabled(self.Bits.RESERVED_0):
            raise RuntimeError("Reserved0 field must be 0")
        if self.isEnabled(self.Bits.RESERVED_1):
            raise RuntimeError("Reserved1 field must be 0")
        if not self.isEnabled(self.Bits.BLOCK_INDEPENDENCE):
            raise RuntimeError("Dependent block stream is unsupported (BLOCK_INDEPENDENCE must be set)")
        if self.version != self.DEFAULT_VERSION:
            raise RuntimeError("Version %d is unsupported" % self.version)


class BD:
    """LZ4FrameOutputStream.java:374-404"""
    RESERVED_MASK = 0x8F

    def __init__(self, blockSizeValue):
        self.blockSizeValue = BLOCKSIZE.valueOf(blockSizeValue)

    @classmethod
    def fromByte(cls, bd):
        if bd & cls.RESERVED_MASK:
            raise RuntimeError("Reserved fields must be 0")
        return cls((bd >> 4) & 7)

    def getBlockMaximumSize(self):
        return BLOCKSIZE.maximumSize(self.blockSizeValue)

    def toByte(self):
        return (self.blockSizeValue & 7) << 4


MAGIC = 0x184D2204
MAGIC_SKIPPABLE_BASE = 0x184D2A50
LZ4_MAX_HEADER_LENGTH = 4 + 1 + 1 + 8 + 1
LZ4_FRAME_INCOMPRESSIBLE_MASK = 0x80000000
PREMATURE_EOS = "Stream ended prematurely"
NOT_SUPPORTED = "Stream unsupported"
BLOCK_HASH_MISMATCH = "Block checksum mismatch"
DESCRIPTOR_HASH_MISMATCH = "Stream frame descriptor corrupted"
CLOSED_STREAM = "The stream is already closed"


class LZ4FrameOutputStream(io.RawIOBase):
    """Twin of lz4/LZ4FrameOutputStream.java.  `out` is any object with write(bytes).  Blocks are queued until
    `batchBlocks` full blocks are buffered (or flush()/close()), then compressed in one launch; the bytes
    written are exactly the reference's for the same input and flags (flush() cuts a short block at the same
    place the reference's flush() does)."""

    def __init__(self, out, blockSize=BLOCKSIZE.SIZE_4MB, knownSize=-1, *bits, engine=None, batchBlocks=64):
        super().__init__()
        self.out = out
        self.engine = engine or HIPEngine()
        if not bits:
            bits = (FLG.Bits.BLOCK_INDEPENDENCE,)
        self.flg = FLG.of(*bits)
        self.bd = BD(blockSize)
        self.maxBlockSize = self.bd.getBlockMaximumSize()
        self.knownSize = knownSize
        self.batchBlocks = max(1, batchBlocks)
        self.buffer = bytearray()
        self.content = self.engine.newStreamingHash32(0) if self.flg.isEnabled(FLG.Bits.CONTENT_CHECKSUM) else None
        self.finished = False
        self._closed = False
        if self.flg.isEnabled(FLG.Bits.CONTENT_SIZE) and knownSize < 0:
            raise ValueError("Known size must be greater than zero in order to use the known size feature")
        self._writeHeader()

    def writable(self):
        return True

    def _writeHeader(self):  # LZ4FrameOutputStream.java:172-190
        head = bytearray(_U32.pack(MAGIC))
        head.append(self.flg.toByte())
        head.append(self.bd.toByte())
        if self.flg.isEnabled(FLG.Bits.CONTENT_SIZE):
            head += _U64.pack(self.knownSize)
        hc = (self.engine.xxh32(bytes(head[4:]), [0], [len(head) - 4], 0)[0] >> 8) & 0xFF
        head.append(hc)
        self.out.write(bytes(head))

    def _ensureNotFinished(self):
        if self.finished or self._closed:
            raise IllegalStateException(CLOSED_STREAM)

    def write(self, b):
        self._ensureNotFinished()
        if isinstance(b, int):
            b = bytes((b & 0xFF,))
        self.buffer += b
        if len(self.buffer) >= self.batchBlocks * self.maxBlockSize:
            full = len(self.buffer) // self.maxBlockSize * self.maxBlockSize
            self._writeBlocks(full)
        return len(b)

    def _writeBlocks(self, nbytes):
        """LZ4FrameOutputStream.writeBlock (:199-235) for every block of buffer[:nbytes] at once"""
        if nbytes == 0:
            return
        data = bytes(self.buffer[:nbytes])
        del self.buffer[:nbytes]
        if self.content is not None:
            self.content.update(data, 0, len(data))  # :211-213: the content checksum streams over the uncompressed bytes
        block_checksum = self.flg.isEnabled(FLG.Bits.BLOCK_CHECKSUM)
        if hasattr(self.engine, "containerBlocks"):   # assembled on the device
            self.out.write(self.engine.containerBlocks(LZ4HIPBatch.FRAME_BLOCKS, data, self.maxBlockSize, block_checksum))
            return
        dst, bound, lens, sizes = _compress_batch(self.engine, data, self.maxBlockSize)
        outb = bytearray()
        spans = []
        for i, (raw_len, clen) in enumerate(zip(lens, sizes)):
            if clen >= raw_len:  # store uncompressed if compression does not gain
                payload = data[i * self.maxBlockSize: i * self.maxBlockSize + raw_len]
                outb += _U32.pack(raw_len | LZ4_FRAME_INCOMPRESSIBLE_MASK)
            else:
                payload = dst[i * bound: i * bound + clen]
                outb += _U32.pack(clen)
            spans.append((len(outb), len(payload)))
            outb += payload
            if block_checksum:
                outb += b"\0\0\0\0"
        if block_checksum:
            hashes = self.engine.xxh32(outb, [s for s, _ in spans], [n for _, n in spans], 0)
            for (s, n), h in zip(spans, hashes):
                _U32.pack_into(outb, s + n, h)
        self.out.write(bytes(outb))

    def _writeEndMark(self):  # :244-252
        tail = bytearray(_U32.pack(0))
        if self.content is not None:
            tail += _U32.pack(self.content.getValue())
            self.content.close()
        self.out.write(bytes(tail))
        self.finished = True

    def flush(self):  # :279-286: flush() writes the pending (possibly short) block
        if self._closed or self.finished or not hasattr(self, "buffer"):
            return
        self._writeBlocks(len(self.buffer))
        if hasattr(self.out, "flush"):
            self.out.flush()

    def close(self):  # :289-298
        if self._closed:
            return
        if not self.finished:
            self._writeBlocks(len(self.buffer))
            self._writeEndMark()
            if hasattr(self.out, "flush"):
                self.out.flush()
        self._closed = True
        super().close()


class IllegalStateException(RuntimeError):
    pass


class _Reader:
    """InputStream helpers over any object with read(n)"""

    def __init__(self, inp):
        self.inp = inp
        self.back = b""      # bytes handed back by the device read path (container bytes it did not consume)

    def unread(self, b):
        self.back = bytes(b) + self.back

    def seekable(self):
        try:
            return bool(self.inp.seekable())
        except Exception:
            return False

    def give_back(self):
        """the container ended and the caller goes on reading `inp` itself (readSingleFrame / stopOnEmptyBlock): the bytes the device
        read path took beyond the container's end go back to `inp` -- the reference's readers consume exactly the container
        (LZ4FrameInputStream.java:258-322, LZ4BlockInputStream.java:191-264).  Only seekable inputs get the device path in those modes."""
        if self.back:
            self.inp.seek(-len(self.back), 1)
            self.back = b""

    def read_upto(self, n):
        parts, got = [], 0
        if self.back:
            parts.append(self.back[:n]); got = len(parts[0]); self.back = self.back[got:]
        while got < n:
            c = self.inp.read(n - got)
            if not c:
                break
            parts.append(c)
            got += len(c)
        return b"".join(parts)

    def read_fully(self, n, exc=IOException):
        b = self.read_upto(n)
        if len(b) < n:
            raise exc(PREMATURE_EOS)
        return b


class LZ4FrameInputStream(io.RawIOBase):
    """Twin of lz4/LZ4FrameInputStream.java: concatenated frames, skippable frames, descriptor / block /
    content checksums, content-size check -- same checks, same messages.  Block headers of up to `batchBlocks`
    blocks are parsed ahead, then the block checksums are verified in one xxh32 launch and the compressed
    blocks decoded in one safe-decompress launch.  A defect in block k surfaces when the reader reaches
    block k, after the bytes of blocks < k have been delivered, as in the reference."""

    def __init__(self, inp, readSingleFrame=False, engine=None, batchBlocks=64, hostWalk=False):
        super().__init__()
        self.hostWalk = hostWalk   # True: headers walked and checksums compared on the host around the batch launches (rounds 1-3)
        self.r = _Reader(inp)
        self.engine = engine or HIPEngine()
        self.readSingleFrame = readSingleFrame
        self.batchBlocks = max(1, batchBlocks)
        self.firstFrameHeaderRead = False
        self.frame_finished = True
        self.flg = self.bd = None
        self.ready = bytearray()  # decoded bytes not yet handed to the reader
        self.pending_exc = None    # raised once `ready` is drained
        self.expectedContentSize = -1
        self.totalContentSize = 0
        self.content = None

    def readable(self):
        return True

    # -- frame level ----------------------------------------------------------------------------
    def _nextFrameInfo(self):  # LZ4FrameInputStream.java:124-160
        while True:
            head = self.r.read_upto(4)
            if len(head) == 0 and self.firstFrameHeaderRead:
                return False  # clean end between frames
            if len(head) < 4:
                raise IOException(PREMATURE_EOS)
            magic = _U32.unpack(head)[0]
            if magic == MAGIC:
                self._readHeader()
                return True
            if magic >> 4 == MAGIC_SKIPPABLE_BASE >> 4:
                skip = _U32.unpack(self.r.read_fully(4))[0]
                self.r.read_fully(skip)
                self.firstFrameHeaderRead = True
            else:
                raise IOException(NOT_SUPPORTED)

    def _readHeader(self):  # :180-224
        fb = self.r.read_fully(2)
        header = bytearray(fb)
        self.flg = FLG.fromByte(fb[0])
        self.bd = BD.fromByte(fb[1])
        if self.flg.isEnabled(FLG.Bits.CONTENT_SIZE):
            cs = self.r.read_fully(8)
            self.expectedContentSize = _U64.unpack(cs)[0]
            header += cs
        self.totalContentSize = 0
        h = (self.engine.xxh32(bytes(header), [0], [len(header)], 0)[0] >> 8) & 0xFF
        expected = self.r.read_fully(1)[0]
        if h != expected:
            raise IOException(DESCRIPTOR_HASH_MISMATCH)
        self.maxBlockSize = self.bd.getBlockMaximumSize()
        if self.content is not None:
            self.content.close()
        self.content = self.engine.newStreamingHash32(0) if self.flg.isEnabled(FLG.Bits.CONTENT_CHECKSUM) else None
        self.firstFrameHeaderRead = True
        self.frame_finished = False

    # -- block level ----------------------------------------------------------------------------
    def _readBlocksDevice(self):
        """the same on the device (engine.containerDecode): one chunk of container bytes goes over, the decoded blocks and a stop
        reason come back; what the device did not consume is handed back to the reader.  Same checks in the same order, same
        messages; a defect in block k surfaces after the bytes of the blocks before it."""
        B = LZ4HIPBatch
        block_checksum = self.flg.isEnabled(FLG.Bits.BLOCK_CHECKSUM)
        want = self.batchBlocks * (self.maxBlockSize + 8) + 4
        chunk = self.r.read_upto(want)
        at_eof = len(chunk) < want
        decoded, sizes, consumed, why, code = self.engine.containerDecode(B.FRAME_BLOCKS, chunk, self.maxBlockSize, self.batchBlocks, block_checksum)
        rest = chunk[consumed:]
        if decoded:
            first = len(self.ready)
            self.ready += decoded
            self.totalContentSize += len(decoded)
            if self.content is not None:  # one update per batch of decoded blocks
                self.content.update(self.ready, first, len(decoded))
        self.r.unread(rest)
        if why == B.CR_END:
            self._endMark()
        elif why == B.CR_BLOCK_TOO_BIG:
            self.r.read_upto(4)
            self.pending_exc = IOException("Block size %s exceeded max: %s" % (_U32.unpack(rest[:4])[0] & ~LZ4_FRAME_INCOMPRESSIBLE_MASK & 0xFFFFFFFF, self.maxBlockSize))
        elif why == B.CR_BLOCK_CHECKSUM:
            self.pending_exc = IOException(BLOCK_HASH_MISMATCH)
        elif why == B.CR_DECODE:  # LZ4JNISafeDecompressor.java:39-41, wrapped in IOException (:307-311)
            self.pending_exc = IOException(LZ4Exception("Error decoding offset %d of input buffer" % (-code)))
        elif why in (B.CR_TRUNCATED, B.CR_MORE) and at_eof:
            self.r.read_upto(len(rest))
            self.pending_exc = IOException(PREMATURE_EOS)
        # (CR_MORE / CR_SLOTS / a chunk that ended inside a block: the next call goes on from the bytes handed back)

    def _endMark(self):
        """behind the end mark (LZ4FrameInputStream.java:264-276): content checksum, content size"""
        try:
            if self.flg.isEnabled(FLG.Bits.CONTENT_CHECKSUM):
                stored = _U32.unpack(self.r.read_fully(4))[0]
                if stored != self.content.getValue():
                    raise IOException("Content checksum mismatch")
            if self.flg.isEnabled(FLG.Bits.CONTENT_SIZE) and self.expectedContentSize != self.totalContentSize:
                raise IOException("Size check mismatch")
        except IOException as e:
            self.pending_exc = e
            return
        self.frame_finished = True

    def _readBlocks(self):
        """LZ4FrameInputStream.readBlock (:258-322) for up to batchBlocks blocks"""
        # (the device path takes a whole chunk of container bytes from `inp`: with readSingleFrame the caller reads on behind the
        # frame, so it needs an input it can hand the surplus back to -- round-4 advisor; a pipe or socket gets the host walk, which
        # consumes exactly the frame and never waits for bytes beyond it)
        if not self.hostWalk and hasattr(self.engine, "containerDecode") and (not self.readSingleFrame or self.r.seekable()):
            self._readBlocksDevice()
            if self.readSingleFrame and self.frame_finished:
                self.r.give_back()
            return
        blocks = []  # (compressed?, payload, stored checksum | None)
        end_mark = False
        exc = None
        block_checksum = self.flg.isEnabled(FLG.Bits.BLOCK_CHECKSUM)
        while len(blocks) < self.batchBlocks:
            try:
                word = _U32.unpack(self.r.read_fully(4))[0]
                compressed = (word & LZ4_FRAME_INCOMPRESSIBLE_MASK) == 0
                size = word & ~LZ4_FRAME_INCOMPRESSIBLE_MASK & 0xFFFFFFFF
                if size == 0:
                    end_mark = True
                    break
                if size > self.maxBlockSize:
                    raise IOException("Block size %s exceeded max: %s" % (size, self.maxBlockSize))
                payload = self.r.read_fully(size)
                stored = _U32.unpack(self.r.read_fully(4))[0] if block_checksum else None
                blocks.append((compressed, payload, stored))
            except IOException as e:
                exc = e
                break
        n = len(blocks)
        if n:
            src = b"".join(p for _, p, _ in blocks)
            offs, o = [], 0
            for _, p, _ in blocks:
                offs.append(o)
                o += len(p)
            lens = [len(p) for _, p, _ in blocks]
            bad = n
            bad_exc = None
            if block_checksum:
                hashes = self.engine.xxh32(src, offs, lens, 0)
                for i, (h, (_, _, stored)) in enumerate(zip(hashes, blocks)):
                    if h != stored:
                        bad, bad_exc = i, IOException(BLOCK_HASH_MISMATCH)
                        break
            cidx = [i for i in range(bad) if blocks[i][0]]
            outs = [None] * bad
            if cidx:
                dst = bytearray(len(cidx) * self.maxBlockSize)
                res = self.engine.decompressSafe(src, [offs[i] for i in cidx], [lens[i] for i in cidx], dst,
                                                 [k * self.maxBlockSize for k in range(len(cidx))],
                                                 [self.maxBlockSize] * len(cidx))
                for k, i in enumerate(cidx):
                    if res[k] < 0:  # LZ4JNISafeDecompressor.java:39-41, wrapped in IOException (:307-311)
                        bad, bad_exc = i, IOException(LZ4Exception("Error decoding offset %d of input buffer" % (-res[k])))
                        break
                    outs[i] = dst[k * self.maxBlockSize: k * self.maxBlockSize + res[k]]
            first = len(self.ready)
            for i in range(bad):
                piece = outs[i] if blocks[i][0] else blocks[i][1]
                self.ready += piece
                self.totalContentSize += len(piece)
            if self.content is not None and len(self.ready) > first:  # one update per batch of decoded blocks
                self.content.update(self.ready, first, len(self.ready) - first)
            if bad_exc is not None:
                self.pending_exc = bad_exc
                return
        if exc is not None:
            self.pending_exc = exc
            return
        if end_mark:
            self._endMark()

    def _fill(self):
        """False at end of stream"""
        while not self.ready:
            if self.pending_exc is not None:
                e, self.pending_exc = self.pending_exc, None
                raise e
            if not self.firstFrameHeaderRead or self.frame_finished:
                if self.firstFrameHeaderRead and self.frame_finished and self.readSingleFrame and self.flg is not None:
                    return False
                if not self._nextFrameInfo():
                    return False
                if self.frame_finished:  # only skippable frames so far
                    continue
            self._readBlocks()
        return True

    def read(self, size=-1):
        if size is None or size < 0:
            return self.readall()
        if size == 0:
            return b""
        if not self._fill():
            return b""
        out = bytes(self.ready[:size])
        del self.ready[:size]
        return out

    def readall(self):
        parts = []
        while self._fill():
            parts.append(bytes(self.ready))
            self.ready.clear()
        return b"".join(parts)

    def readinto(self, b):
        data = self.read(len(b))
        b[: len(data)] = data
        return len(data)

    def skip(self, n):  # LZ4FrameInputStream.java:361-379: at most what is decoded and waiting (here: up to one batch of blocks)
        if n <= 0 or not self._fill():
            return 0
        k = min(n, len(self.ready))
        del self.ready[:k]
        return k

    def available(self):  # :382-384
        return len(self.ready)

    def markSupported(self):  # :402-404
        return False

    def mark(self, readlimit):  # :392-394
        raise NotImplementedError("mark not supported")

    def reset(self):  # :397-399
        raise NotImplementedError("reset not supported")

    def getExpectedContentSize(self):  # :375-383
        if not self.firstFrameHeaderRead:
            self._fill()
        return self.expectedContentSize

    def isExpectedContentSizeDefined(self):
        return self.getExpectedContentSize() >= 0


# =================================================================================================
# lz4-java "LZ4Block" container
# =================================================================================================
BLOCK_MAGIC = b"LZ4Block"
MAGIC_LENGTH = len(BLOCK_MAGIC)
HEADER_LENGTH = MAGIC_LENGTH + 1 + 4 + 4 + 4  # LZ4BlockOutputStream.java:42-47
COMPRESSION_LEVEL_BASE = 10
MIN_BLOCK_SIZE = 64
MAX_BLOCK_SIZE = 1 << (COMPRESSION_LEVEL_BASE + 0x0F)
COMPRESSION_METHOD_RAW = 0x10
COMPRESSION_METHOD_LZ4 = 0x20
DEFAULT_SEED = 0x9747B28C
_CHECK_MASK = 0x0FFFFFFF  # StreamingXXHash32.asChecksum keeps 28 bits (StreamingXXHash32.java:101-107)


def _compressionLevel(blockSize):  # LZ4BlockOutputStream.java:57-69
    if blockSize < MIN_BLOCK_SIZE:
        raise ValueError("blockSize must be >= %d, got %d" % (MIN_BLOCK_SIZE, blockSize))
    if blockSize > MAX_BLOCK_SIZE:
        raise ValueError("blockSize must be <= %d, got %d" % (MAX_BLOCK_SIZE, blockSize))
    level = (blockSize - 1).bit_length()  # ceil(log2)
    return max(0, level - COMPRESSION_LEVEL_BASE)


def _user_checks(checksum, pieces):
    """per-block values of a caller-supplied java.util.zip.Checksum-like object (reset / update / getValue), as the reference
    computes them: `checksum.reset(); checksum.update(block); (int) checksum.getValue()` (LZ4BlockOutputStream.java:207-209)"""
    out = []
    for piece in pieces:
        checksum.reset()
        checksum.update(piece, 0, len(piece))
        out.append(checksum.getValue() & 0xFFFFFFFF)
    return out


class LZ4BlockOutputStream(io.RawIOBase):
    """Twin of lz4/LZ4BlockOutputStream.java.  checksum=None is the reference's default (XXH32, seed 0x9747b28c, 28 bits:
    LZ4BlockOutputStream.java:125), computed for all blocks of a batch in one launch; any object with reset() / update(buf, off,
    len) / getValue() -- the java.util.zip.Checksum contract of the 5-argument constructor (:96-101), e.g. an Adler32 or CRC32
    adapter -- is used per block on the host, as the reference does."""

    def __init__(self, out, blockSize=1 << 16, engine=None, syncFlush=False, batchBlocks=256, checksum=None):
        super().__init__()
        self.checksum = checksum
        self.out = out
        self.blockSize = blockSize
        self.compressionLevel = _compressionLevel(blockSize)
        self.engine = engine or HIPEngine()
        self.syncFlush = syncFlush
        self.batchBlocks = max(1, batchBlocks)
        self.buffer = bytearray()
        self.finished = False
        self._closed = False

    def writable(self):
        return True

    def _ensureNotFinished(self):
        if self.finished:
            raise IllegalStateException("This stream is already closed")

    def write(self, b):
        self._ensureNotFinished()
        if isinstance(b, int):
            b = bytes((b & 0xFF,))
        self.buffer += b
        if len(self.buffer) >= self.batchBlocks * self.blockSize:
            self._flushBlocks(len(self.buffer) // self.blockSize * self.blockSize)
        return len(b)

    def _flushBlocks(self, nbytes):
        """flushBufferedData (:203-227) for every block of buffer[:nbytes] at once"""
        if nbytes == 0:
            return
        data = bytes(self.buffer[:nbytes])
        del self.buffer[:nbytes]
        if self.checksum is None and hasattr(self.engine, "containerBlocks"):   # assembled on the device (default XXH32 checksum)
            self.out.write(self.engine.containerBlocks(LZ4HIPBatch.LZ4BLOCK_BLOCKS, data, self.blockSize))
            return
        dst, bound, lens, sizes = _compress_batch(self.engine, data, self.blockSize)
        if self.checksum is None:
            checks = [c & _CHECK_MASK for c in self.engine.xxh32(data, [i * self.blockSize for i in range(len(lens))], lens, DEFAULT_SEED)]
        else:
            checks = _user_checks(self.checksum, [data[i * self.blockSize: i * self.blockSize + n] for i, n in enumerate(lens)])
        outb = bytearray()
        for i, (raw_len, clen, check) in enumerate(zip(lens, sizes, checks)):
            if clen >= raw_len:
                method, clen = COMPRESSION_METHOD_RAW, raw_len
                payload = data[i * self.blockSize: i * self.blockSize + raw_len]
            else:
                method = COMPRESSION_METHOD_LZ4
                payload = dst[i * bound: i * bound + clen]
            outb += BLOCK_MAGIC
            outb.append(method | self.compressionLevel)
            outb += _U32.pack(clen) + _U32.pack(raw_len) + _U32.pack(check)
            outb += payload
        self.out.write(bytes(outb))

    def flush(self):  # :241-248
        if self._closed or not hasattr(self, "buffer"):
            return
        if self.syncFlush and not self.finished:
            self._flushBlocks(len(self.buffer))
        if hasattr(self.out, "flush"):
            self.out.flush()

    def finish(self):  # :256-268
        self._ensureNotFinished()
        self._flushBlocks(len(self.buffer))
        self.out.write(BLOCK_MAGIC + bytes((COMPRESSION_METHOD_RAW | self.compressionLevel,)) + b"\0" * 12)
        self.finished = True
        if hasattr(self.out, "flush"):
            self.out.flush()

    def close(self):
        if self._closed:
            return
        if not self.finished:
            self.finish()
        self._closed = True
        super().close()


class LZ4BlockInputStream(io.RawIOBase):
    """Twin of lz4/LZ4BlockInputStream.java (fast decompressor + default checksum).  Headers of up to
    `batchBlocks` blocks are parsed ahead; LZ4 blocks are decoded in one decompress_fast launch and every
    block's checksum verified in one xxh32 launch."""

    CORRUPTED = "Stream is corrupted"

    def __init__(self, inp, stopOnEmptyBlock=True, engine=None, batchBlocks=256, checksum=None, hostWalk=False):
        super().__init__()
        self.hostWalk = hostWalk   # True: headers walked and checksums compared on the host around the batch launches (rounds 1-3)
        self.checksum = checksum   # None: the default XXH32 (batched); else a Checksum-like object (LZ4BlockInputStream.java:71-76)
        self.r = _Reader(inp)
        self.engine = engine or HIPEngine()
        self.stopOnEmptyBlock = stopOnEmptyBlock
        self.batchBlocks = max(1, batchBlocks)
        self.ready = bytearray()
        self.pending_exc = None
        self.finished = False

    def readable(self):
        return True

    def _refillDevice(self):
        """LZ4BlockInputStream.refill for a run of blocks on the device (engine.containerDecode): headers walked with the reference's
        rules, LZ4 blocks through the fast decoder (its return value must be the header's compressed length), raw ones copied, the
        checksums of the decoded bytes compared -- every defect is "Stream is corrupted", delivered after the blocks before it.
        False: this run is for the host path (a block bigger than the first header announced)."""
        B = LZ4HIPBatch
        head = self.r.read_upto(HEADER_LENGTH)
        self.r.unread(head)
        if len(head) < HEADER_LENGTH or head[:MAGIC_LENGTH] != BLOCK_MAGIC:
            return False   # (the host path produces the reference's exception / end of stream)
        maxBlock = 1 << (COMPRESSION_LEVEL_BASE + (head[MAGIC_LENGTH] & 0x0F))
        want = self.batchBlocks * (maxBlock + HEADER_LENGTH) + HEADER_LENGTH
        chunk = self.r.read_upto(want)
        at_eof = len(chunk) < want
        decoded, sizes, consumed, why, code = self.engine.containerDecode(B.LZ4BLOCK_BLOCKS, chunk, maxBlock, self.batchBlocks, False)
        if why == B.CR_BLOCK_TOO_BIG:
            self.r.unread(chunk)
            return False
        if consumed == 0 and not decoded and why in (B.CR_TRUNCATED, B.CR_MORE) and not at_eof:
            # no progress and more input to come: the first header announces a payload longer than this chunk (compressedLen is not
            # bounded by the block size: one flipped bit is enough).  The host path reads exactly what the header says and raises
            # the reference's exception (LZ4BlockInputStream.java:236-253); asking the device again with the same bytes would
            # never end (round-4 advisor, high)
            self.r.unread(chunk)
            return False
        rest = chunk[consumed:]
        self.ready += decoded
        self.r.unread(rest)
        if why == B.CR_END:
            if self.stopOnEmptyBlock:
                self.finished = True
        elif why == B.CR_CORRUPT:
            self.pending_exc = IOException(self.CORRUPTED)
        elif why in (B.CR_TRUNCATED, B.CR_MORE) and at_eof:
            self.r.read_upto(len(rest))
            if len(rest) < HEADER_LENGTH and not self.stopOnEmptyBlock:
                self.finished = True                      # (:192-199: the stream may end at -- or inside -- a header)
            else:
                self.pending_exc = EOFException(PREMATURE_EOS)
        return True

    def _refill(self):  # LZ4BlockInputStream.java:191-264, for up to batchBlocks blocks
        # (stopOnEmptyBlock: the caller reads on behind the empty block, so the device path -- which takes whole chunks from `inp` --
        # needs an input it can hand the surplus back to; see LZ4FrameInputStream._readBlocks)
        if (self.checksum is None and not self.hostWalk and hasattr(self.engine, "containerDecode")
                and (not self.stopOnEmptyBlock or self.r.seekable()) and self._refillDevice()):
            if self.stopOnEmptyBlock and self.finished:
                self.r.give_back()
            return
        blocks = []  # (method, payload, originalLen, check)
        exc = None
        finished = False
        while len(blocks) < self.batchBlocks:
            head = self.r.read_upto(HEADER_LENGTH)
            if len(head) < HEADER_LENGTH:
                if not self.stopOnEmptyBlock:
                    finished = True
                else:
                    exc = EOFException(PREMATURE_EOS)
                break
            try:
                if head[:MAGIC_LENGTH] != BLOCK_MAGIC:
                    raise IOException(self.CORRUPTED)
                token = head[MAGIC_LENGTH]
                method = token & 0xF0
                level = COMPRESSION_LEVEL_BASE + (token & 0x0F)
                if method not in (COMPRESSION_METHOD_RAW, COMPRESSION_METHOD_LZ4):
                    raise IOException(self.CORRUPTED)
                clen, olen, check = struct.unpack_from("<iii", head, MAGIC_LENGTH + 1)
                if (olen > 1 << level or olen < 0 or clen < 0 or (olen == 0 and clen != 0) or (olen != 0 and clen == 0)
                        or (method == COMPRESSION_METHOD_RAW and olen != clen)):
                    raise IOException(self.CORRUPTED)
                if olen == 0 and clen == 0:
                    if check != 0:
                        raise IOException(self.CORRUPTED)
                    if not self.stopOnEmptyBlock:
                        continue
                    finished = True
                    break
                payload = self.r.read_fully(clen, EOFException)
                blocks.append((method, payload, olen, check & 0xFFFFFFFF))
            except IOException as e:
                exc = e
                break
        n = len(blocks)
        if n:
            bad, bad_exc = n, None
            lz = [i for i in range(n) if blocks[i][0] == COMPRESSION_METHOD_LZ4]
            raw = [None] * n
            if lz:
                # decompress_fast reads the source without a length; pad so the bounded kernel has slack
                src = b"".join(blocks[i][1] for i in lz)
                soff, o = [], 0
                for i in lz:
                    soff.append(o)
                    o += len(blocks[i][1])
                doff, o = [], 0
                for i in lz:
                    doff.append(o)
                    o += blocks[i][2]
                dst = bytearray(o)
                res = self.engine.decompressFast(src, soff, [len(blocks[i][1]) for i in lz], dst, doff,
                                                 [blocks[i][2] for i in lz])
                for k, i in enumerate(lz):
                    if res[k] < 0 or res[k] != len(blocks[i][1]):  # :246-253
                        bad, bad_exc = i, IOException(self.CORRUPTED)
                        break
                    raw[i] = bytes(dst[doff[k]: doff[k] + blocks[i][2]])
            for i in range(bad):
                if raw[i] is None:
                    raw[i] = blocks[i][1]
            if bad:
                allb = b"".join(raw[:bad])
                offs, o = [], 0
                for i in range(bad):
                    offs.append(o)
                    o += len(raw[i])
                if self.checksum is None:
                    hashes = [h & _CHECK_MASK for h in self.engine.xxh32(allb, offs, [len(raw[i]) for i in range(bad)], DEFAULT_SEED)]
                else:
                    hashes = _user_checks(self.checksum, raw[:bad])
                for i in range(bad):
                    if hashes[i] != blocks[i][3]:
                        bad, bad_exc = i, IOException(self.CORRUPTED)
                        break
            for i in range(bad):
                self.ready += raw[i]
            if bad_exc is not None:
                self.pending_exc = bad_exc
                return
        if exc is not None:
            self.pending_exc = exc
        elif finished:
            self.finished = True

    def _fill(self):
        while not self.ready:
            if self.pending_exc is not None:
                e, self.pending_exc = self.pending_exc, None
                raise e
            if self.finished:
                return False
            self._refill()
        return True

    def read(self, size=-1):
        if size is None or size < 0:
            return self.readall()
        if size == 0 or not self._fill():
            return b""
        out = bytes(self.ready[:size])
        del self.ready[:size]
        return out

    def readall(self):
        parts = []
        while self._fill():
            parts.append(bytes(self.ready))
            self.ready.clear()
        return b"".join(parts)

    def readinto(self, b):
        data = self.read(len(b))
        b[: len(data)] = data
        return len(data)

    def skip(self, n):  # LZ4BlockInputStream.java:176-189: at most what is decoded and waiting (here: up to one batch of blocks)
        if n <= 0 or self.finished and not self.ready:
            return 0
        if not self._fill():
            return 0
        k = min(n, len(self.ready))
        del self.ready[:k]
        return k

    def available(self):  # :134-136
        return len(self.ready)

    def markSupported(self):  # :288-290
        return False

    def mark(self, readlimit):  # :294-296: unsupported, silently
        pass

    def reset(self):  # :300-302
        raise IOException("mark/reset not supported")


# =================================================================================================
# length-prefixed blocks
# =================================================================================================
class LZ4CompressorWithLength:
    """lz4/LZ4CompressorWithLength.java: 4-byte little-endian decompressed length, then the block.
    compressMany() is the batched form (one launch for a list of buffers)."""

    def __init__(self, compressor=None, engine=None):
        self.compressor = compressor
        self.engine = engine or HIPEngine()

    def maxCompressedLength(self, length):  # :65-67
        return maxCompressedLength(length) + 4

    def compress(self, src, srcOff=0, srcLen=None):  # :76-110
        srcLen = len(src) - srcOff if srcLen is None else srcLen
        if self.compressor is not None:
            body = self.compressor.compress(src, srcOff, srcLen)
            return _I32.pack(srcLen) + body
        return self.compressMany([bytes(src[srcOff: srcOff + srcLen])])[0]

    def compressMany(self, bufs):
        src = b"".join(bufs)
        offs, o = [], 0
        for b in bufs:
            offs.append(o)
            o += len(b)
        caps = [maxCompressedLength(len(b)) for b in bufs]
        doffs, o = [], 0
        for c in caps:
            doffs.append(o)
            o += c
        dst = bytearray(o)
        sizes = self.engine.compress(src, offs, [len(b) for b in bufs], dst, doffs, caps)
        out = []
        for b, d, s in zip(bufs, doffs, sizes):
            if s <= 0:
                raise LZ4Exception("maxDestLen is too small")
            out.append(_I32.pack(len(b)) + bytes(dst[d: d + s]))
        return out


class LZ4DecompressorWithLength:
    """lz4/LZ4DecompressorWithLength.java.  fast=True follows the LZ4FastDecompressor constructor (:84-87: the
    result is the number of source bytes read), fast=False the LZ4SafeDecompressor one (:94-97)."""

    def __init__(self, fast=True, engine=None):
        self.fast = fast
        self.engine = engine or HIPEngine()

    @staticmethod
    def getDecompressedLength(src, srcOff=0):  # :52-54
        return _I32.unpack_from(src, srcOff)[0]

    def decompress(self, src, srcOff=0, srcLen=None):
        srcLen = len(src) - srcOff if srcLen is None else srcLen
        return self.decompressMany([bytes(src[srcOff: srcOff + srcLen])])[0]

    def decompressMany(self, bufs):
        lens = [self.getDecompressedLength(b) for b in bufs]
        for n in lens:
            if n < 0:
                raise ValueError("lengths must be >= 0")
        src = b"".join(bufs)
        soff, o = [], 0
        for b in bufs:
            soff.append(o + 4)
            o += len(b)
        doff, o = [], 0
        for n in lens:
            doff.append(o)
            o += n
        dst = bytearray(o)
        slens = [len(b) - 4 for b in bufs]
        fn = self.engine.decompressFast if self.fast else self.engine.decompressSafe
        res = fn(src, soff, slens, dst, doff, lens)
        out = []
        for i, r in enumerate(res):
            if r < 0:
                raise LZ4Exception("Error decoding offset %d of input buffer" % (4 - r))
            # safe path: LZ4SafeDecompressor.decompress(src, off, len, maxDestLen) trims to the decoded length
            out.append(bytes(dst[doff[i]: doff[i] + (lens[i] if self.fast else r)]))
        return out
