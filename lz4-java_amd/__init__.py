"""lz4-java_amd -- host-side mirror of lz4-java's plugin interface for the "HIP" family.

The reference is Java; this image has no JDK, so the host side above the C ABI
(``include/lz4hip.h`` / ``liblz4hip.so``) is mirrored here 1:1 for the test-suite and bench (the
Java classes + JNI shim a maintainer would add are in ``java/`` and ``INTEGRATION.md``):

====================================  =========================================================
reference (src/java/net/jpountz/...)  here
====================================  =========================================================
lz4/LZ4Factory.java:91-126,229-291    ``LZ4Factory.hipInstance()`` .fastCompressor() .highCompressor()
                                      .fastDecompressor() .safeDecompressor()
lz4/LZ4Compressor.java:36,59,96-147   ``LZ4Compressor.maxCompressedLength / compress(...)``
lz4/LZ4JNICompressor.java:35-43       ``LZ4HIPCompressor`` (range checks -> native -> LZ4Exception)
lz4/LZ4SafeDecompressor.java:45-135   ``LZ4SafeDecompressor.decompress(...)``
lz4/LZ4FastDecompressor.java:48-121   ``LZ4FastDecompressor.decompress(...)``
lz4/LZ4Exception.java                 ``LZ4Exception``
util/SafeUtils.java:24-42             ``_check_range`` (ArrayIndexOutOfBounds -> IndexError,
                                      IllegalArgument -> ValueError)
xxhash/XXHashFactory.java:80,211,220  ``XXHashFactory.hipInstance().hash32() / hash64()``
xxhash/XXHash32.java:38 / XXHash64    ``XXHash32.hash(buf, off, len, seed)`` / ``XXHash64.hash``
(no equivalent: one block per call)   ``LZ4HIPBatch`` -- many independent blocks per HIP launch
====================================  =========================================================

There is no CPU implementation in this package: every codec call goes through liblz4hip.so and
fails with ``LZ4HIPError`` when the library or a GPU is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (developer A/B builds: LZ4HIP_LIBRARY names another build of the SAME library, e.g. lz4-java_amd/variants/<name>.so of tools/build_variant.sh)
_LIB_PATH = os.environ.get("LZ4HIP_LIBRARY") or os.path.join(_HERE, "liblz4hip.so")
_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)

# every symbol include/lz4hip.h declares (tests check the built library exports all of them)
C_ABI = {
    "lz4hip_init": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "lz4hip_shutdown": (None, []),
    "lz4hip_device_count": (C.c_int, []),
    "lz4hip_last_error": (C.c_char_p, []),
    "lz4hip_version": (C.c_int, []),
    "lz4hip_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "lz4hip_last_decode_route": (C.c_int, [C.c_int, C.POINTER(C.c_uint32)]),
    "lz4hip_compress_bound": (C.c_int, [C.c_int]),
    "lz4hip_compress_fast_batch": (C.c_int, [C.c_void_p, _u64p, _i32p, C.c_void_p, _u64p, _i32p, _i32p, C.c_uint32]),
    "lz4hip_compress_hc_batch": (C.c_int, [C.c_void_p, _u64p, _i32p, C.c_void_p, _u64p, _i32p, _i32p, C.c_uint32, C.c_int]),
    "lz4hip_decompress_safe_batch": (C.c_int, [C.c_void_p, _u64p, _i32p, C.c_void_p, _u64p, _i32p, _i32p, C.c_uint32]),
    "lz4hip_decompress_fast_batch": (C.c_int, [C.c_void_p, _u64p, _i32p, C.c_void_p, _u64p, _i32p, _i32p, C.c_uint32]),
    "lz4hip_xxh32_batch": (C.c_int, [C.c_void_p, _u64p, _i32p, C.c_uint32, _u32p, C.c_uint32]),
    "lz4hip_xxh64_batch": (C.c_int, [C.c_void_p, _u64p, _i32p, C.c_uint64, _u64p, C.c_uint32]),
    "lz4hip_compress_fast_batch_dev": (C.c_int, [C.c_void_p] * 7 + [C.c_uint32, C.c_int, C.c_void_p]),
    "lz4hip_compress_hc_batch_dev": (C.c_int, [C.c_void_p] * 7 + [C.c_uint32, C.c_int, C.c_int, C.c_void_p]),
    "lz4hip_hc_workspace_bytes": (C.c_size_t, [C.c_uint64, C.c_uint32, C.c_int]),
    "lz4hip_compress_hc_batch_dev_ws": (C.c_int, [C.c_void_p] * 7 + [C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]),
    "lz4hip_decompress_safe_batch_dev": (C.c_int, [C.c_void_p] * 7 + [C.c_uint32, C.c_int, C.c_void_p]),
    "lz4hip_decompress_fast_batch_dev": (C.c_int, [C.c_void_p] * 7 + [C.c_uint32, C.c_int, C.c_void_p]),
    "lz4hip_xxh32_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    "lz4hip_xxh64_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    "lz4hip_compress_fast": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "lz4hip_compress_hc": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "lz4hip_decompress_safe": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "lz4hip_decompress_fast": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "lz4hip_xxh32": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, _u32p]),
    "lz4hip_xxh64": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, _u64p]),
    "lz4hip_xxh32_stream_create": (C.c_int, [C.c_uint32, C.POINTER(C.c_void_p)]),
    "lz4hip_xxh64_stream_create": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p)]),
    "lz4hip_xxh_stream_reset": (C.c_int, [C.c_void_p, C.c_uint64]),
    "lz4hip_xxh_stream_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "lz4hip_xxh_stream_update_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "lz4hip_xxh32_stream_digest": (C.c_int, [C.c_void_p, _u32p]),
    "lz4hip_xxh64_stream_digest": (C.c_int, [C.c_void_p, _u64p]),
    "lz4hip_xxh_stream_free": (None, [C.c_void_p]),
    "lz4hip_container_workspace_bytes": (C.c_size_t, [C.c_uint64, C.c_uint32, C.c_int]),
    "lz4hip_container_blocks": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, _u64p]),
    "lz4hip_container_blocks_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                              C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "lz4hip_container_decode_workspace_bytes": (C.c_size_t, [C.c_uint32]),
    "lz4hip_container_decode_dev": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "lz4hip_container_decode": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "lz4hip_container_decode_bound": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "lz4hip_gen_blocks_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_int, C.c_void_p]),
}

_INT32_MIN = -(2 ** 31)


class LZ4Exception(Exception):
    """lz4/LZ4Exception.java -- codec-level failure (dest too small, malformed input)."""


class LZ4HIPError(RuntimeError):
    """Library-level failure: liblz4hip.so missing, no GPU, HIP error.  Never a silent fallback."""


class ReadOnlyBufferException(TypeError):
    """java.nio.ReadOnlyBufferException analogue (ByteBufferUtils.java:99-103)."""


_lib = None


def lib():
    """The loaded C-ABI library (raises LZ4HIPError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise LZ4HIPError("%s not built: run lz4-java_amd/build.sh (or __graft_entry__.build())" % _LIB_PATH)
        try:
            import torch  # noqa: F401  -- share PyTorch's HIP runtime (same SONAME) when it is there
        except Exception:
            pass
        l = C.CDLL(_LIB_PATH)
        for name, (res, args) in C_ABI.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        if hasattr(l, "lz4hip_dbg_compress_fast_profile_dev"):   # developer builds only (-DLZ4HIP_DEV_TOOLS)
            l.lz4hip_dbg_compress_fast_profile_dev.restype = C.c_int
            l.lz4hip_dbg_compress_fast_profile_dev.argtypes = [C.c_void_p] * 7 + [C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        _lib = l
    return _lib


def _chk(rc):
    if rc != 0:
        raise LZ4HIPError("liblz4hip status %d: %s" % (rc, (lib().lz4hip_last_error() or b"").decode()))


def _single(ret):
    if ret < _INT32_MIN + 64:
        raise LZ4HIPError("liblz4hip status %d: %s" % (-(ret - _INT32_MIN), (lib().lz4hip_last_error() or b"").decode()))
    return ret


def _check_length(n):  # SafeUtils.java:38-42
    if n < 0:
        raise ValueError("lengths must be >= 0")


def _check_range(buf, off, length=None):  # SafeUtils.java:24-36
    if length is None:
        if off < 0 or off >= len(buf):
            raise IndexError(off)
        return
    _check_length(length)
    if length > 0:
        _check_range(buf, off)
        _check_range(buf, off + length - 1)


def _ro_ptr(buf):
    """(address, keepalive) of a readable bytes-like object"""
    if isinstance(buf, bytes):
        return C.cast(C.c_char_p(buf), C.c_void_p).value or 0, buf
    mv = memoryview(buf)
    if mv.readonly:
        b = bytes(mv)
        return C.cast(C.c_char_p(b), C.c_void_p).value or 0, b
    arr = (C.c_uint8 * max(len(mv), 1)).from_buffer(mv) if len(mv) else (C.c_uint8 * 1)()
    return C.addressof(arr), arr


def _rw_ptr(buf):
    mv = memoryview(buf)
    if mv.readonly:
        raise ReadOnlyBufferException("dest is read-only")
    arr = (C.c_uint8 * max(len(mv), 1)).from_buffer(mv) if len(mv) else (C.c_uint8 * 1)()
    return C.addressof(arr), arr


def maxCompressedLength(length):  # LZ4Utils.java:34-41
    if length < 0:
        raise ValueError("length must be >= 0, got %d" % length)
    if length >= 0x7E000000:
        raise ValueError("length must be < 0x7E000000")
    return length + length // 255 + 16


# ----------------------------------------------------------------------------------------------
# LZ4Compressor family
# ----------------------------------------------------------------------------------------------
class LZ4Compressor:
    """lz4/LZ4Compressor.java"""

    def maxCompressedLength(self, length):
        return maxCompressedLength(length)

    def _native(self, sp, src_len, dp, max_dest_len):
        raise NotImplementedError

    def compress(self, src, srcOff=None, srcLen=None, dest=None, destOff=None, maxDestLen=None):
        """compress(src) -> bytes | compress(src, srcOff, srcLen) -> bytes |
        compress(src, srcOff, srcLen, dest, destOff[, maxDestLen]) -> int  (LZ4Compressor.java:59-147)"""
        if dest is None:
            srcOff = 0 if srcOff is None else srcOff
            srcLen = len(src) - srcOff if srcLen is None else srcLen
            out = bytearray(self.maxCompressedLength(srcLen))
            n = self.compress(src, srcOff, srcLen, out, 0, len(out))
            return bytes(out[:n])
        destOff = 0 if destOff is None else destOff
        maxDestLen = len(dest) - destOff if maxDestLen is None else maxDestLen
        dp, dk = _rw_ptr(dest)                      # checkNotReadOnly(dest)
        _check_range(src, srcOff, srcLen)           # LZ4JNICompressor.java:36-37
        _check_range(dest, destOff, maxDestLen)
        sp, sk = _ro_ptr(src)
        result = self._native(sp + srcOff, srcLen, dp + destOff, maxDestLen)
        if result <= 0:
            raise LZ4Exception("maxDestLen is too small")
        return result


class LZ4HIPCompressor(LZ4Compressor):
    """twin of lz4/LZ4JNICompressor.java: fast compressor over lz4hip_compress_fast"""

    def _native(self, sp, src_len, dp, max_dest_len):
        return _single(lib().lz4hip_compress_fast(sp, src_len, dp, max_dest_len))

    def __str__(self):
        return "LZ4HIPCompressor"


class LZ4HCHIPCompressor(LZ4Compressor):
    """twin of lz4/LZ4HCJNICompressor.java:42-51 (level clamp: LZ4Factory.java:263-270)"""

    def __init__(self, compressionLevel=9):
        self.compressionLevel = compressionLevel

    def _native(self, sp, src_len, dp, max_dest_len):
        return _single(lib().lz4hip_compress_hc(sp, src_len, dp, max_dest_len, self.compressionLevel))

    def compress(self, src, srcOff=None, srcLen=None, dest=None, destOff=None, maxDestLen=None):
        try:
            return super().compress(src, srcOff, srcLen, dest, destOff, maxDestLen)
        except LZ4Exception:
            raise LZ4Exception()  # LZ4HCJNICompressor.java:47-49 throws without a message


# ----------------------------------------------------------------------------------------------
# decompressors
# ----------------------------------------------------------------------------------------------
class LZ4SafeDecompressor:
    """lz4/LZ4SafeDecompressor.java; JNI twin LZ4JNISafeDecompressor.java:34-43"""

    def decompress(self, src, srcOff=None, srcLen=None, dest=None, destOff=None, maxDestLen=None):
        """decompress(src, maxDestLen) -> bytes | decompress(src, srcOff, srcLen, maxDestLen) -> bytes |
        decompress(src, srcOff, srcLen, dest, destOff[, maxDestLen]) -> int"""
        if dest is None or isinstance(dest, int):
            if dest is None and srcLen is None:       # decompress(src, maxDestLen)
                max_len, srcOff, srcLen = srcOff, 0, len(src)
            else:                                      # decompress(src, srcOff, srcLen, maxDestLen)
                max_len = dest
            out = bytearray(max_len)
            n = self.decompress(src, srcOff, srcLen, out, 0, max_len)
            return bytes(out[:n])
        destOff = 0 if destOff is None else destOff
        maxDestLen = len(dest) - destOff if maxDestLen is None else maxDestLen
        dp, dk = _rw_ptr(dest)
        _check_range(src, srcOff, srcLen)
        _check_range(dest, destOff, maxDestLen)
        sp, sk = _ro_ptr(src)
        result = _single(lib().lz4hip_decompress_safe(sp + srcOff, srcLen, dp + destOff, maxDestLen))
        if result < 0:
            raise LZ4Exception("Error decoding offset %d of input buffer" % (srcOff - result))
        return result


class LZ4FastDecompressor:
    """lz4/LZ4FastDecompressor.java; JNI twin LZ4JNIFastDecompressor.java:35-44"""

    def decompress(self, src, srcOff=None, dest=None, destOff=None, destLen=None):
        """decompress(src, destLen) -> bytes | decompress(src, srcOff, destLen) -> bytes |
        decompress(src, srcOff, dest, destOff, destLen) -> int (bytes read from src)"""
        if dest is None or isinstance(dest, int):
            if dest is None:                           # decompress(src, destLen)
                dest_len, srcOff = srcOff, 0
            else:                                      # decompress(src, srcOff, destLen)
                dest_len = dest
            out = bytearray(dest_len)
            self.decompress(src, srcOff, out, 0, dest_len)
            return bytes(out)
        dp, dk = _rw_ptr(dest)
        _check_range(src, srcOff) if len(src) or srcOff else None
        _check_range(dest, destOff, destLen)
        sp, sk = _ro_ptr(src)
        result = _single(lib().lz4hip_decompress_fast(sp + srcOff, len(src) - srcOff, dp + destOff, destLen))
        if result < 0:
            raise LZ4Exception("Error decoding offset %d of input buffer" % (srcOff - result))
        return result


# ----------------------------------------------------------------------------------------------
# factory
# ----------------------------------------------------------------------------------------------
class LZ4Factory:
    """lz4/LZ4Factory.java -- only the new fourth accessor exists here (the other three families are
    the reference's own and are not rebuilt)."""

    _HIP = None

    def __init__(self, impl):
        if impl != "HIP":
            raise ValueError("only the HIP family lives in this package")
        self.impl = impl
        self._fast = LZ4HIPCompressor()
        self._hc = {}
        self._fast_dec = LZ4FastDecompressor()
        self._safe_dec = LZ4SafeDecompressor()
        # LZ4Factory.java:204-220: the constructor round-trips a 20-byte vector through all members
        original = b"abcd      abcdefghij"
        for compressor in (self._fast, self.highCompressor()):
            compressed = compressor.compress(original)
            if self._fast_dec.decompress(compressed, len(original)) != original:
                raise AssertionError("fast decompressor self-test failed")
            if self._safe_dec.decompress(compressed, len(original)) != original:
                raise AssertionError("safe decompressor self-test failed")

    @classmethod
    def hipInstance(cls):
        if cls._HIP is None:
            cls._HIP = cls("HIP")
        return cls._HIP

    def fastCompressor(self):
        return self._fast

    def highCompressor(self, compressionLevel=9):
        if compressionLevel > 17:
            compressionLevel = 17
        elif compressionLevel < 1:
            compressionLevel = 9
        if compressionLevel not in self._hc:
            self._hc[compressionLevel] = LZ4HCHIPCompressor(compressionLevel)
        return self._hc[compressionLevel]

    def fastDecompressor(self):
        return self._fast_dec

    def safeDecompressor(self):
        return self._safe_dec

    def __str__(self):
        return "LZ4Factory:HIP"


# ----------------------------------------------------------------------------------------------
# xxhash
# ----------------------------------------------------------------------------------------------
class XXHash32:
    """xxhash/XXHash32.java:38; JNI twin XXHash32JNI.java:29-33"""

    def hash(self, buf, off=0, length=None, seed=0):
        length = len(buf) - off if length is None else length
        _check_range(buf, off, length)
        p, k = _ro_ptr(buf)
        out = C.c_uint32(0)
        _chk(lib().lz4hip_xxh32(p + off, length, seed & 0xFFFFFFFF, C.byref(out)))
        return out.value


class XXHash64:
    """xxhash/XXHash64.java:38; JNI twin XXHash64JNI.java:29-33"""

    def hash(self, buf, off=0, length=None, seed=0):
        length = len(buf) - off if length is None else length
        _check_range(buf, off, length)
        p, k = _ro_ptr(buf)
        out = C.c_uint64(0)
        _chk(lib().lz4hip_xxh64(p + off, length, seed & 0xFFFFFFFFFFFFFFFF, C.byref(out)))
        return out.value


class _Checksum:
    """java.util.zip.Checksum view of a streaming hash (StreamingXXHash32.java:96-126, StreamingXXHash64.java:96-126)"""

    def __init__(self, h, mask):
        self._h, self._mask = h, mask

    def getValue(self):
        return self._h.getValue() & self._mask

    def reset(self):
        self._h.reset()

    def update(self, b, off=None, length=None):
        if isinstance(b, int):
            self._h.update(bytes([b & 0xFF]), 0, 1)
        else:
            self._h.update(b, 0 if off is None else off, (len(b) if off is None else len(b) - off) if length is None else length)


class _StreamingXXHash:
    """Common part of the streaming twins: the state is a record in device memory behind a lz4hip_xxh_stream handle;
    update() continues it with one launch; getValue() may be called any number of times between updates."""

    _IS64 = False

    def __init__(self, seed):
        self.seed = seed
        self._state = C.c_void_p(None)
        if self._IS64:
            _chk(lib().lz4hip_xxh64_stream_create(seed & 0xFFFFFFFFFFFFFFFF, C.byref(self._state)))
        else:
            _chk(lib().lz4hip_xxh32_stream_create(seed & 0xFFFFFFFF, C.byref(self._state)))

    def _check_state(self):  # StreamingXXHash32JNI.java:47-51
        if not self._state:
            raise AssertionError("Already finalized")

    def reset(self):
        self._check_state()
        _chk(lib().lz4hip_xxh_stream_reset(self._state, self.seed & 0xFFFFFFFFFFFFFFFF))

    def update(self, buf, off=0, length=None):
        self._check_state()
        length = len(buf) - off if length is None else length
        _check_range(buf, off, length)
        p, k = _ro_ptr(buf)
        _chk(lib().lz4hip_xxh_stream_update(self._state, p + off, length))

    def update_device(self, dptr, length, stream=None):
        """Not in the reference: absorbs `length` bytes at DEVICE address `dptr` where they lie (asynchronous on `stream`)."""
        self._check_state()
        if length < 0:
            raise IndexError("length must be >= 0")
        _chk(lib().lz4hip_xxh_stream_update_dev(self._state, dptr, length, stream))

    def close(self):
        if self._state:
            lib().lz4hip_xxh_stream_free(self._state)
            self._state = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __repr__(self):
        return "%s(seed=%d)" % (type(self).__name__, self.seed)


class StreamingXXHash32(_StreamingXXHash):
    """xxhash/StreamingXXHash32.java:38-129; JNI twin StreamingXXHash32JNI.java:28-104 (XXH32_init/_update/_digest/_free)"""

    def getValue(self):
        self._check_state()
        out = C.c_uint32(0)
        _chk(lib().lz4hip_xxh32_stream_digest(self._state, C.byref(out)))
        return out.value

    def asChecksum(self):
        return _Checksum(self, 0xFFFFFFF)  # 28 bits: StreamingXXHash32.java:101-107


class StreamingXXHash64(_StreamingXXHash):
    """xxhash/StreamingXXHash64.java:38-129; JNI twin StreamingXXHash64JNI.java:28-104"""

    _IS64 = True

    def getValue(self):
        self._check_state()
        out = C.c_uint64(0)
        _chk(lib().lz4hip_xxh64_stream_digest(self._state, C.byref(out)))
        return out.value

    def asChecksum(self):
        return _Checksum(self, 0xFFFFFFFFFFFFFFFF)


class XXHashFactory:
    """xxhash/XXHashFactory.java: hash32() / hash64() one-shot hashes, newStreamingHash32/64(seed) streaming states"""

    _HIP = None

    @classmethod
    def hipInstance(cls):
        if cls._HIP is None:
            cls._HIP = cls()
        return cls._HIP

    def hash32(self):
        return XXHash32()

    def hash64(self):
        return XXHash64()

    def newStreamingHash32(self, seed):  # XXHashFactory.java:230-232
        return StreamingXXHash32(seed)

    def newStreamingHash64(self, seed):  # XXHashFactory.java:240-242
        return StreamingXXHash64(seed)


# ----------------------------------------------------------------------------------------------
# batch helper (host memory)
# ----------------------------------------------------------------------------------------------
def _arr(ctype, values):
    """ctypes array of the per-block offsets / lengths; a numpy array of the matching width is used in place (no copy)"""
    n = len(values)
    if hasattr(values, "ctypes") and hasattr(values, "dtype") and values.dtype.itemsize == C.sizeof(ctype) and \
            values.dtype.kind in "iu" and values.flags["C_CONTIGUOUS"] and n:
        return (ctype * n).from_address(values.ctypes.data)   # (the caller's array outlives the call)
    return (ctype * max(n, 1))(*values)


class LZ4HIPBatch:
    """Many independent blocks per launch -- the entry point the reference lacks.
    `src`/`dst` are single host buffers; block i lives at src[srcOff[i]:+srcLen[i]] and owns the
    slot dst[dstOff[i]:+dstCap[i]]."""

    @staticmethod
    def _check_ranges(buf, off, length):
        """SafeUtils.checkRange for every block: 0 <= off and off + len <= len(buf) (for decompressFast `length` is the readable
        capacity of the slot -- it bounds what the engine may read, so it must lie inside the buffer like any other range)"""
        n = len(off)
        if hasattr(off, "dtype") or hasattr(length, "dtype"):
            import numpy as np
            o, l = np.asarray(off, dtype=np.int64), np.asarray(length, dtype=np.int64)
            if n and ((l < 0).any() or (o < 0).any() or ((o + l) > len(buf)).any()):
                raise IndexError("block range outside its buffer")
        else:
            for i in range(n):
                _check_range(buf, off[i], length[i])

    @classmethod
    def _call(cls, fn, src, srcOff, srcLen, dst, dstOff, dstCap):
        n = len(srcOff)
        if not (len(srcLen) == len(dstOff) == len(dstCap) == n):
            raise ValueError("per-block arrays differ in length")
        cls._check_ranges(src, srcOff, srcLen)
        cls._check_ranges(dst, dstOff, dstCap)
        sp, sk = _ro_ptr(src)
        dp, dk = _rw_ptr(dst)
        out = (C.c_int32 * max(n, 1))()
        _chk(getattr(lib(), fn)(sp, _arr(C.c_uint64, srcOff), _arr(C.c_int32, srcLen), dp, _arr(C.c_uint64, dstOff),
                                _arr(C.c_int32, dstCap), out, n))
        if hasattr(srcOff, "dtype"):
            import numpy as np
            return np.frombuffer(out, dtype=np.int32, count=n).copy()
        return list(out[:n])

    @classmethod
    def compress(cls, src, srcOff, srcLen, dst, dstOff, dstCap):
        return cls._call("lz4hip_compress_fast_batch", src, srcOff, srcLen, dst, dstOff, dstCap)

    @classmethod
    def compressHC(cls, src, srcOff, srcLen, dst, dstOff, dstCap, level=9):
        n = len(srcOff)
        if not (len(srcLen) == len(dstOff) == len(dstCap) == n):
            raise ValueError("per-block arrays differ in length")
        cls._check_ranges(src, srcOff, srcLen)
        cls._check_ranges(dst, dstOff, dstCap)
        sp, sk = _ro_ptr(src)
        dp, dk = _rw_ptr(dst)
        out = (C.c_int32 * max(n, 1))()
        _chk(lib().lz4hip_compress_hc_batch(sp, _arr(C.c_uint64, srcOff), _arr(C.c_int32, srcLen), dp, _arr(C.c_uint64, dstOff),
                                            _arr(C.c_int32, dstCap), out, n, level))
        return list(out[:n])

    FRAME_BLOCKS, LZ4BLOCK_BLOCKS = 0, 1

    @staticmethod
    def containerBlocks(kind, data, blockSize, blockChecksum=False, level=0):
        """the data blocks of an LZ4 Frame (kind FRAME_BLOCKS) or of lz4-java's LZ4Block container (LZ4BLOCK_BLOCKS) for `data` cut
        into blockSize pieces, assembled on the device (compress, raw fallback, headers, compaction, checksums): the bytes the
        reference's LZ4FrameOutputStream.writeBlock / LZ4BlockOutputStream.flushBufferedData emit for the same blocks"""
        n = (len(data) + blockSize - 1) // blockSize
        cap = len(data) + n * (8 if kind == 0 else 21)
        dst = bytearray(max(cap, 1))
        sp, sk = _ro_ptr(data)
        dp, dk = _rw_ptr(dst)
        out = C.c_uint64(0)
        _chk(lib().lz4hip_container_blocks(kind, 1 if blockChecksum else 0, level, sp, len(data), blockSize, dp, cap, C.byref(out)))
        return bytes(dst[:out.value])

    # stop reasons of containerDecode (include/lz4hip.h)
    CR_END, CR_MORE, CR_TRUNCATED, CR_BLOCK_TOO_BIG, CR_BLOCK_CHECKSUM, CR_DECODE, CR_CORRUPT, CR_SLOTS = range(8)

    @staticmethod
    def containerDecode(kind, body, maxBlock, nMax, blockChecksum=False):
        """the READ path of the container formats on the device: the data blocks of an LZ4 Frame body / of an LZ4Block stream in `body`
        are walked, verified and decoded there (LZ4FrameInputStream.readBlock / LZ4BlockInputStream.refill for a run of blocks)
        -> (decoded bytes of the delivered blocks, [their sizes], bytes of body consumed, stop reason, liblz4 code of a failed decode)"""
        sp, sk = _ro_ptr(body)
        nb, need = C.c_uint32(0), C.c_uint64(0)   # the destination by what the body holds, not by nMax x maxBlock (round-4 advisor)
        _chk(lib().lz4hip_container_decode_bound(kind, 1 if blockChecksum else 0, sp, len(body), maxBlock, nMax, C.byref(nb), C.byref(need)))
        dst = bytearray(max(need.value, 1))
        sizes = (C.c_int32 * nMax)()
        info = (C.c_uint64 * 5)()
        dp, dk = _rw_ptr(dst)
        _chk(lib().lz4hip_container_decode(kind, 1 if blockChecksum else 0, sp, len(body), maxBlock, nMax, dp, len(dst), sizes, info))
        n_ok, consumed, why, total, code = (int(info[i]) for i in range(5))
        if code >= 1 << 63:
            code -= 1 << 64
        return bytes(dst[:total]), [int(sizes[i]) for i in range(n_ok)], consumed, why, code

    @classmethod
    def decompressSafe(cls, src, srcOff, srcLen, dst, dstOff, dstCap):
        return cls._call("lz4hip_decompress_safe_batch", src, srcOff, srcLen, dst, dstOff, dstCap)

    @classmethod
    def decompressFast(cls, src, srcOff, srcCap, dst, dstOff, dstLen):
        return cls._call("lz4hip_decompress_fast_batch", src, srcOff, srcCap, dst, dstOff, dstLen)

    @classmethod
    def xxh32(cls, buf, off, length, seed=0):
        n = len(off)
        if len(length) != n:
            raise ValueError("per-buffer arrays differ in length")
        cls._check_ranges(buf, off, length)
        p, k = _ro_ptr(buf)
        out = (C.c_uint32 * max(n, 1))()
        _chk(lib().lz4hip_xxh32_batch(p, _arr(C.c_uint64, off), _arr(C.c_int32, length), seed & 0xFFFFFFFF, out, n))
        return list(out[:n])

    @classmethod
    def xxh64(cls, buf, off, length, seed=0):
        n = len(off)
        if len(length) != n:
            raise ValueError("per-buffer arrays differ in length")
        cls._check_ranges(buf, off, length)
        p, k = _ro_ptr(buf)
        out = (C.c_uint64 * max(n, 1))()
        _chk(lib().lz4hip_xxh64_batch(p, _arr(C.c_uint64, off), _arr(C.c_int32, length), seed & 0xFFFFFFFFFFFFFFFF, out, n))
        return list(out[:n])


# ----------------------------------------------------------------------------------------------
# device-resident batches (torch tensors are only the memory/stream plumbing)
# ----------------------------------------------------------------------------------------------
class DeviceBatch:
    """Device-pointer entry points on torch CUDA(HIP) tensors: uint8 data tensors, int64 offset
    tensors (reinterpreted as uint64), int32 length/capacity/result tensors.  Launches are enqueued
    on torch's current stream of the tensors' device and do not synchronise."""

    @staticmethod
    def _stream_dev(t):
        import torch
        return t.device.index or 0, torch.cuda.current_stream(t.device).cuda_stream

    @classmethod
    def _call(cls, fn, src, src_off, src_len, dst, dst_off, dst_cap, out):
        dev, st = cls._stream_dev(src)
        _chk(getattr(lib(), fn)(src.data_ptr(), src_off.data_ptr(), src_len.data_ptr(), dst.data_ptr(), dst_off.data_ptr(),
                                dst_cap.data_ptr(), out.data_ptr(), src_off.numel(), dev, st))

    @classmethod
    def compress_fast(cls, src, src_off, src_len, dst, dst_off, dst_cap, out):
        cls._call("lz4hip_compress_fast_batch_dev", src, src_off, src_len, dst, dst_off, dst_cap, out)

    @classmethod
    def compress_hc(cls, src, src_off, src_len, dst, dst_off, dst_cap, out, level=9):
        """asynchronous: the workspace is a torch tensor sized from the source tensor (an upper bound of the batch's source span);
        the caching allocator keeps it alive until the stream has used it"""
        import torch
        dev, st = cls._stream_dev(src)
        span = src.numel() * src.element_size()     # BYTES of the source tensor (the workspace holds one u16 per source byte)
        nb = lib().lz4hip_hc_workspace_bytes(span, src_off.numel(), level)
        ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=src.device)
        _chk(lib().lz4hip_compress_hc_batch_dev_ws(src.data_ptr(), src_off.data_ptr(), src_len.data_ptr(), dst.data_ptr(),
                                                   dst_off.data_ptr(), dst_cap.data_ptr(), out.data_ptr(), src_off.numel(), level, dev, st,
                                                   span, ws.data_ptr(), nb))
        ws.record_stream(torch.cuda.current_stream(src.device))

    @classmethod
    def container_blocks(cls, kind, src, block_size, dst, total, block_checksum=False, level=0):
        """device-resident container assembly, asynchronous on the current stream: src (uint8 tensor) cut into block_size pieces ->
        the LZ4 Frame (kind 0) / LZ4Block (kind 1) data blocks in dst; total = int64 tensor[1] receiving the bytes written (more
        than dst.numel(): the result is unusable)"""
        import torch
        dev, st = cls._stream_dev(src)
        nbytes = src.numel() * src.element_size()
        wsb = lib().lz4hip_container_workspace_bytes(nbytes, block_size, level)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=src.device)
        _chk(lib().lz4hip_container_blocks_dev(kind, 1 if block_checksum else 0, level, src.data_ptr(), nbytes, block_size, dst.data_ptr(),
                                               dst.numel() * dst.element_size(), total.data_ptr(), ws.data_ptr(), wsb, dev, st))
        ws.record_stream(torch.cuda.current_stream(src.device))

    @classmethod
    def compress_hc_sync(cls, src, src_off, src_len, dst, dst_off, dst_cap, out, level=9):
        """the entry that sizes its own workspace (synchronises the stream once)"""
        dev, st = cls._stream_dev(src)
        _chk(lib().lz4hip_compress_hc_batch_dev(src.data_ptr(), src_off.data_ptr(), src_len.data_ptr(), dst.data_ptr(),
                                                dst_off.data_ptr(), dst_cap.data_ptr(), out.data_ptr(), src_off.numel(), level, dev, st))

    @classmethod
    def compress_fast_profile(cls, src, src_off, src_len, dst, dst_off, dst_cap, out, prof):
        """developer diagnostics: prof = int64 tensor [n, 12]"""
        dev, st = cls._stream_dev(src)
        _chk(lib().lz4hip_dbg_compress_fast_profile_dev(src.data_ptr(), src_off.data_ptr(), src_len.data_ptr(), dst.data_ptr(),
                                                        dst_off.data_ptr(), dst_cap.data_ptr(), out.data_ptr(), src_off.numel(),
                                                        prof.data_ptr(), dev, st))

    @classmethod
    def decompress_safe(cls, src, src_off, src_len, dst, dst_off, dst_cap, out):
        cls._call("lz4hip_decompress_safe_batch_dev", src, src_off, src_len, dst, dst_off, dst_cap, out)

    @classmethod
    def decompress_fast(cls, src, src_off, src_cap, dst, dst_off, dst_len, out):
        cls._call("lz4hip_decompress_fast_batch_dev", src, src_off, src_cap, dst, dst_off, dst_len, out)

    @classmethod
    def xxh32(cls, buf, off, length, seed, out):
        dev, st = cls._stream_dev(buf)
        _chk(lib().lz4hip_xxh32_batch_dev(buf.data_ptr(), off.data_ptr(), length.data_ptr(), seed & 0xFFFFFFFF, out.data_ptr(),
                                          off.numel(), dev, st))

    @classmethod
    def xxh64(cls, buf, off, length, seed, out):
        dev, st = cls._stream_dev(buf)
        _chk(lib().lz4hip_xxh64_batch_dev(buf.data_ptr(), off.data_ptr(), length.data_ptr(), seed & 0xFFFFFFFFFFFFFFFF,
                                          out.data_ptr(), off.numel(), dev, st))

    @classmethod
    def gen_blocks(cls, dst, stride, block_len, n_blocks, first_idx=0, seed=0x4C5A3447, litmax=38, win=65535):
        dev, st = cls._stream_dev(dst)
        _chk(lib().lz4hip_gen_blocks_dev(dst.data_ptr(), stride, block_len, seed, first_idx, litmax, win, n_blocks, dev, st))


def set_option(name, value):
    _chk(lib().lz4hip_set_option(name.encode(), value))


def last_decode_route(device=0):
    """diagnostic: (route, sampled hops, sampled stream bytes, average compressed size, sampled offsets within 6 KB, sampled sequences,
    their output bytes, 0) of the last decode launch that was routed on the device (more than 16 blocks per compute unit, every knob at
    its default); route 0 = lane-group default of the batch size, 1 = ring loop, 2 = wave loop, 3 = deep loop instead of the staged one"""
    out = (C.c_uint32 * 8)()
    _chk(lib().lz4hip_last_decode_route(device, out))
    return tuple(int(x) for x in out)
