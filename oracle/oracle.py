"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes access to the two CPU checkers:

* ``port()``  -> oracle/liblz4oracle.so, the plain-C restatement in this directory.
* ``ref()``   -> the reference's own liblz4 1.9.3: oracle/_ref/liblz4-java.so (the prebuilt JNI
  library of /root/reference/src/resources/net/jpountz/util/linux/amd64, copied by
  ``make -C oracle ref``).  ``/root/reference`` does not exist on the GPU box; the copy under
  oracle/_ref travels with the snapshot.  If that copy is missing too, the image's system
  liblz4.so.1 (also 1.9.3, byte-identical output per SURVEY.md fact 3) is used for the LZ4
  entry points and python-xxhash for XXH32/XXH64.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (lz4-java_amd) never does.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_uint8)


def _buf(b):
    """bytes/bytearray/memoryview/numpy -> (ctypes pointer, keepalive)"""
    if isinstance(b, (bytes, bytearray)):
        arr = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) + (b"\0" if len(b) == 0 else b""))
        return C.cast(arr, _u8p), arr
    import numpy as np
    a = np.ascontiguousarray(b, dtype=np.uint8)
    return a.ctypes.data_as(_u8p), a


def build_port():
    so = os.path.join(_HERE, "liblz4oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("lz4_oracle.c", "lz4hc_oracle.c") if os.path.exists(os.path.join(_HERE, f))]
    newest = max(os.path.getmtime(s) for s in srcs + [os.path.join(_HERE, "lz4_oracle.h")])
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-shared", "-o", so] + srcs)
    return so


class _Lib:
    """Uniform API over either checker library."""

    def __init__(self, kind, lib, names, xxh=None):
        self.kind, self._l, self._xxh = kind, lib, xxh
        n = names
        self._cf = getattr(lib, n["compress_fast"]); self._cf.restype = C.c_int
        self._cf.argtypes = [_u8p, _u8p, C.c_int, C.c_int] if kind == "reference" else [_u8p, C.c_int, _u8p, C.c_int]
        self._ds = getattr(lib, n["decompress_safe"]); self._ds.restype = C.c_int
        self._ds.argtypes = [_u8p, _u8p, C.c_int, C.c_int] if kind == "reference" else [_u8p, C.c_int, _u8p, C.c_int]
        self._df = getattr(lib, n["decompress_fast"]); self._df.restype = C.c_int
        self._df.argtypes = [_u8p, _u8p, C.c_int]
        self._cb = getattr(lib, n["compress_bound"]); self._cb.restype = C.c_int; self._cb.argtypes = [C.c_int]
        self._hc = getattr(lib, n["compress_hc"], None)
        if self._hc is not None:
            self._hc.restype = C.c_int
            self._hc.argtypes = ([_u8p, _u8p, C.c_int, C.c_int, C.c_int] if kind == "reference"
                                 else [_u8p, C.c_int, _u8p, C.c_int, C.c_int])
        self._x32 = getattr(lib, n["xxh32"], None)
        self._x64 = getattr(lib, n["xxh64"], None)
        if self._x32 is not None:
            if kind == "reference":
                self._x32.restype = C.c_uint32; self._x32.argtypes = [_u8p, C.c_size_t, C.c_uint32]
                self._x64.restype = C.c_uint64; self._x64.argtypes = [_u8p, C.c_size_t, C.c_uint64]
            else:
                self._x32.restype = C.c_uint32; self._x32.argtypes = [_u8p, C.c_int64, C.c_uint32]
                self._x64.restype = C.c_uint64; self._x64.argtypes = [_u8p, C.c_int64, C.c_uint64]

    # -- LZ4 block ---------------------------------------------------------------------------
    def compress_bound(self, n):
        return self._cb(n)

    def compress_fast_raw(self, src, cap):
        """returns (ret, bytes written buffer of size cap)"""
        p, _k = _buf(src)
        out = (C.c_uint8 * max(cap, 1))()
        if self.kind == "reference":
            r = self._cf(p, C.cast(out, _u8p), len(src), cap)
        else:
            r = self._cf(p, len(src), C.cast(out, _u8p), cap)
        return r, bytes(out[: max(r, 0)])

    def compress_fast(self, src, cap=None):
        cap = self.compress_bound(len(src)) if cap is None else cap
        r, b = self.compress_fast_raw(src, cap)
        if r <= 0:
            raise ValueError("maxDestLen is too small")
        return b

    def compress_hc_raw(self, src, level, cap):
        p, _k = _buf(src)
        out = (C.c_uint8 * max(cap, 1))()
        if self.kind == "reference":
            r = self._hc(p, C.cast(out, _u8p), len(src), cap, level)
        else:
            r = self._hc(p, len(src), C.cast(out, _u8p), cap, level)
        return r, bytes(out[: max(r, 0)])

    def compress_hc(self, src, level=9, cap=None):
        cap = self.compress_bound(len(src)) if cap is None else cap
        r, b = self.compress_hc_raw(src, level, cap)
        if r <= 0:
            raise ValueError("maxDestLen is too small")
        return b

    def decompress_safe_raw(self, src, cap, src_len=None, prefill=0xA5):
        """returns (ret, dst bytes of size cap)"""
        src_len = len(src) if src_len is None else src_len
        p, _k = _buf(bytes(src) + b"\0" * 64)           # slack: liblz4 may over-read by design
        out = (C.c_uint8 * (cap + 64))(*([prefill] * 0))
        C.memset(out, prefill, cap + 64)
        if self.kind == "reference":
            r = self._ds(p, C.cast(out, _u8p), src_len, cap)
        else:
            r = self._ds(p, src_len, C.cast(out, _u8p), cap)
        return r, bytes(out[:cap])

    def decompress_safe(self, src, cap):
        r, b = self.decompress_safe_raw(src, cap)
        if r < 0:
            raise ValueError("Error decoding offset %d of input buffer" % (-r))
        return b[:r]

    def decompress_fast_raw(self, src, dst_len, prefill=0xA5):
        p, _k = _buf(bytes(src) + b"\0" * 64)
        out = (C.c_uint8 * (dst_len + 64))()
        C.memset(out, prefill, dst_len + 64)
        r = self._df(p, C.cast(out, _u8p), dst_len)
        return r, bytes(out[:dst_len])

    # -- xxhash ------------------------------------------------------------------------------
    def xxh32(self, data, seed=0):
        if self._x32 is None:
            return self._xxh.xxh32_intdigest(bytes(data), seed & 0xFFFFFFFF)
        p, _k = _buf(data)
        return self._x32(p, len(data), seed & 0xFFFFFFFF)

    def xxh64(self, data, seed=0):
        if self._x64 is None:
            return self._xxh.xxh64_intdigest(bytes(data), seed & 0xFFFFFFFFFFFFFFFF)
        p, _k = _buf(data)
        return self._x64(p, len(data), seed & 0xFFFFFFFFFFFFFFFF)


_PORT = None
_REF = None


def port():
    global _PORT
    if _PORT is None:
        lib = C.CDLL(build_port())
        _PORT = _Lib("port", lib, dict(compress_fast="lz4o_compress_fast", decompress_safe="lz4o_decompress_safe",
                                       decompress_fast="lz4o_decompress_fast", compress_bound="lz4o_compress_bound",
                                       compress_hc="lz4o_compress_hc", xxh32="lz4o_xxh32", xxh64="lz4o_xxh64"))
        lib.lz4o_gen_block.restype = None
        lib.lz4o_gen_block.argtypes = [_u8p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
        lib.lz4o_decompress_fast_bounded.restype = C.c_int
        lib.lz4o_decompress_fast_bounded.argtypes = [_u8p, C.c_int, _u8p, C.c_int]
    return _PORT


def ref_path():
    """path of the reference liblz4 1.9.3 available on THIS box (or None)"""
    for p in (os.path.join(_HERE, "_ref", "liblz4-java.so"),
              "/usr/lib/x86_64-linux-gnu/liblz4.so.1", "/opt/conda/lib/liblz4.so.1"):
        if os.path.exists(p):
            return p
    return None


def ref():
    global _REF
    if _REF is None:
        p = ref_path()
        if p is None:
            raise RuntimeError("no reference liblz4 1.9.3 on this box (run `make -C oracle ref` where /root/reference exists)")
        lib = C.CDLL(p)
        lib.LZ4_versionNumber.restype = C.c_int
        if lib.LZ4_versionNumber() != 10903:
            raise RuntimeError("%s is liblz4 %d, the pinned reference is 1.9.3 (10903)" % (p, lib.LZ4_versionNumber()))
        xxh = None
        if not hasattr(lib, "XXH32"):
            import xxhash as xxh
        _REF = _Lib("reference", lib, dict(compress_fast="LZ4_compress_default", decompress_safe="LZ4_decompress_safe",
                                           decompress_fast="LZ4_decompress_fast", compress_bound="LZ4_compressBound",
                                           compress_hc="LZ4_compress_HC", xxh32="XXH32", xxh64="XXH64"), xxh)
        _REF.path = p
    return _REF


SEED = 0x4C5A3447


def gen_block(n, idx, seed=SEED, litmax=38, win=65535):
    """SURVEY.md App. F synthetic block (C implementation in lz4_oracle.c)."""
    port()
    out = (C.c_uint8 * max(n, 1))()
    _PORT._l.lz4o_gen_block(C.cast(out, _u8p), n, seed, idx, litmax, win)
    return bytes(out[:n])


def decompress_fast_bounded(src, src_cap, dst_len):
    port()
    p, _k = _buf(bytes(src) + b"\0" * (8 + max(0, src_cap - len(src))))
    out = (C.c_uint8 * (dst_len + 8))()
    C.memset(out, 0xA5, dst_len + 8)
    r = _PORT._l.lz4o_decompress_fast_bounded(p, src_cap, C.cast(out, _u8p), dst_len)
    return r, bytes(out[:dst_len])
