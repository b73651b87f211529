"""No-GPU checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol that
include/lz4hip.h declares, and FAILS LOUDLY (no CPU fallback) when there is no device."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def header_symbols():
    h = open(os.path.join(ROOT, "include", "lz4hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    h = re.sub(r"#ifdef LZ4HIP_DEV_TOOLS.*?#endif", "", h, flags=re.S)   # developer-build-only diagnostics: not in the release library
    return sorted(set(re.findall(r"\b(lz4hip_[a-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol(amd):
    so = os.path.join(ROOT, "lz4-java_amd", "liblz4hip.so")
    assert os.path.exists(so), "run lz4-java_amd/build.sh (or __graft_entry__.build())"
    exported = set(re.findall(r" T (lz4hip_\w+)", subprocess.check_output(["nm", "-D", so]).decode()))
    declared = header_symbols()
    assert len(declared) >= 26
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert set(declared) == set(amd.C_ABI), "python binding table out of sync with the header"
    l = amd.lib()
    for s in declared:
        assert hasattr(l, s)
    assert l.lz4hip_version() == 100


def test_no_oracle_in_product():
    """the product must not reference the oracle (test infrastructure) anywhere"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lz4-java_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".inc", ".sh", ".java", ".c")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liblz4oracle" not in txt and "lz4o_" not in txt, f
    so = os.path.join(ROOT, "lz4-java_amd", "liblz4hip.so")
    deps = subprocess.check_output(["readelf", "-d", so]).decode()
    assert "lz4oracle" not in deps and "liblz4.so" not in deps and "lz4-java" not in deps


def test_compress_bound(amd, golden):
    l = amd.lib()
    for n, b in golden["compress_bound"].items():
        assert l.lz4hip_compress_bound(int(n)) == b           # LZ4Test.java:80-87 testMaxCompressedLength
    assert l.lz4hip_compress_bound(-1) == 0
    assert amd.maxCompressedLength(65536) == 65809
    with pytest.raises(ValueError):
        amd.maxCompressedLength(-1)
    with pytest.raises(ValueError):
        amd.maxCompressedLength(0x7E000000)                    # LZ4Utils.java:37-39


def test_container_decode_bound_is_a_host_walk(amd):
    """lz4hip_container_decode_bound (no device work: runs without a GPU) sizes the destination of lz4hip_container_decode by what
    the body holds -- round-4 advisor: the readers allocated nMax x maxBlock (256 MiB for a 100-byte frame; gigabytes for one LZ4Block
    header with level nibble 15).  Frame bodies: max_block per compressed block, the stored size of a raw one; LZ4Block: the headers'
    original lengths; the walk stops where the device walk stops (end mark, a block cut short, a damaged header, n_max)."""
    import ctypes as C
    import struct
    l = amd.lib()

    def bound(kind, flags, body, max_block, n_max):
        nb, need = C.c_uint32(0), C.c_uint64(0)
        buf = (C.c_uint8 * max(len(body), 1)).from_buffer_copy(body + b"\0" * (0 if body else 1))
        assert l.lz4hip_container_decode_bound(kind, flags, buf, len(body), max_block, n_max, C.byref(nb), C.byref(need)) == 0
        return nb.value, need.value
    blk = lambda size, raw=False: struct.pack("<I", size | (0x80000000 if raw else 0)) + bytes(size)
    body = blk(100) + blk(70, raw=True) + blk(5) + struct.pack("<I", 0) + b"trailing"
    assert bound(0, 0, body, 4 << 20, 64) == (3, 2 * (4 << 20) + 70)
    assert bound(0, 0, body, 4 << 20, 2) == (2, (4 << 20) + 70)            # n_max
    assert bound(0, 0, body[:150], 4 << 20, 64) == (1, 4 << 20)            # the second block is cut short
    assert bound(0, 1, blk(10) + b"CKSM" + blk(10)[:8], 65536, 64) == (1, 65536)   # block checksums: 4 more bytes per block
    assert bound(0, 0, blk(70000), 65536, 64) == (0, 0)                    # size > max_block: the walk stops (the device says why)
    assert bound(0, 0, b"", 65536, 64) == (0, 0)
    hdr = lambda method, level, clen, olen: b"LZ4Block" + bytes([method | level]) + struct.pack("<iiI", clen, olen, 0)
    bs = hdr(0x20, 6, 50, 1000) + bytes(50) + hdr(0x10, 6, 30, 30) + bytes(30) + hdr(0x10, 6, 0, 0)
    assert bound(1, 0, bs, 1 << 16, 256) == (2, 1030)                      # stops at the empty block
    assert bound(1, 0, hdr(0x20, 15, 0x7FFFFFF0, 1 << 25), 1 << 25, 256) == (0, 0)   # one header, a huge compressedLen: nothing to allocate
    assert bound(1, 0, b"LZ4Blocc" + bs[8:], 1 << 16, 256) == (0, 0)       # damaged magic
    # round-5 advisor: 5 KB of hostile headers {level nibble 15, original length 32 MiB, compressedLen 1} sized 8 GiB of device slots and
    # of destination before block 0 was found corrupt.  One compressed byte decodes to at most 255 bytes: the walk stops at such a header
    # (the device: "Stream is corrupted" without decoding) and it sizes nothing
    hostile = (hdr(0x20, 15, 1, 1 << 25) + b"\0") * 256
    assert bound(1, 0, hostile, 1 << 25, 257) == (0, 0)
    assert bound(1, 0, bs[:71] + hostile, 1 << 25, 257) == (1, 1000)
    # ... and a block bigger than the caller's max_block is where the walk stops (the device: stop reason 3, the readers' host path)
    assert bound(1, 0, bs[:71] + hdr(0x20, 15, 100, 1 << 20) + bytes(100), 1 << 16, 256) == (1, 1000)
    nb, need = C.c_uint32(0), C.c_uint64(0)
    assert l.lz4hip_container_decode_bound(2, 0, None, 0, 65536, 1, C.byref(nb), C.byref(need)) != 0      # kind
    assert l.lz4hip_container_decode_bound(0, 0, None, 0, 65536, 0, C.byref(nb), C.byref(need)) != 0      # n_max


def test_range_checks_before_native(amd):
    """argument checking order of LZ4JNICompressor.java:47-49 / SafeUtils.java:24-42 (no GPU needed:
    the checks fire before the native call)"""
    c = amd.LZ4HIPCompressor()
    with pytest.raises(IndexError):
        c.compress(b"abcdef", 2, 10, bytearray(100), 0, 100)
    with pytest.raises(ValueError):
        c.compress(b"abcdef", 0, -1, bytearray(100), 0, 100)
    with pytest.raises(IndexError):
        c.compress(b"abcdef", 0, 6, bytearray(10), 5, 10)
    with pytest.raises(amd.ReadOnlyBufferException):
        c.compress(b"abcdef", 0, 6, b"\0" * 100, 0, 100)      # LZ4Test.java:421-454
    d = amd.LZ4SafeDecompressor()
    with pytest.raises(IndexError):
        d.decompress(b"abc", 1, 5, bytearray(10), 0, 10)
    with pytest.raises(IndexError):
        amd.XXHash32().hash(b"abc", 2, 5, 0)


def test_fails_loudly_without_gpu(amd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(amd.LZ4HIPError):
        amd.LZ4HIPCompressor().compress(b"hello hello hello hello")
    with pytest.raises(amd.LZ4HIPError):
        amd.LZ4Factory.hipInstance()
    with pytest.raises(amd.LZ4HIPError):
        amd.last_decode_route()                       # (the route diagnostic reads device memory: no device, no answer)
    amd.set_option("decode_route_short", 8)           # knobs are host state: settable without a device, ranges checked
    with pytest.raises(amd.LZ4HIPError):
        amd.set_option("decode_route_short", 256)
    with pytest.raises(amd.LZ4HIPError):
        amd.set_option("decode_pipe", 6)
    for v in (7, 8, -1):
        amd.set_option("decode_pipe", v)
    assert amd.lib().lz4hip_device_count() == 0


def test_cpp_host_mirror_builds_and_jni_shim_typechecks():
    """the compiled-language mirror of the reference's host API links against the C ABI; the JNI shim
    type-checks against a stub jni.h (no JDK in this image)"""
    exe = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"),
                           "-L" + os.path.join(ROOT, "lz4-java_amd"), "-llz4hip", "-Wl,-rpath," + os.path.join(ROOT, "lz4-java_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    subprocess.check_call(["gcc", "-fsyntax-only", "-Wall", "-std=c11", "-I" + os.path.join(ROOT, "tests", "jni_stub"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "lz4-java_amd", "jni", "net_jpountz_lz4_LZ4HIPJNI.c")])
    import torch
    if not torch.cuda.is_available():
        assert subprocess.call([exe], stderr=subprocess.DEVNULL) == 3   # loud failure, no CPU path


def test_cpp_stream_mirror_builds():
    """the C++ twins of the reference's stream / container classes (lz4-java_amd/host/lz4hip_streams.hpp) compile warning-free
    and link against the C ABI; without a GPU they fail loudly like everything else"""
    exe = os.path.join(ROOT, "tests", "cpp", "stream_mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "stream_mirror_test.cpp"),
                           "-L" + os.path.join(ROOT, "lz4-java_amd"), "-llz4hip", "-Wl,-rpath," + os.path.join(ROOT, "lz4-java_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    import torch
    if not torch.cuda.is_available():
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            assert subprocess.call([exe, d], stderr=subprocess.DEVNULL) == 3


def test_java_natives_match_jni_shim():
    """every `static native` method of the Java binding classes has its Java_* function in the shim and vice versa
    (JNI name mangling: '_' -> '_1'); the factory-by-name classes XXHashFactory.instance("HIP") looks up exist."""
    import glob
    import re
    shim = open(os.path.join(ROOT, "lz4-java_amd", "jni", "net_jpountz_lz4_LZ4HIPJNI.c")).read()
    exported = set(re.findall(r"JNICALL\s+(Java_\w+)\s*\(", shim))
    declared = set()
    for path in glob.glob(os.path.join(ROOT, "lz4-java_amd", "java", "net", "jpountz", "*", "*HIPJNI.java")):
        src = open(path).read()
        pkg = re.search(r"package\s+([\w.]+);", src).group(1)
        cls = os.path.basename(path)[:-5]
        for name in re.findall(r"static\s+native\s+[\w\[\]]+\s+(\w+)\s*\(", src):
            declared.add("Java_%s_%s_%s" % (pkg.replace(".", "_"), cls, name.replace("_", "_1")))
    assert declared and declared == exported, (sorted(declared - exported), sorted(exported - declared))
    xx = os.path.join(ROOT, "lz4-java_amd", "java", "net", "jpountz", "xxhash")
    for cls in ("XXHash32HIP", "XXHash64HIP", "StreamingXXHash32HIP", "StreamingXXHash64HIP"):   # XXHashFactory.java:178-182
        assert os.path.exists(os.path.join(xx, cls + ".java")), cls
    for cls in ("StreamingXXHash32HIP", "StreamingXXHash64HIP"):
        assert "static class Factory implements" in open(os.path.join(xx, cls + ".java")).read()


def test_jni_shim_executes_against_fake_jnienv_without_device():
    """the JNI shim RUNS (no JVM needed): tests/jni_stub/fake_jni.c is a JNIEnv function table over malloc'd arrays, linked with
    the shim and liblz4hip.so.  Without a device every compute entry point must fail loudly -- library error code from the codec
    calls, RuntimeException from the hash calls -- with no staging buffer leaked and no array left pinned.  (The full scenario
    list, incl. the reference's leak path LZ4JNI.c:59-73, runs on the GPU: tests/test_gpu_jni.py.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: covered by tests/test_gpu_jni.py")
    d = os.path.join(ROOT, "tests", "jni_stub")
    subprocess.check_call(["bash", os.path.join(d, "build.sh")])
    out = subprocess.check_output([os.path.join(d, "fake_jni"), "--no-gpu"]).decode()
    assert "checks ok" in out, out


def test_batch_entry_points_validate_ranges_before_touching_memory(amd):
    """round-1 advisor finding: LZ4HIPBatch.decompressFast / xxh32 / xxh64 handed unchecked offsets to the library, which memcpy's
    from them.  Every batch entry checks 0 <= off and off + len <= len(buffer) for every block (lists and numpy arrays) first."""
    import numpy as np
    src, dst = bytes(100), bytearray(200)
    B = amd.LZ4HIPBatch
    bad = [([90], [20]), ([-1], [4]), ([0], [-4]), ([101], [0 + 1])]
    for fn in (B.compress, B.decompressSafe, B.decompressFast, B.compressHC):
        for so, sl in bad:
            with pytest.raises((IndexError, ValueError)):
                fn(src, so, sl, dst, [0], [50])
            with pytest.raises((IndexError, ValueError)):
                fn(src, np.array(so, dtype=np.int64), np.array(sl, dtype=np.int32), dst, np.array([0], dtype=np.int64), np.array([50], dtype=np.int32))
        with pytest.raises((IndexError, ValueError)):
            fn(src, [0], [10], dst, [190], [20])          # destination slot outside dst
        with pytest.raises(ValueError):
            fn(src, [0, 1], [10], dst, [0], [50])         # ragged descriptor arrays
    for fn in (B.xxh32, B.xxh64):
        for so, sl in bad:
            with pytest.raises((IndexError, ValueError)):
                fn(src, so, sl)
            with pytest.raises((IndexError, ValueError)):
                fn(src, np.array(so, dtype=np.int64), np.array(sl, dtype=np.int32))
