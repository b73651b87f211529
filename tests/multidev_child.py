"""Child process of tests/test_gpu_multidev.py: initialises liblz4hip on a device LIST WITH REPEATS ([0] * D) before any other call,
so that the C ABI's multi-device branch (csrc/api.cpp: D > 1 -- one thread + staging set per listed device, contiguous block ranges
per device) runs on a box with one GPU, then pushes a ragged multi-chunk batch through every host-pointer batch
entry point and checks every size / return code / hash and a byte sample against the reference library.  Prints 'multidev ok D=<D>'."""
import ctypes as C
import importlib
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from oracle import oracle as O  # noqa: E402

D = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5200
amd = importlib.import_module("lz4-java_amd")
L = amd.lib()
ids = (C.c_int * D)(*([0] * D))
assert L.lz4hip_init(ids, D) == 0, L.lz4hip_last_error()
assert L.lz4hip_device_count() == D
ref = O.ref()
rng = random.Random(50 + D)
base = [O.gen_block(65536, 100 + s) for s in range(24)] + [rng.randbytes(65536) for _ in range(4)] + [bytes(65536)]
lens = [rng.choice([65536, 65536, 65536, rng.randrange(0, 65537), rng.randrange(13, 2000)]) for _ in range(n)]
srcs = [base[i % len(base)][:ln] for i, ln in enumerate(lens)]
src = b"".join(srcs)
so = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
memo = {}


def clen(v):
    if v not in memo:
        memo[v] = ref.compress_fast(v)
    return memo[v]


want = [len(clen(v)) for v in srcs]
caps = [amd.maxCompressedLength(ln) if i % 7 else max(0, want[i] - 1) for i, ln in enumerate(lens)]   # every 7th: one byte short -> 0
do = np.concatenate([[0], np.cumsum(caps)[:-1]]).astype(np.uint64)
dst = bytearray(int(sum(caps)) + 1)
out = amd.LZ4HIPBatch.compress(src, so, np.array(lens, dtype=np.int32), dst, do, np.array(caps, dtype=np.int32))
for i in range(n):
    exp = want[i] if caps[i] >= want[i] else 0
    assert out[i] == exp, ("compress size", i, lens[i], caps[i], int(out[i]), exp)
# bytes: the blocks either side of every device boundary, and a random sample
edge = set()
for d in range(1, D):
    b = n * d // D
    edge.update(range(max(0, b - 3), min(n, b + 3)))
for i in sorted(edge | set(rng.sample(range(n), 400))):
    if out[i] > 0:
        assert bytes(dst[int(do[i]):int(do[i]) + int(out[i])]) == clen(srcs[i]), ("compressed bytes", i)
ok = [i for i in range(n) if out[i] > 0]
back = bytearray(len(src) + 1)
got = amd.LZ4HIPBatch.decompressSafe(dst, do[ok], np.array([out[i] for i in ok], dtype=np.int32), back, so[ok], np.array([lens[i] for i in ok], dtype=np.int32))
assert list(got) == [lens[i] for i in ok]
for i in ok:
    assert bytes(back[int(so[i]):int(so[i]) + lens[i]]) == srcs[i], ("decoded bytes", i)
back2 = bytearray(len(src) + 1)
got = amd.LZ4HIPBatch.decompressFast(dst, do[ok], np.array([caps[i] for i in ok], dtype=np.int32), back2, so[ok], np.array([lens[i] for i in ok], dtype=np.int32))
assert list(got) == [int(out[i]) for i in ok]
assert bytes(back2[:len(src)]) == bytes(back[:len(src)])
# xxhash batches through the same device list (xxh_shard)
hl = [min(ln, 4096) for ln in lens]
h32 = amd.LZ4HIPBatch.xxh32(src, so, np.array(hl, dtype=np.int32), 0x9747b28c)
h64 = amd.LZ4HIPBatch.xxh64(src, so, np.array(hl, dtype=np.int32), 12345)
for i in sorted(edge | set(rng.sample(range(n), 600))):
    v = srcs[i][:hl[i]]
    assert (h32[i] & 0xFFFFFFFF) == ref.xxh32(v, 0x9747b28c), ("xxh32", i)
    assert (h64[i] & 0xFFFFFFFFFFFFFFFF) == ref.xxh64(v, 12345), ("xxh64", i)
# HC level 9 through the device list as well (>= 64 blocks per listed device so that the batch is split)
m = 64 * D + 7
hcaps = [amd.maxCompressedLength(x) for x in lens[:m]]
hdo = np.concatenate([[0], np.cumsum(hcaps)[:-1]]).astype(np.uint64)
hdst = bytearray(int(sum(hcaps)) + 1)
hout = amd.LZ4HIPBatch.compressHC(src, so[:m], np.array(lens[:m], dtype=np.int32), hdst, hdo, np.array(hcaps, dtype=np.int32), 9)
for i in range(m):
    e = ref.compress_hc(srcs[i], 9)
    assert hout[i] == len(e) and bytes(hdst[int(hdo[i]):int(hdo[i]) + len(e)]) == e, ("HC", i)
print("multidev ok D=%d blocks=%d" % (D, n))
