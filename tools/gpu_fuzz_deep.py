#!/usr/bin/env python3
"""Volume parity check of the decoder's deep interior loop on LONG streams, valid and corrupted: gpu_fuzz_deep.py <n_blocks> [seed]
n 64 KiB blocks of mixed data (App. F with several windows / literal lengths, text, binary, runs) are compressed by the reference
library; every stream is decoded as it is and in four damaged forms (flipped bytes, truncated, too small / too big a capacity) by the
safe decoder with decode_pipe 2, groups of 4 / 8 / 16 lanes; return codes and bytes against LZ4_decompress_safe of the reference."""
import importlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
amd = importlib.import_module("lz4-java_amd")
n = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ref = O.ref(); rng = random.Random(seed)
book = open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read()
geo = open(os.path.join(ROOT, "tests/golden/geo_65536.bin"), "rb").read()
pic = open(os.path.join(ROOT, "tests/golden/pic_65536.bin"), "rb").read()
def block(i):
    k = i % 8
    if k == 0: return O.gen_block(65536, seed * 100000 + i)
    if k == 1: return O.gen_block(rng.randrange(20000, 200000), seed * 100000 + i, win=4096)
    if k == 2: return O.gen_block(65536, seed * 100000 + i, litmax=rng.choice([2, 4, 70, 200]), win=rng.choice([64, 300, 65535]))
    if k == 3: o = rng.randrange(0, len(book) - 65536); return book[o:o + 65536]
    if k == 4: return geo[rng.randrange(0, 2000):]
    if k == 5: return pic[rng.randrange(0, 2000):]
    if k == 6: return bytes(rng.randrange(1000, 30000)) + rng.randbytes(rng.randrange(100, 5000)) + bytes(rng.randrange(1000, 30000))
    return (rng.randbytes(rng.randrange(3, 200)) * 2000)[:rng.randrange(30000, 100000)]
streams, caps = [], []
for i in range(n):
    v = block(i); c = ref.compress_fast(v)
    streams.append(c); caps.append(len(v))
    b = bytearray(c)
    for _ in range(rng.randrange(1, 4)): b[rng.randrange(len(b))] = rng.randrange(256)
    streams.append(bytes(b)); caps.append(len(v))
    streams.append(c[:rng.randrange(len(c) // 3, len(c))]); caps.append(len(v))
    streams.append(c); caps.append(len(v) - rng.randrange(1, 900))
    streams.append(c); caps.append(len(v) + rng.randrange(1, 64))
want = [ref.decompress_safe_raw(c, cap) for c, cap in zip(streams, caps)]
def pack(blocks, caps):
    src = b"".join(blocks); so, sl, do, p, q = [], [], [], 0, 0
    for b, c in zip(blocks, caps):
        so.append(p); sl.append(len(b)); do.append(q); p += len(b); q += c
    return src, so, sl, bytearray(max(q, 1)), do
src, so, sl, dst, do = pack(streams, caps)
# FUZZ_PIPE=3 [FUZZ_RING=<bytes>]: the ring loop (lz4_decode_ring.h) instead of the deep loop; lanes 1 / 4 / 8 / 16
pipe = int(os.environ.get("FUZZ_PIPE", "2")); ring = int(os.environ.get("FUZZ_RING", "0"))
# FUZZ_PIPE=4 [FUZZ_RING=8192|16384|32768|65536]: the wave loop (lz4_decode_wave.h), a wavefront per block
# FUZZ_PIPE=7 [FUZZ_RING=16384|32768|65536]: the pair loop (lz4_decode_pair.h), a parser and a copier wavefront per block
for lanes in ((4, 8, 16) if pipe == 2 else (64,) if pipe in (4, 5, 7, 8) else (1, 4, 8, 16)):
    amd.set_option("decode_lanes", lanes); amd.set_option("decode_pipe", pipe); amd.set_option("decode_stage", 0)
    amd.set_option("decode_ring", ring if pipe in (4, 5, 7, 8) or lanes != 1 or ring in (0, 256, 512) else 256)
    out = amd.LZ4HIPBatch.decompressSafe(src, so, sl, dst, do, caps)
    bad = 0
    for k, (r, (er, ed)) in enumerate(zip(out, want)):
        if r != er or (er >= 0 and bytes(dst[do[k]:do[k] + er]) != ed[:er]):
            print("MISMATCH lanes", lanes, "stream", k, "kind", k % 5, "len", len(streams[k]), "cap", caps[k], "got", r, "want", er); bad += 1
            if bad > 5: sys.exit(1)
    if bad: sys.exit(1)
    print("lanes %d: %d streams (%d damaged) -- return codes and bytes equal the reference's" % (lanes, len(streams), len(streams) * 4 // 5), flush=True)
print("deep fuzz ok")
