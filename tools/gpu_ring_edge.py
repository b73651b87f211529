#!/usr/bin/env python3
"""Volume check of the wave loops on hand-assembled streams (tests/conftest.py ring_edge_stream / wild_piece_stream: match distances at the
edge of what an output ring holds, one-sequence steps between short trips, slow copies): gpu_ring_edge.py <streams> <seed>
Every stream through decode_pipe 5 and 4 at every ring size; bytes and return codes against the reference library."""
import importlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ring_edge_stream, wild_piece_stream
from oracle import oracle as O
amd = importlib.import_module("lz4-java_amd")
ref = O.ref() if O.ref_path() else O.port()
n, seed = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed)
cases = []
for i in range(n):
    if i % 4 == 3:
        cases.append(wild_piece_stream(1 << rng.choice([13, 14, 15, 16]), rng))
    else:
        cases.append(ring_edge_stream(rng, rng.choice([20000, 60000, 90000, 150000])))
streams = [c for c, _ in cases]; caps = [m for _, m in cases]
want = [ref.decompress_safe_raw(c, m) for c, m in cases]
assert all(r == m for (r, _), m in zip(want, caps))
src = b"".join(streams)
so, do, p, q = [], [], 0, 0
for c, m in cases:
    so.append(p); do.append(q); p += len(c); q += m + 64
total = 0
try:
    variants = ((5, 0), (5, 8192), (5, 16384), (5, 32768), (5, 65536), (4, 0), (4, 8192), (4, 16384), (4, 65536))
    if os.environ.get("RING_EDGE_PIPES") == "78":   # the pair and trio loops (round 6)
        variants = ((7, 0), (7, 16384), (7, 32768), (7, 65536), (8, 0), (8, 8192), (8, 16384), (8, 32768), (8, 65536))
    for pipe, ring in variants:
        amd.set_option("decode_lanes", 64); amd.set_option("decode_pipe", pipe); amd.set_option("decode_stage", 0); amd.set_option("decode_ring", ring)
        dst = bytearray(q)
        out = amd.LZ4HIPBatch.decompressSafe(src, so, [len(c) for c in streams], dst, do, caps)
        for k, (r, (er, ed)) in enumerate(zip(out, want)):
            assert r == er and bytes(dst[do[k]:do[k] + er]) == ed[:er], ("MISMATCH", pipe, ring, k, r, er)
        total += sum(caps)
finally:
    amd.set_option("decode_lanes", 0); amd.set_option("decode_pipe", -1); amd.set_option("decode_stage", -1); amd.set_option("decode_ring", 0)
print("ring-edge streams: %d streams x 9 variants, %.1f MB decoded, all equal the reference's" % (n, total / 1e6))
