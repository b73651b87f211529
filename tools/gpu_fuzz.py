#!/usr/bin/env python3
"""Long randomized parity campaign on the GPU (beyond the test-suite): gpu_fuzz.py <n_inputs> [seed]
every input through fast compress with each core (window-parallel, lean, adaptive), tight and full capacities, safe/fast decode, HC level 9 on a
subset -- all compared with the reference's own liblz4 (oracle/_ref).  Prints one line per phase; exits 1 on the first mismatch."""
import importlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
import conftest
amd = importlib.import_module("lz4-java_amd")
n = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ref = O.ref()
book1 = open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read()
corpus = {"book1[:200000]": book1, "pic[:65536]": open(os.path.join(ROOT, "tests/golden/pic_65536.bin"), "rb").read(),
          "geo[:65536]": open(os.path.join(ROOT, "tests/golden/geo_65536.bin"), "rb").read()}
inputs = conftest.rnd_inputs(O, corpus, seed, n)
rng = random.Random(seed * 7 + 1)
t0 = time.time()
exp_full = [ref.compress_fast(v) for v in inputs]
print("reference compressed %d inputs (%.1f MB) in %.1f s" % (n, sum(map(len, inputs)) / 1e6, time.time() - t0), flush=True)

def pack(blocks, caps):
    src = b"".join(blocks); so, sl, do, p, q = [], [], [], 0, 0
    for b, c in zip(blocks, caps):
        so.append(p); sl.append(len(b)); do.append(q); p += len(b); q += c
    return src, so, sl, bytearray(max(q, 1)), do

for core in (1, 3, 5):
    amd.set_option("compress_core", core)
    caps = [ref.compress_bound(len(v)) for v in inputs]
    src, so, sl, dst, do = pack(inputs, caps)
    out = amd.LZ4HIPBatch.compress(src, so, sl, dst, do, caps)
    for i, (r, o) in enumerate(zip(out, do)):
        if r != len(exp_full[i]) or bytes(dst[o:o + r]) != exp_full[i]:
            print("MISMATCH core", core, "input", i, "len", len(inputs[i]), r, len(exp_full[i])); sys.exit(1)
    # tight capacities: the return value must flip from 0 to the size exactly at the reference's threshold
    caps = [max(0, len(e) + rng.choice([-2, -1, 0, 0, 1, 3])) for e in exp_full]
    src, so, sl, dst, do = pack(inputs, caps)
    out = amd.LZ4HIPBatch.compress(src, so, sl, dst, do, caps)
    for i, (r, o) in enumerate(zip(out, do)):
        er, eb = ref.compress_fast_raw(inputs[i], caps[i])
        if r != er or (er > 0 and bytes(dst[o:o + r]) != eb[:er]):
            print("MISMATCH tight core", core, "input", i, "len", len(inputs[i]), "cap", caps[i], r, er); sys.exit(1)
    print("core %d: %d inputs bit-exact (full and tight capacities)" % (core, n), flush=True)
amd.set_option("compress_core", 5)
caps = [len(v) for v in inputs]
src, so, sl, dst, do = pack(exp_full, caps)
# every lane count x plain / pipelined / staged interior loop
for lanes, pipe, stage in ((0, -1, -1), (4, 0, 0), (4, 1, 0), (8, 1, 0), (16, 1, 0), (64, 1, 0), (64, 0, 0), (4, 0, 1), (8, 0, 1), (16, 0, 1), (64, 0, 1)):
    amd.set_option("decode_lanes", lanes); amd.set_option("decode_pipe", pipe); amd.set_option("decode_stage", stage)
    out = amd.LZ4HIPBatch.decompressSafe(src, so, sl, dst, do, caps)
    for i, (r, o) in enumerate(zip(out, do)):
        if r != len(inputs[i]) or bytes(dst[o:o + r]) != inputs[i]:
            print("MISMATCH decode_safe input", i, "lanes", lanes, "pipe", pipe, "stage", stage); sys.exit(1)
amd.set_option("decode_lanes", 0); amd.set_option("decode_pipe", -1); amd.set_option("decode_stage", -1)
for lanes, pipe, stage in ((0, -1, -1), (4, 0, 1), (8, 1, 0), (16, 0, 1)):
    amd.set_option("decode_lanes", lanes); amd.set_option("decode_pipe", pipe); amd.set_option("decode_stage", stage)
    out = amd.LZ4HIPBatch.decompressFast(src, so, sl, dst, do, caps)
    for i, (r, o) in enumerate(zip(out, do)):
        if r != len(exp_full[i]) or bytes(dst[o:o + len(inputs[i])]) != inputs[i]:
            print("MISMATCH decode_fast input", i, r, len(exp_full[i]), "lanes", lanes, "pipe", pipe, "stage", stage); sys.exit(1)
amd.set_option("decode_lanes", 0); amd.set_option("decode_pipe", -1); amd.set_option("decode_stage", -1)
print("decode safe/fast: %d streams bit-exact" % n, flush=True)
sub = inputs[: max(1, n // 8)]
caps = [ref.compress_bound(len(v)) for v in sub]
src, so, sl, dst, do = pack(sub, caps)
out = amd.LZ4HIPBatch.compressHC(src, so, sl, dst, do, caps, 9)
for i, (r, o) in enumerate(zip(out, do)):
    e = ref.compress_hc(sub[i], 9)
    if r != len(e) or bytes(dst[o:o + r]) != e:
        print("MISMATCH HC input", i, len(sub[i]), r, len(e)); sys.exit(1)
print("HC level 9: %d inputs bit-exact" % len(sub), flush=True)
sub = inputs[: max(1, n // 40)]
caps = [ref.compress_bound(len(v)) for v in sub]
src, so, sl, dst, do = pack(sub, caps)
for level in (1, 3, 6, 10, 12):
    out = amd.LZ4HIPBatch.compressHC(src, so, sl, dst, do, caps, level)
    for i, (r, o) in enumerate(zip(out, do)):
        e = ref.compress_hc(sub[i], level)
        if r != len(e) or bytes(dst[o:o + r]) != e:
            print("MISMATCH HC level", level, "input", i, len(sub[i]), r, len(e)); sys.exit(1)
print("HC levels 1, 3, 6, 10, 12: %d inputs each bit-exact" % len(sub), flush=True)
print("fuzz ok")
