#!/usr/bin/env python3
"""First mismatches of fast compress against the reference on the suite's fuzz inputs, with where they differ: gpu_fuzz_dbg.py [seed] [count]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
import conftest
amd = importlib.import_module("lz4-java_amd")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 41
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
ref = O.ref()
corpus = {"book1[:200000]": open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read(),
          "pic[:65536]": open(os.path.join(ROOT, "tests/golden/pic_65536.bin"), "rb").read(),
          "geo[:65536]": open(os.path.join(ROOT, "tests/golden/geo_65536.bin"), "rb").read()}
inputs = conftest.rnd_inputs(O, corpus, seed, count)
caps = [ref.compress_bound(len(v)) for v in inputs]
src = b"".join(inputs); so, sl, do, p, q = [], [], [], 0, 0
for b, c in zip(inputs, caps):
    so.append(p); sl.append(len(b)); do.append(q); p += len(b); q += c
dst = bytearray(max(q, 1))
out = amd.LZ4HIPBatch.compress(src, so, sl, dst, do, caps)
bad = 0
for i, (r, o) in enumerate(zip(out, do)):
    e = ref.compress_fast(inputs[i]); g = bytes(dst[o:o + max(r, 0)])
    if g != e:
        k = next((j for j in range(min(len(g), len(e))) if g[j] != e[j]), min(len(g), len(e)))
        # input position of the first differing sequence: decode the common prefix
        try:
            pos = len(ref.decompress_safe_partial(e[:k], len(inputs[i]))) if hasattr(ref, "decompress_safe_partial") else -1
        except Exception:
            pos = -1
        print("MISMATCH input %d len %d: sizes %d vs %d, first diff at output byte %d (%s)" % (i, len(inputs[i]), r, len(e), k, inputs[i][:16].hex()))
        open("/tmp/bad_%d.bin" % i, "wb").write(inputs[i])
        bad += 1
        if bad >= 8: break
print("checked %d inputs, %d mismatches" % (len(inputs), bad))
if bad:
    import shutil
    os.makedirs(os.path.join(ROOT, "gpurun_out", "bad"), exist_ok=True)
    for f in os.listdir("/tmp"):
        if f.startswith("bad_"): shutil.copy("/tmp/" + f, os.path.join(ROOT, "gpurun_out", "bad", f))
