#!/usr/bin/env python3
"""fast compress of saved inputs (gpurun_out/bad/*.bin) against the reference, with every variant library given: gpu_bad.py [variant ...]"""
import importlib, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    amd = importlib.import_module("lz4-java_amd")
    ref = O.ref()
    d = os.path.join(ROOT, "tests", "golden", "regress")
    for f in sorted(os.listdir(d)):
        v = open(os.path.join(d, f), "rb").read()
        e = ref.compress_fast(v)
        cap = ref.compress_bound(len(v)); dst = bytearray(cap)
        out = amd.LZ4HIPBatch.compress(v, [0], [len(v)], dst, [0], [cap])
        g = bytes(dst[:out[0]])
        k = next((j for j in range(min(len(g), len(e))) if g[j] != e[j]), min(len(g), len(e)))
        print("  %s: %s (sizes %d vs %d, first diff %d)" % (f, "ok" if g == e else "MISMATCH", len(g), len(e), k))
        if g != e: print("    gpu " + g[:120].hex())
        if hasattr(amd.lib(), "lz4hip_dev_asm_dbg"):
            import ctypes
            buf = (ctypes.c_uint * 1600)(); amd.lib().lz4hip_dev_asm_dbg(buf)
            for k in range(24):
                r = buf[k * 8:k * 8 + 8]
                print("    entry ip %d php %d pc %d -> code %d ip %d php %d pc %d" % (r[0], r[1] if r[1] < 1 << 31 else -1, r[2], r[3], r[4], r[5] if r[5] < 1 << 31 else -1, r[6]))
    sys.exit(0)
lib = os.path.join(ROOT, "lz4-java_amd", "liblz4hip.so")
shutil.copy(lib, "/tmp/base_keep.so")
try:
    for v in ["base"] + sys.argv[1:]:
        shutil.copy("/tmp/base_keep.so" if v == "base" else os.path.join(ROOT, "lz4-java_amd", "variants", v + ".so"), lib)
        print("== " + v, flush=True)
        subprocess.run(["timeout", "60", sys.executable, __file__, "--child"])
finally:
    shutil.copy("/tmp/base_keep.so", lib)
