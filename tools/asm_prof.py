#!/usr/bin/env python3
"""Phase profile of the hand-scheduled lean step (needs a -DLZ4HIP_V2_ASM_PROF=1 build as lz4-java_amd/variants/aprof.so):
asm_prof.py [n_blocks] [data]  -- shader cycles per step and per block, by phase."""
import ctypes, importlib, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "lz4-java_amd", "liblz4hip.so")
shutil.copy(lib, "/tmp/base_keep.so")
shutil.copy(os.path.join(ROOT, "lz4-java_amd", "variants", "aprof.so"), lib)
try:
    import torch
    amd = importlib.import_module("lz4-java_amd")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    data = sys.argv[2] if len(sys.argv) > 2 else "synth"
    dev = torch.device("cuda:0"); blk = 65536; cap = amd.maxCompressedLength(blk)
    if data == "synth":
        src = torch.empty(n * blk, dtype=torch.uint8, device=dev); amd.DeviceBatch.gen_blocks(src, blk, blk, n)
    else:
        import numpy as np
        b = open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read()[:blk] if data == "book1" else open(os.path.join(ROOT, "tests/golden/%s_65536.bin" % data), "rb").read()
        src = torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(dev).repeat(n)
    so = torch.arange(n, dtype=torch.int64, device=dev) * blk
    sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
    comp = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    co = torch.arange(n, dtype=torch.int64, device=dev) * cap
    cc = torch.full((n,), cap, dtype=torch.int32, device=dev)
    clen = torch.zeros(n, dtype=torch.int32, device=dev)
    L = amd.lib()
    out = (ctypes.c_ulonglong * 16)()
    for rep in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L.lz4hip_dev_asm_prof(out)
        a.record(); amd.DeviceBatch.compress_fast(src, so, sl, comp, co, cc, clen); b.record(); torch.cuda.synchronize()
        L.lz4hip_dev_asm_prof(out)
        v = list(out)
        ms = a.elapsed_time(b)
    steps, blocks = max(v[4], 1), max(v[8], 1)
    print("compress %.3f ms (%.1f GB/s), %d blocks" % (ms, n * blk / ms / 1e6, blocks))
    print("per asm step (cycles): window+hash+table %.0f | select+commit+issue %.0f | rows wait %.0f | count+park %.0f | sum %.0f" %
          (v[0] / steps, v[1] / steps, v[2] / steps, v[3] / steps, sum(v[:4]) / steps))
    print("per block: asm steps %.0f, code-2 exits %.0f, lean() calls %.1f" % (v[4] / blocks, v[5] / blocks, v[9] / blocks))
    print("per block (k cycles): run %.0f = asm phases %.0f + rest of lean (C++ steps, posts, exits) %.0f + outside lean (exact path, head, tail, fill) %.0f" %
          (v[7] / blocks / 1e3, sum(v[:4]) / blocks / 1e3, (v[6] - sum(v[:4])) / blocks / 1e3, (v[7] - v[6]) / blocks / 1e3))
finally:
    shutil.copy("/tmp/base_keep.so", lib)
