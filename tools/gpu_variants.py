#!/usr/bin/env python3
"""times compress/decode for several builds of liblz4hip (developer tool): gpu_variants.py <n_blocks> name1 name2 ..."""
import ctypes as C, importlib, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 3:
    for name in sys.argv[2:]:
        out = subprocess.run([sys.executable, __file__, sys.argv[1], name], capture_output=True, text=True)
        print(name, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
    sys.exit(0)
name = sys.argv[2]
import shutil
src = os.path.join(ROOT, "lz4-java_amd", "liblz4hip_%s.so" % name)
shutil.copy(src, os.path.join(ROOT, "lz4-java_amd", "liblz4hip.so"))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
n = int(sys.argv[1]); dev = torch.device("cuda:0"); blk = 65536; cap = amd.maxCompressedLength(blk)
src_t = torch.empty(n * blk, dtype=torch.uint8, device=dev); amd.DeviceBatch.gen_blocks(src_t, blk, blk, n)
so = torch.arange(n, dtype=torch.int64, device=dev) * blk; sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
comp = torch.empty(n * cap, dtype=torch.uint8, device=dev); co = torch.arange(n, dtype=torch.int64, device=dev) * cap
cc = torch.full((n,), cap, dtype=torch.int32, device=dev); clen = torch.zeros(n, dtype=torch.int32, device=dev)
back = torch.zeros(n * blk, dtype=torch.uint8, device=dev); dlen = torch.zeros(n, dtype=torch.int32, device=dev)
best = [1e9, 1e9]
for _ in range(3):
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a.record(); amd.DeviceBatch.compress_fast(src_t, so, sl, comp, co, cc, clen); b.record()
    amd.DeviceBatch.decompress_safe(comp, co, clen, back, so, sl, dlen); c.record(); torch.cuda.synchronize()
    best = [min(best[0], a.elapsed_time(b)), min(best[1], b.elapsed_time(c))]
print("compress %.2f ms (%.1f GB/s) decode %.2f ms (%.1f GB/s) ok=%s" % (best[0], n * blk / best[0] / 1e6, best[1], n * blk / best[1] / 1e6, bool(torch.equal(back, src_t))))
