#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the fast-compress step (developer diagnostics kernel).
usage: gpu_phase_profile.py <n_blocks> [data ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
amd = importlib.import_module("lz4-java_amd")
n = int(sys.argv[1]); kinds = sys.argv[2:] or ["synth", "book1"]
dev = torch.device("cuda:0"); blk = 65536; cap = amd.maxCompressedLength(blk)
if os.environ.get("CC"):
    amd.set_option("compress_core", int(os.environ["CC"]))
ms = os.environ.get("CC", "1") == "1"
v2 = os.environ.get("CC", "1") == "3"
names_v2 = ["window+hash", "table+ballot", "commit+fetch issue", "atomic result", "fetch wait+count", "catch-up+park(+flush)", "exact path (cycles/lean step)", "exact-path calls x steps"]
names_ms = ["window+hash", "bucket+fp ballot", "cand fetch issue", "emit prev window", "wait cand+verdict", "chain walk", "commit+collision", "handover+prepare"]
names = ["window+hash", "table+ballot", "issue commit/fetch", "emit prev", "atomic/collision", "wait candidate", "extend+bookkeep", "-"]
for data in kinds:
    if data == "synth":
        src = torch.empty(n * blk, dtype=torch.uint8, device=dev); amd.DeviceBatch.gen_blocks(src, blk, blk, n)
    else:
        b = {"book1": open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read()[:blk], "zeros": bytes(blk), "random": os.urandom(blk)}[data]
        src = torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(dev).repeat(n)
    so = torch.arange(n, dtype=torch.int64, device=dev) * blk; sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
    comp = torch.empty(n * cap, dtype=torch.uint8, device=dev); co = torch.arange(n, dtype=torch.int64, device=dev) * cap
    cc = torch.full((n,), cap, dtype=torch.int32, device=dev); clen = torch.zeros(n, dtype=torch.int32, device=dev)
    prof = torch.zeros((n, 12), dtype=torch.int64, device=dev)
    for _ in range(2):
        a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); amd.DeviceBatch.compress_fast_profile(src, so, sl, comp, co, cc, clen, prof); b2.record(); torch.cuda.synchronize()
    p = prof.double().mean(0).cpu().tolist()
    steps = p[0]
    print("== %s: %d blocks, kernel %.2f ms; per block: steps %.0f collision-steps %.0f false-pos %.1f sequences %.0f" % (data, n, a.elapsed_time(b2), p[0], p[1], p[2], p[3]))
    tot = sum(p[4:12])
    for i in range(8):
        print("   %-20s %8.0f cycles/step  %5.1f%%" % ((names_v2 if v2 else names_ms if ms else names)[i], p[4 + i] / steps, 100 * p[4 + i] / tot))
    print("   %-20s %8.0f cycles/step" % ("total", tot / steps))
    if v2:
        print("   exact path: %.0f calls per block, %.0f cycles per call, %.0f cycles per block; lean steps %.0f x %.0f cycles = %.0f per block"
              % (p[11], p[10] / max(p[11], 1), p[10], steps, sum(p[4:10]) / steps, sum(p[4:10])))
