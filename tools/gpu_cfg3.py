#!/usr/bin/env python3
"""BASELINE.json configs[2] shape: safe decompress of 4 MiB blocks (gen_block(4 MiB, win=4096)).
usage: gpu_cfg3.py [n_blocks=4096] [reps=4] [lanes ...]   -- 4 reps over a 4096-block working set = 16384 blocks = 64 GiB of output.
Blocks are generated and fast-compressed on the device (a sample is checked against the reference library when it is there)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lanes_list = [int(x) for x in sys.argv[3:]] or [0]
blk = 4 << 20
dev = torch.device("cuda:0"); cap = amd.maxCompressedLength(blk)
src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
amd.DeviceBatch.gen_blocks(src, blk, blk, n, win=4096)
so = torch.arange(n, dtype=torch.int64, device=dev) * blk
sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
comp = torch.empty(n * cap, dtype=torch.uint8, device=dev)
co = torch.arange(n, dtype=torch.int64, device=dev) * cap
cc = torch.full((n,), cap, dtype=torch.int32, device=dev)
clen = torch.zeros(n, dtype=torch.int32, device=dev)
back = torch.zeros(n * blk, dtype=torch.uint8, device=dev)
dlen = torch.zeros(n, dtype=torch.int32, device=dev)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
amd.DeviceBatch.compress_fast(src, so, sl, comp, co, cc, clen); torch.cuda.synchronize()
a.record(); amd.DeviceBatch.compress_fast(src, so, sl, comp, co, cc, clen); b.record(); torch.cuda.synchronize()
csum = int(clen.sum().item())
print("fast compress %d x 4 MiB: %.1f ms (%.1f GB/s), ratio %.3f" % (n, a.elapsed_time(b), n * blk / a.elapsed_time(b) / 1e6, n * blk / csum), flush=True)
try:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    ref = O.ref()
    h = src[:blk].cpu().numpy().tobytes()
    got = comp[:int(clen[0])].cpu().numpy().tobytes()
    print("block 0 vs reference library: %s" % ("bit-exact" if got == ref.compress_fast(h) else "MISMATCH"), flush=True)
except Exception as e:  # the reference library is test infrastructure; its absence only skips this check
    print("reference check skipped: %r" % (e,))
if os.environ.get("DS"):
    amd.set_option("decode_stage", int(os.environ["DS"]))
if os.environ.get("DP"):
    amd.set_option("decode_pipe", int(os.environ["DP"]))
for lanes in lanes_list:
    amd.set_option("decode_lanes", lanes)
    amd.DeviceBatch.decompress_safe(comp, co, clen, back, so, sl, dlen); torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        amd.DeviceBatch.decompress_safe(comp, co, clen, back, so, sl, dlen)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    ok = bool(torch.equal(back, src)) and bool(torch.equal(dlen, sl))
    print("safe decompress, lanes %s: %d x 4 MiB in %.1f ms = %.1f GB/s of output (%.1f GiB)  ok=%s"
          % (lanes or "default", n * reps, ms, n * reps * blk / ms / 1e6, n * reps * blk / 2**30, ok), flush=True)
