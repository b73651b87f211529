#!/usr/bin/env python3
"""BASELINE.json configs[4] shape: XXH32 / XXH64 of 1 Mi x 4 KiB buffers (device-resident), both batch kernels, checked against the
reference library on a sample.  usage: xxh_cfg5.py [n_buffers=1048576]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda:0"); blk = 65536; n = nb * 4096 // blk
src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
amd.DeviceBatch.gen_blocks(src, blk, blk, n)
off = torch.arange(nb, dtype=torch.int64, device=dev) * 4096
ln = torch.full((nb,), 4096, dtype=torch.int32, device=dev)
o32 = torch.zeros(nb, dtype=torch.int32, device=dev); o64 = torch.zeros(nb, dtype=torch.int64, device=dev)
ref = None
try:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    ref = O.ref()
except Exception as e:
    print("reference check skipped: %r" % (e,))
host = src[: 64 * 4096].cpu().numpy().tobytes()
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for kern in (1,):
    for seed in (0, 0x9747b28c):
        m32 = timed(lambda: amd.DeviceBatch.xxh32(src, off, ln, seed, o32))
        m64 = timed(lambda: amd.DeviceBatch.xxh64(src, off, ln, seed, o64))
        ok = "-"
        if ref:
            a, b = o32[:64].cpu().tolist(), o64[:64].cpu().tolist()
            ok = all((a[i] & 0xFFFFFFFF) == ref.xxh32(host[i * 4096:(i + 1) * 4096], seed) and (b[i] & 0xFFFFFFFFFFFFFFFF) == ref.xxh64(host[i * 4096:(i + 1) * 4096], seed) for i in range(64))
        print("xxh_kernel=%d seed=%#x: XXH32 %.3f ms = %.2f TB/s, XXH64 %.3f ms = %.2f TB/s  ok=%s" % (kern, seed, m32, nb * 4096 / m32 / 1e9, m64, nb * 4096 / m64 / 1e9, ok), flush=True)
