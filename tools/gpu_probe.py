#!/usr/bin/env python3
"""GPU probe: staged sanity + timing sweep (developer tool, prints as it goes)."""
import importlib, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
amd = importlib.import_module("lz4-java_amd")
from oracle import oracle as O

def log(*a):
    print(*a, flush=True)

def main():
    nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    log("device:", torch.cuda.get_device_name(0), "ref:", O.ref_path())
    chk = O.ref()
    # stage 1: tiny
    f = amd.LZ4Factory.hipInstance()
    log("factory self-test ok")
    for name, b in (("gen64k", O.gen_block(65536, 0)), ("zeros", bytes(65536)), ("empty", b""), ("g1m", O.gen_block(1 << 20, 3, win=4096))):
        t = time.time(); c = f.fastCompressor().compress(b); dt = time.time() - t
        log(name, len(b), "->", len(c), "match" if c == chk.compress_fast(b) else "MISMATCH", "%.1f ms" % (dt * 1e3))
        assert f.safeDecompressor().decompress(c, len(b)) == b
        assert f.fastDecompressor().decompress(c, len(b)) == b
    # stage 2: device batches
    dev = torch.device("cuda:0"); blk = 65536; n = nblk
    cap = amd.maxCompressedLength(blk)
    src = torch.empty(n * blk, dtype=torch.uint8, device=dev)
    t = time.time(); amd.DeviceBatch.gen_blocks(src, blk, blk, n); torch.cuda.synchronize(); log("gen %d blocks: %.1f ms" % (n, (time.time() - t) * 1e3))
    so = torch.arange(n, dtype=torch.int64, device=dev) * blk
    sl = torch.full((n,), blk, dtype=torch.int32, device=dev)
    comp = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    co = torch.arange(n, dtype=torch.int64, device=dev) * cap
    cc = torch.full((n,), cap, dtype=torch.int32, device=dev)
    clen = torch.zeros(n, dtype=torch.int32, device=dev)
    back = torch.zeros(n * blk, dtype=torch.uint8, device=dev)
    dlen = torch.zeros(n, dtype=torch.int32, device=dev)
    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        return best
    ms = timed(lambda: amd.DeviceBatch.compress_fast(src, so, sl, comp, co, cc, clen))
    csum = int(clen.sum()); ratio = n * blk / csum
    log("compress: %.3f ms  %.1f GB/s (uncompressed)  ratio %.4f" % (ms, n * blk / ms / 1e6, ratio))
    for lanes in (4, 8, 16, 32, 64):
        amd.set_option("decode_lanes", lanes)
        back.zero_()
        ms = timed(lambda: amd.DeviceBatch.decompress_safe(comp, co, clen, back, so, sl, dlen))
        ok = bool(torch.equal(back, src)) and bool(torch.equal(dlen, sl))
        log("decode_safe lanes=%2d: %.3f ms  %.1f GB/s  ok=%s" % (lanes, ms, n * blk / ms / 1e6, ok))
    amd.set_option("decode_lanes", 0)
    used = torch.zeros(n, dtype=torch.int32, device=dev)
    ms = timed(lambda: amd.DeviceBatch.decompress_fast(comp, co, cc, back, so, sl, used))
    log("decode_fast default: %.3f ms  %.1f GB/s ok=%s" % (ms, n * blk / ms / 1e6, bool(torch.equal(back, src)) and bool(torch.equal(used, clen))))
    # xxhash cfg5 shape
    nb = n * blk // 4096
    off = torch.arange(nb, dtype=torch.int64, device=dev) * 4096
    ln = torch.full((nb,), 4096, dtype=torch.int32, device=dev)
    o32 = torch.zeros(nb, dtype=torch.int32, device=dev); o64 = torch.zeros(nb, dtype=torch.int64, device=dev)
    ms = timed(lambda: amd.DeviceBatch.xxh32(src, off, ln, 0, o32)); log("xxh32 %d x 4KiB: %.3f ms %.1f GB/s" % (nb, ms, n * blk / ms / 1e6))
    ms = timed(lambda: amd.DeviceBatch.xxh64(src, off, ln, 0, o64)); log("xxh64 %d x 4KiB: %.3f ms %.1f GB/s" % (nb, ms, n * blk / ms / 1e6))
    h = src[:4096].cpu().numpy().tobytes()
    log("xxh32 ok:", (int(o32[0]) & 0xFFFFFFFF) == chk.xxh32(h, 0), "xxh64 ok:", (int(o64[0]) & 0xFFFFFFFFFFFFFFFF) == chk.xxh64(h, 0))
    # other data shapes for compress/decode: text-like (book1 slices), zeros, random
    book = open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read()
    import numpy as np
    for name, block in (("book1", book[:65536]), ("zeros", bytes(65536)), ("random", os.urandom(65536))):
        m = 4096
        hsrc = torch.from_numpy(np.frombuffer(block * 1, dtype=np.uint8).copy()).to(dev).repeat(m)
        s_o = so[:m]; s_l = sl[:m]; c_o = co[:m]; c_c = cc[:m]; cl = clen[:m]; dl = dlen[:m]
        ms = timed(lambda: amd.DeviceBatch.compress_fast(hsrc, s_o, s_l, comp, c_o, c_c, cl))
        r = m * blk / int(cl.sum())
        bk = back[:m * blk]; bk.zero_()
        ms2 = timed(lambda: amd.DeviceBatch.decompress_safe(comp, c_o, cl, bk, s_o, s_l, dl))
        log("%s x%d: compress %.1f GB/s ratio %.3f ; decode %.1f GB/s ok=%s" % (name, m, m * blk / ms / 1e6, r, m * blk / ms2 / 1e6, bool(torch.equal(bk, hsrc))))

if __name__ == "__main__":
    main()
