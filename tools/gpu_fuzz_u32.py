#!/usr/bin/env python3
"""byU32 blocks (>= 65547 bytes: 5-byte hash, 64-bit table entries; the lean loop of csrc/lz4_fast_v2_asm32.h) against the
reference's liblz4: gpu_fuzz_u32.py <n_inputs> [seed] [max_bytes]
Inputs of 65547 .. max_bytes bytes -- synthetic (every literal / window mix), text, image and geo data repeated with noise, periodic,
low-entropy, random, far copies beyond 65535 bytes -- through the lean core (3) and the adaptive scheme (5), full and tight
capacities; the streams are decoded back.  Prints a timing of a 4 MiB batch at the end.  Exits 1 on the first mismatch."""
import importlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
amd = importlib.import_module("lz4-java_amd")


def u32_inputs(O, corpus, seed, count, max_n=600000):
    rng = random.Random(seed)
    book1, pic, geo = corpus["book1[:200000]"], corpus["pic[:65536]"], corpus["geo[:65536]"]
    out = []
    for _ in range(count):
        t = rng.randrange(9)
        n = rng.choice([65547, 65548, 65600, 70000, 131072, 200000]) if rng.random() < 0.5 else rng.randrange(65547, max_n)
        if t == 0:
            v = O.gen_block(n, rng.randrange(1 << 20), litmax=rng.choice([2, 4, 38, 200]), win=rng.choice([4, 8, 64, 300, 4096, 65535]))
        elif t == 1:   # text with noise: many short matches, far copies of the same text
            v = bytearray((book1 * (n // len(book1) + 1))[:n])
            for _ in range(rng.randrange(0, 200)): v[rng.randrange(n)] = rng.randrange(256)
            v = bytes(v)
        elif t == 2:
            v = (pic * (n // len(pic) + 1))[:n]
        elif t == 3:
            v = bytes(rng.randrange(2) for _ in range(40000)); v = (v * (n // len(v) + 1))[:n]
        elif t == 4:
            p = rng.randbytes(rng.randrange(1, 300)); v = (p * (n // len(p) + 1))[:n]
        elif t == 5:
            v = (geo * (n // len(geo) + 1))[:n]
        elif t == 6:   # copies at distances around the 65535 limit
            a = rng.randbytes(rng.randrange(200, 3000)); gap = rng.choice([65535, 65536, 65534, 65530, 66000, 60000]) - len(a)
            unit = a + rng.randbytes(max(0, gap))
            v = (unit * (n // len(unit) + 2))[:n]
        elif t == 7:
            v = rng.randbytes(n)
        else:          # synthetic blocks glued together: different statistics inside one block
            parts, m = [], 0
            while m < n:
                k = rng.randrange(1000, 90000); parts.append(O.gen_block(k, rng.randrange(1 << 20), litmax=rng.choice([2, 38, 200]), win=rng.choice([8, 300, 65535]))); m += k
            v = b"".join(parts)[:n]
        out.append(v)
    return out


def pack(blocks, caps):
    src = b"".join(blocks); so, sl, do, p, q = [], [], [], 0, 0
    for b, c in zip(blocks, caps):
        so.append(p); sl.append(len(b)); do.append(q); p += len(b); q += c
    return src, so, sl, bytearray(max(q, 1)), do


def run(n, seed, max_n, ref, corpus, cores=(3, 5), log=print):
    inputs = u32_inputs(O, corpus, seed, n, max_n)
    rng = random.Random(seed * 11 + 5)
    exp = [ref.compress_fast(v) for v in inputs]
    for core in cores:
        amd.set_option("compress_core", core)
        caps = [ref.compress_bound(len(v)) for v in inputs]
        src, so, sl, dst, do = pack(inputs, caps)
        out = amd.LZ4HIPBatch.compress(src, so, sl, dst, do, caps)
        for i, (r, o) in enumerate(zip(out, do)):
            if r != len(exp[i]) or bytes(dst[o:o + r]) != exp[i]:
                log("MISMATCH core %d input %d len %d: %d vs %d" % (core, i, len(inputs[i]), r, len(exp[i]))); return False
        caps = [max(0, len(e) + rng.choice([-2, -1, 0, 0, 1, 3])) for e in exp]
        src, so, sl, dst, do = pack(inputs, caps)
        out = amd.LZ4HIPBatch.compress(src, so, sl, dst, do, caps)
        for i, (r, o) in enumerate(zip(out, do)):
            er, eb = ref.compress_fast_raw(inputs[i], caps[i])
            if r != er or (er > 0 and bytes(dst[o:o + r]) != eb[:er]):
                log("MISMATCH tight core %d input %d len %d cap %d: %d vs %d" % (core, i, len(inputs[i]), caps[i], r, er)); return False
        log("core %d: %d byU32 inputs (%.1f MB) bit-exact, full and tight capacities" % (core, n, sum(map(len, inputs)) / 1e6))
    amd.set_option("compress_core", 5)
    caps = [len(v) for v in inputs]
    src, so, sl, dst, do = pack(exp, caps)
    out = amd.LZ4HIPBatch.decompressSafe(src, so, sl, dst, do, caps)
    for i, (r, o) in enumerate(zip(out, do)):
        if r != len(inputs[i]) or bytes(dst[o:o + r]) != inputs[i]:
            log("MISMATCH decode input %d" % i); return False
    return True


if __name__ == "__main__":
    n = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1; max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 600000
    ref = O.ref()
    corpus = {"book1[:200000]": open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read(),
              "pic[:65536]": open(os.path.join(ROOT, "tests/golden/pic_65536.bin"), "rb").read(),
              "geo[:65536]": open(os.path.join(ROOT, "tests/golden/geo_65536.bin"), "rb").read()}
    t0 = time.time()
    if not run(n, seed, max_n, ref, corpus, log=lambda s: print(s, flush=True)): sys.exit(1)
    print("byU32 fuzz ok (%.0f s)" % (time.time() - t0), flush=True)
    if os.environ.get("U32_TIMING", "1") != "0":
        import torch
        b, nb = 4 << 20, int(os.environ.get("U32_BLOCKS", "2048"))
        dev = torch.device("cuda:0")
        s = torch.empty(nb * b, dtype=torch.uint8, device=dev)
        amd.DeviceBatch.gen_blocks(s, b, b, nb, first_idx=1 << 24, litmax=38, win=4096)
        cap = amd.maxCompressedLength(b)
        c = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
        so = torch.arange(nb, dtype=torch.int64, device=dev) * b; sl = torch.full((nb,), b, dtype=torch.int32, device=dev)
        do = torch.arange(nb, dtype=torch.int64, device=dev) * cap; dc = torch.full((nb,), cap, dtype=torch.int32, device=dev)
        res = torch.empty(nb, dtype=torch.int32, device=dev)
        for core, pack in ((3, 1), (5, 1), (3, 0)):
            amd.set_option("compress_core", core); amd.set_option("compress_pack", pack)
            best = 1e9
            for _ in range(4):
                torch.cuda.synchronize(); t = time.time()
                amd.DeviceBatch.compress_fast(s, so, sl, c, do, dc, res)
                torch.cuda.synchronize(); best = min(best, time.time() - t)
            print("core %d pack %d: %d x 4 MiB (win 4096): %.2f ms, %.1f GB/s, ratio %.3f" % (core, pack, nb, best * 1e3, nb * b / best / 1e9, nb * b / float(res.sum().item())), flush=True)
        amd.set_option("compress_pack", 1); amd.set_option("compress_core", 5)
