"""XXH32 / XXH64 kernels vs the oracle (XXHash32Test.java:122-142 testInstances shape: random
buffers, random offsets, random seeds -- every implementation must equal the native one)."""
import random

import pytest

pytestmark = pytest.mark.gpu


def test_known_answers(amd, golden, corpus):
    h32, h64 = amd.XXHashFactory.hipInstance().hash32(), amd.XXHashFactory.hipInstance().hash64()
    assert "%08x" % h32.hash(b"", 0, 0, 0) == "02cc5d05" and "%08x" % h32.hash(b"a", 0, 1, 0) == "550d7456"
    assert "%016x" % h64.hash(b"", 0, 0, 0) == "ef46db3751d8e999" and "%016x" % h64.hash(b"a", 0, 1, 0) == "d24ec4f1a98c6e5b"
    assert "%08x" % h32.hash(b"12345345234572", 0, 14, 0x9747b28c) == "1e34488c"      # README.md:54 example
    assert "%016x" % h64.hash(b"12345345234572", 0, 14, 0x9747b28c) == "ea6b9bde2112e286"
    for name, data in corpus.items():
        g = golden["inputs"][name]
        assert "%08x" % h32.hash(data, 0, len(data), 0) == g["xxh32_seed0"], name
        assert "%016x" % h64.hash(data, 0, len(data), 0) == g["xxh64_seed0"], name
        assert "%08x" % h32.hash(data, 0, len(data), 0x9747b28c) == g["xxh32_seed9747b28c"], name
        assert "%016x" % h64.hash(data, 0, len(data), 0x9747b28c) == g["xxh64_seed9747b28c"], name


def test_batch_random_vs_oracle(amd, ref):
    rng = random.Random(17)
    buf = rng.randbytes(1 << 20)
    off, ln = [], []
    for _ in range(3000):
        n = rng.choice([0, 1, 3, 4, 5, 15, 16, 17, 31, 32, 33, 63, 64, 100, 4096, rng.randrange(0, 20000)])
        o = rng.randrange(0, len(buf) - n + 1)
        off.append(o); ln.append(n)
    for seed in (0, 0x9747b28c, rng.getrandbits(32)):
        got = amd.LZ4HIPBatch.xxh32(buf, off, ln, seed)
        for o, n, g in zip(off, ln, got):
            assert g == ref.xxh32(buf[o:o + n], seed)
    for seed in (0, 0x9747b28c, rng.getrandbits(64)):
        got = amd.LZ4HIPBatch.xxh64(buf, off, ln, seed)
        for o, n, g in zip(off, ln, got):
            assert g == ref.xxh64(buf[o:o + n], seed)


def test_cfg5_shape_device(amd, O, ref):
    """BASELINE.json configs[4] shape: 4 KiB slices of the 64 KiB synthetic blocks, seeds 0 and 0x9747b28c"""
    import torch
    nblk, blk, sl = 64, 65536, 4096
    dev = torch.device("cuda:0")
    data = torch.empty(nblk * blk, dtype=torch.uint8, device=dev)
    amd.DeviceBatch.gen_blocks(data, blk, blk, nblk)
    n = nblk * blk // sl
    off = torch.arange(n, dtype=torch.int64, device=dev) * sl
    ln = torch.full((n,), sl, dtype=torch.int32, device=dev)
    o32 = torch.zeros(n, dtype=torch.int32, device=dev)
    o64 = torch.zeros(n, dtype=torch.int64, device=dev)
    host = data.cpu().numpy().tobytes()
    for seed in (0, 0x9747b28c):
        amd.DeviceBatch.xxh32(data, off, ln, seed, o32)
        amd.DeviceBatch.xxh64(data, off, ln, seed, o64)
        torch.cuda.synchronize()
        a, b = o32.cpu().tolist(), o64.cpu().tolist()
        for i in range(0, n, 37):
            assert a[i] & 0xFFFFFFFF == ref.xxh32(host[i * sl:(i + 1) * sl], seed)
            assert b[i] & 0xFFFFFFFFFFFFFFFF == ref.xxh64(host[i * sl:(i + 1) * sl], seed)


def test_few_long_buffers_wave_kernel(amd, ref):
    """<= 512 buffers take the wave-per-buffer XXH32 kernel (LDS-streamed, lanes 0..3 own the accumulators): lengths around the
    8192-byte switch and the 4 KiB chunk boundaries, unaligned offsets, the frame content-checksum shape (one long buffer)"""
    rng = random.Random(23)
    buf = rng.randbytes((3 << 20) + 77)
    off, ln = [], []
    for n in (0, 1, 15, 16, 8191, 8192, 8193, 12287, 12288, 12289, 16384 + 15, 65536, 65537, 100000, 1 << 20, (1 << 20) + 4095,
              (3 << 20) + 70):
        for o in (0, 1, 3, 7):
            if o + n <= len(buf):
                off.append(o); ln.append(n)
    for seed in (0, 0x9747b28c, rng.getrandbits(32)):
        got = amd.LZ4HIPBatch.xxh32(buf, off, ln, seed)
        for o, n, g in zip(off, ln, got):
            assert g == ref.xxh32(buf[o:o + n], seed), (o, n, seed)
    h32 = amd.XXHashFactory.hipInstance().hash32()
    assert h32.hash(buf, 5, len(buf) - 5, 1) == ref.xxh32(buf[5:], 1)
