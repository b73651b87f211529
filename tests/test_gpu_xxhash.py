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
            assert g == ref.xxh32(buf[o:o + n], seed), (o, n)
    for seed in (0, 0x9747b28c, rng.getrandbits(64)):
        got = amd.LZ4HIPBatch.xxh64(buf, off, ln, seed)
        for o, n, g in zip(off, ln, got):
            assert g == ref.xxh64(buf[o:o + n], seed), (o, n)


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


def test_streaming_equals_one_shot(amd, ref):
    """XXHash32Test / XXHash64Test streaming cases: any split of the input into update() calls gives the one-shot
    hash of the whole; getValue() between updates is the hash of the prefix; reset() starts over with the same seed."""
    rng = random.Random(23)
    f = amd.XXHashFactory.hipInstance()
    data = rng.randbytes(300000)
    for seed32, seed64 in ((0, 0), (0x9747b28c, 0x9747b28c), (rng.getrandbits(32), rng.getrandbits(64))):
        with f.newStreamingHash32(seed32) as h32, f.newStreamingHash64(seed64) as h64:
            assert h32.getValue() == ref.xxh32(b"", seed32) and h64.getValue() == ref.xxh64(b"", seed64)
            for trial in range(6):
                n = rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 100, 4096 + 7, 8192 * 3 + 5, len(data)])
                pos = 0
                while pos < n:
                    step = min(n - pos, rng.choice([1, 2, 3, 5, 11, 15, 16, 17, 31, 32, 33, 63, 64, 1000, 4096, 8192, 50000]))
                    h32.update(data, pos, step)
                    h64.update(data, pos, step)
                    pos += step
                    if rng.random() < 0.3:
                        assert h32.getValue() == ref.xxh32(data[:pos], seed32), (n, pos)
                        assert h64.getValue() == ref.xxh64(data[:pos], seed64), (n, pos)
                assert h32.getValue() == ref.xxh32(data[:n], seed32), n
                assert h64.getValue() == ref.xxh64(data[:n], seed64), n
                assert h32.asChecksum().getValue() == ref.xxh32(data[:n], seed32) & 0xFFFFFFF
                h32.reset()
                h64.reset()
        with pytest.raises(AssertionError):
            h32.getValue()  # closed: "Already finalized" (StreamingXXHash32JNI.java:47-51)
    with f.newStreamingHash32(1) as h:
        with pytest.raises(IndexError):
            h.update(b"abc", 2, 5)


def test_streaming_device_updates_and_long_xxh64(amd, ref):
    """update_device hashes device-resident bytes where they lie; a long single XXH64 buffer takes the wave kernel"""
    import torch
    rng = random.Random(29)
    data = rng.randbytes((1 << 20) + 13)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    f = amd.XXHashFactory.hipInstance()
    with f.newStreamingHash32(7) as h32, f.newStreamingHash64(7) as h64:
        pos = 0
        for step in (5, 4096, 100000, 31, len(data)):
            step = min(step, len(data) - pos)
            h32.update_device(t.data_ptr() + pos, step)
            h64.update_device(t.data_ptr() + pos, step)
            pos += step
        assert pos == len(data)
        assert h32.getValue() == ref.xxh32(data, 7) and h64.getValue() == ref.xxh64(data, 7)
    assert f.hash64().hash(data, 0, len(data), 99) == ref.xxh64(data, 99)
    assert f.hash64().hash(data, 3, 8192, 99) == ref.xxh64(data[3:3 + 8192], 99)
    got = amd.LZ4HIPBatch.xxh64(data, [0, 1, 100], [len(data), 70000, 8191], 5)
    assert list(got) == [ref.xxh64(data, 5), ref.xxh64(data[1:70001], 5), ref.xxh64(data[100:100 + 8191], 5)]


def test_streaming_more_than_4GiB(amd):
    """XXHash32Test.java:144-165 / XXHash64Test test4GB: a stream longer than 2^32 bytes (the 32-bit length wraps inside XXH32;
    the state keeps 64 bits).  17 x 256 MiB device-resident updates, checked against the python xxhash package's streaming state."""
    import torch
    xxhash = pytest.importorskip("xxhash")
    chunk = 256 << 20
    t = torch.randint(0, 256, (chunk,), dtype=torch.uint8, device="cuda:0")
    host = t.cpu().numpy().tobytes()
    f = amd.XXHashFactory.hipInstance()
    e32, e64 = xxhash.xxh32(seed=0x9747b28c), xxhash.xxh64(seed=0x9747b28c)
    with f.newStreamingHash32(0x9747b28c) as h32, f.newStreamingHash64(0x9747b28c) as h64:
        for i in range(17):
            h32.update_device(t.data_ptr(), chunk)
            h64.update_device(t.data_ptr(), chunk)
            e32.update(host); e64.update(host)
            if i in (0, 15, 16):   # below, at and past 2^32 bytes
                assert h32.getValue() == e32.intdigest(), i
                assert h64.getValue() == e64.intdigest(), i
