"""world_size-2 (3, uneven; 8: a whole node) gloo test of the multi-GPU path: contiguous block ranges per rank and the
int32 size all-gather (lz4-java_amd/shard.py).  The per-rank codec is played by the oracle here (test
infrastructure standing in for the device launch); on GPUs bench.py drives the same functions over RCCL."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_blocks, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = importlib.import_module("lz4-java_amd.shard")
    from oracle import oracle as O
    chk = O.port()

    def codec(b0, b1):  # stand-in for DeviceBatch.compress_fast on this rank's slice
        return torch.tensor([len(chk.compress_fast(O.gen_block(4096, i))) for i in range(b0, b1)], dtype=torch.int32)

    (b0, b1), sizes = shard.compress_sharded(codec, n_blocks)
    q.put((rank, b0, b1, sizes.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_blocks", [(2, 64), (3, 50), (2, 1), (8, 16384), (8, 5)])   # (8, 16384): BASELINE configs[2] over the 8 GPUs of a node; (8, 5): ranks without a block
def test_contiguous_shards_and_size_gather(world, n_blocks):
    from oracle import oracle as O
    chk = O.port()
    expect = [len(chk.compress_fast(O.gen_block(4096, i))) for i in range(n_blocks)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_blocks, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = []
    for rank, b0, b1, sizes in sorted(res):
        assert sizes == expect            # every rank ends up with every block's size, in block order
        covered += list(range(b0, b1))
    assert covered == list(range(n_blocks))  # ranges are contiguous, disjoint and complete


def test_block_range_balance():
    shard = importlib.import_module("lz4-java_amd.shard")
    for n in (0, 1, 7, 65536, 65537):
        for w in (1, 2, 3, 8):
            r = [shard.block_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_bench_relaunch_command(monkeypatch):
    """`bench.py --gpus N` without WORLD_SIZE re-executes itself under torch.distributed.run with N ranks on 127.0.0.1 (round 1's
    flag was parsed and ignored); with WORLD_SIZE set it is a rank and must not relaunch"""
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.relaunch(4)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def test_bench_live_traffic_falls_back_without_a_gpu(monkeypatch):
    """bench.py measures the headline's HBM traffic itself with two short rocprofv3 --pmc child runs; where that cannot work -- no
    rocprofv3, a run that is itself being profiled, a pass that fails (here: no GPU) -- it returns nothing and the reason, and the
    line carries the committed passes (profiles/traffic.json) instead.  Never an exception, never a hang."""
    import argparse
    import bench
    a = argparse.Namespace(blocks=64, block_size=65536, litmax=38, win=65535, decode_lanes=0)
    got, note = bench.live_traffic(a, timeout_s=120)
    assert got == {} and isinstance(note, str) and note
    monkeypatch.setenv("ROCPROFILER_FAKE", "1")        # "this run is itself under a profiler"
    got, note = bench.live_traffic(a, timeout_s=5)
    assert got == {} and "profiler" in note
    monkeypatch.delenv("ROCPROFILER_FAKE")
    import shutil
    monkeypatch.setattr(shutil, "which", lambda *_: None)
    monkeypatch.setattr(bench.os.path, "exists", lambda p_: False if "rocprofv3" in str(p_) else os.path.lexists(p_))
    got, note = bench.live_traffic(a, timeout_s=5)
    assert got == {} and "rocprofv3" in note
    monkeypatch.undo()
    # a pass that hangs: killed with its whole process group when the time is up
    import stat, tempfile, time
    with tempfile.TemporaryDirectory() as d:
        fake = os.path.join(d, "rocprofv3")
        open(fake, "w").write("#!/bin/bash\nsleep 300 &\necho $! > %s/grandchild.pid\nwait\n" % d)
        os.chmod(fake, os.stat(fake).st_mode | stat.S_IXUSR)
        monkeypatch.setattr(shutil, "which", lambda *_: fake)
        t0 = time.time()
        got, note = bench.live_traffic(a, timeout_s=2)
        assert got == {} and "TimeoutExpired" in note and time.time() - t0 < 30
        pid = int(open(os.path.join(d, "grandchild.pid")).read())
        time.sleep(0.5)
        alive = os.path.exists("/proc/%d" % pid) and "sleep" in open("/proc/%d/cmdline" % pid).read()
        assert not alive, "the hanging pass's grandchild survived the time-out"


def test_bench_names_the_decoder_the_library_routes_to():
    """bench.py's roofline objects name the decoder instantiation by batch size (decode_kernel_name); the table must be the one of
    lz4-java_amd/csrc/kernels.hip launch_decompress / launch_decode_wave on a 256-CU device -- and every name it can return must be
    a key of profiles/traffic.json's per-kernel table when the committed passes are current (a renamed kernel would silently lose
    its `traffic`)."""
    import json
    import bench
    want = {1: "decode_trio_kernel<1, 65536, 2048, true, 1>", 256: "decode_trio_kernel<1, 65536, 2048, true, 1>",
            257: "decode_trio_kernel<2, 65536, 2048, true, 1>", 512: "decode_trio_kernel<2, 65536, 2048, true, 1>",
            1024: "decode_trio_kernel<4, 32768, 2048, true, 1>", 1280: "decode_trio_kernel<5, 16384, 2048, true, 1>",
            1281: "decode_wave_kernel<8, 16384, 2048, true, 5>", 2048: "decode_wave_kernel<8, 16384, 2048, true, 5>",
            4096: "decode_wave_kernel<16, 8192, 1024, true, 5>", 4097: "decode_deep_kernel<8, true>", 16384: "decode_deep_kernel<8, true>",
            40959: "decode_deep_kernel<8, true>", 40960: "decode_kernel<4, true, 0, true>", 65536: "decode_kernel<4, true, 0, true>"}
    for n, name in want.items():
        assert bench.decode_kernel_name(n) == name, (n, bench.decode_kernel_name(n))
    assert bench.decode_kernel_name(16384, big_blocks=True) == "decode_ring_kernel<4, 2048, true>"
    assert bench.decode_kernel_name(65536, safe=False) == "decode_kernel<4, false, 0, true>"
    src = open(os.path.join(ROOT, "lz4-java_amd", "csrc", "kernels.hip")).read()
    assert "a.n <= 5u * device_cus()" in src and "a.n <= 16u * device_cus()" in src and "a.n >= 40960u ? 4 : 8" in src and "a.n >= 16384u && a.n < 40960u" in src and "a.n <= 32u * device_cus()" in src   # the thresholds the table restates
    tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    if tr.get("kernel_source_hash") == bench.kernel_source_hash():
        for name in ("decode_kernel<4, true, 0, true>", "decode_kernel<4, false, 0, true>", "decode_ring_kernel<4, 2048, true>", "decode_wave_kernel<8, 16384, 2048, true, 5>",
                     "decode_trio_kernel<1, 65536, 2048, true, 1>", "compress_fast_v2w_cu_kernel", "compress_fast_v2wp_cu_kernel"):
            assert name in tr["kernels"], name
