#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: uncompressed GB/s (compress + decompress) per GPU, 64 KiB blocks.

Workload (BASELINE.json configs[1]): 65536 x 64 KiB synthetic blocks per GPU
(App. F generator, seed 0x4C5A3447, litmax 38, win 65535, fast ratio ~2.00), generated ON the device
so nothing crosses PCIe.  One "step" = LZ4 fast-compress every block of the rank's shard, gather the
compressed sizes, LZ4 safe-decompress every block (the LZ4Factory fastCompressor() +
safeDecompressor() hot path of the reference, one block per call there, one launch per batch here).
Multi-GPU (`--gpus N`, one process per GPU under torch.distributed.run): blocks are independent, so
each rank owns a contiguous range of block indices (weak scaling: 65536 blocks per GPU) and the only
collective is the all-gather of the int32 compressed sizes over RCCL.

Prints ONE JSON line on rank 0.  `value` = uncompressed bytes that went through compress+decompress
on all ranks / wall time (max over ranks) -- inputs resident in HBM.  `roofline` describes the
dominant kernel (compress); `roofline_decode` the decoder; `cpu_baseline` is the reference's own
liblz4 1.9.3 (oracle/_ref) timed on this box's host cores on a bounded sample of the same blocks.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def measured_traffic(n_blocks, block):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/traffic.json, produced by tools/traffic_passes.sh + tools/traffic_json.py): FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs, KiB -> bytes, FETCH_SIZE doubled as the gfx950 note of
    MI355X_MICROARCH.md (HBM section) prescribes.  None when no pass exists for this workload."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if t.get("blocks_per_gpu") == n_blocks and t.get("block_bytes") == block:
            return t
    except Exception:
        pass
    return None


def cpu_baseline(n_blocks, block, litmax, win):
    """oracle leg: the reference liblz4 (kind "reference") or the C port, all host cores, bounded sample"""
    from oracle import oracle as O
    O.build_port()
    odir = os.path.join(ROOT, "oracle")
    exe = os.path.join(odir, "cpu_bench")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(os.path.join(odir, "cpu_bench.c")):
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-o", exe, os.path.join(odir, "cpu_bench.c"), "-ldl", "-lpthread"])
    cores = os.cpu_count() or 1
    port_so = os.path.join(odir, "liblz4oracle.so")
    ref = O.ref_path()
    kind, lib = ("reference", ref) if ref else ("port", port_so)
    sample = min(n_blocks, 512 * cores)
    out = subprocess.check_output([exe, kind, lib, port_so, str(sample), str(block), str(cores), "3", "0", str(litmax), str(win)])
    r = json.loads(out.decode().strip().splitlines()[-1])
    return {"value": round(r["roundtrip_GBps"], 3), "unit": "GB/s", "cores": cores, "kind": kind,
            "sample": "%d x %d B blocks (same generator/seed), %d threads, best of 3; LZ4_compress_default + LZ4_decompress_safe; no JVM/JNI overhead"
                      % (sample, block, cores),
            "compress_GBps": round(r["compress_GBps"], 3), "decompress_safe_GBps": round(r["decompress_safe_GBps"], 3),
            "decompress_fast_GBps": round(r["decompress_fast_GBps"], 3), "lib": os.path.relpath(lib, ROOT) if lib.startswith(ROOT) else lib}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=65536, help="blocks per GPU")
    ap.add_argument("--block-size", type=int, default=65536)
    ap.add_argument("--litmax", type=int, default=38)
    ap.add_argument("--win", type=int, default=65535)
    ap.add_argument("--decode-lanes", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    amd = importlib.import_module("lz4-java_amd")
    shard = importlib.import_module("lz4-java_amd.shard")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.decode_lanes:
        amd.set_option("decode_lanes", args.decode_lanes)

    n, blk = args.blocks, args.block_size
    cap = amd.maxCompressedLength(blk)
    i64, i32, u8 = torch.int64, torch.int32, torch.uint8
    src = torch.empty(n * blk, dtype=u8, device=dev)
    comp = torch.empty(n * cap, dtype=u8, device=dev)
    back = torch.zeros(n * blk, dtype=u8, device=dev)
    so = torch.arange(n, dtype=i64, device=dev) * blk
    sl = torch.full((n,), blk, dtype=i32, device=dev)
    co = torch.arange(n, dtype=i64, device=dev) * cap
    cc = torch.full((n,), cap, dtype=i32, device=dev)
    clen = torch.zeros(n, dtype=i32, device=dev)
    dlen = torch.zeros(n, dtype=i32, device=dev)
    assert shard.block_range(n * world, world, rank) == (rank * n, (rank + 1) * n)
    # rank r owns block indices [r*n, (r+1)*n): contiguous ranges, no data exchange
    amd.DeviceBatch.gen_blocks(src, blk, blk, n, first_idx=rank * n, litmax=args.litmax, win=args.win)
    torch.cuda.synchronize()

    ev = lambda: torch.cuda.Event(enable_timing=True)

    def step(events=None):
        if events:
            events[0].record()
        amd.DeviceBatch.compress_fast(src, so, sl, comp, co, cc, clen)      # launched on torch's current stream
        if events:
            events[1].record()
        if world > 1:
            shard.gather_sizes(clen, n * world)                              # the ONLY collective: sizes, 4 B/block
        if events:
            events[2].record()
        amd.DeviceBatch.decompress_safe(comp, co, clen, back, so, sl, dlen)
        if events:
            events[3].record()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    evs = [[ev() for _ in range(4)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- checks (outside the timed region): the work was real ----
    ok = bool(torch.equal(back, src)) and bool((clen > 0).all()) and bool(torch.equal(dlen, sl))
    csum = int(clen.sum().item())
    t_c = sum(e[0].elapsed_time(e[1]) for e in evs) / args.steps * 1e-3   # s per compress launch (HIP events on the launch stream)
    t_d = sum(e[2].elapsed_time(e[3]) for e in evs) / args.steps * 1e-3
    if world > 1:
        okt = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())

    if rank == 0:
        nbytes = float(n) * blk
        ratio = nbytes / csum
        value = world * nbytes * args.steps / dt / 1e9
        tr = measured_traffic(n, blk) or {}
        alg_c = (nbytes + csum) / 1e9   # compress: reads N, writes C  (1 + 1/ratio B/B)
        alg_d = (csum + nbytes) / 1e9   # decompress: reads C, writes N
        out = {
            "metric": "uncompressed GB/s (compress + decompress) per GPU, 64 KiB blocks",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d x %d B blocks per GPU, SURVEY App.F gen_block(seed 0x4C5A3447, litmax %d, win %d); "
                                   "step = LZ4 fast compress + int32 size gather + LZ4 safe decompress, device-resident"
                                   % (n, blk, args.litmax, args.win),
                       "blocks_per_gpu": n, "block_bytes": blk, "ratio": round(ratio, 4), "parallelism": "blocks sharded x%d" % world},
            "verified": ok,
            "compress_GBps": round(nbytes / t_c / 1e9, 3), "decompress_GBps": round(nbytes / t_d / 1e9, 3),
            "roofline": {"kernel": "compress_fast_cu_kernel", "bound": "hbm", "achieved": round(alg_c / t_c, 3), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(alg_c / t_c / HBM_PEAK_GBPS, 5), "traffic": tr.get("compress_fast_cu_kernel"),
                         "algorithmic_bytes_per_launch": int(nbytes + csum), "avg_launch_ms": round(t_c * 1e3, 4)},
            "roofline_decode": {"kernel": "decode_kernel", "bound": "hbm", "achieved": round(alg_d / t_d, 3), "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": round(alg_d / t_d / HBM_PEAK_GBPS, 5), "traffic": tr.get("decode_kernel"),
                                "algorithmic_bytes_per_launch": int(nbytes + csum), "avg_launch_ms": round(t_d * 1e3, 4)},
        }
        if tr:
            out["traffic_source"] = tr.get("source")
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n, blk, args.litmax, args.win)
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("verification failed: decode(encode(x)) != x")


if __name__ == "__main__":
    main()
