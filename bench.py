#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: uncompressed GB/s (compress + decompress) per GPU, 64 KiB blocks.

Headline workload (BASELINE.json configs[1]): 65536 x 64 KiB synthetic blocks per GPU
(App. F generator, seed 0x4C5A3447, litmax 38, win 65535, fast ratio ~2.00), generated ON the device
so nothing crosses PCIe.  One "step" = LZ4 fast-compress every block of the rank's shard, gather the
compressed sizes, LZ4 safe-decompress every block (the LZ4Factory fastCompressor() +
safeDecompressor() hot path of the reference, one block per call there, one launch per batch here).

Multi-GPU: `--gpus N` runs N ranks, one process per GPU.  Under torch.distributed.run (WORLD_SIZE set) the
script is a rank; started plainly with N > 1 it re-executes itself under `python -m torch.distributed.run
--nproc-per-node N` (127.0.0.1).  Blocks are independent, so each rank owns a contiguous range of block
indices (weak scaling: 65536 blocks per GPU) and the only collective is the all-gather of the int32
compressed sizes over RCCL.  `n_gpus` in the line is the number of ranks that actually joined the nccl
group; a run that asked for N and got fewer fails.

Prints ONE JSON line on rank 0.  `value` = uncompressed bytes that went through compress+decompress
on all ranks / wall time (max over ranks) -- inputs resident in HBM.  `roofline` describes the dominant
kernel (fast compress); `roofline_decode` the decoder; `cpu_baseline` is the reference's own liblz4 1.9.3
(oracle/_ref) timed on this box's host cores on a bounded sample of the same blocks.  `configs` carries the
other BASELINE.json workloads, each timed outside the headline region with its own roofline and cpu_baseline:
decompress_fast on the headline blocks, configs[2] (safe decode of 4 MiB blocks) and the byU32 fast compress of the same blocks,
configs[3] (HC level 9 of 1 MiB blocks), configs[4] (XXH32 / XXH64 of 4 KiB buffers), real text (65536 slices of Calgary book1:
the stand-in for configs[0]'s Silesia/dickens block, compressed bytes checked against the reference library), and
`end_to_end`: the host-pointer batch API on the headline blocks from pageable host memory, PCIe included (never `value`).
Every cpu_baseline carries best-of-N and the median over the N runs, for all host threads and for one thread.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def kernel_source_hash():
    """sha256 over the kernel sources: profiles/traffic.json records the hash it was measured at, so a traffic figure that
    predates a kernel change is reported as stale instead of silently quoted"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "lz4-java_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


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


def live_traffic(args, timeout_s=300):
    """HBM bytes per launch of the headline's two kernels, MEASURED BY THIS RUN: two short child runs of this same file (headline launches
    only, --warmup 1 --steps 2) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, KiB -> bytes,
    FETCH_SIZE doubled: MI355X_MICROARCH.md's HBM section -- before the parent touches the GPU.  Returns ({kernel: bytes}, note); an empty
    dict and the reason when rocprofv3 is missing, this process is itself being profiled, or a pass fails: the committed passes
    (profiles/traffic.json) are what the line carries then."""
    import shutil, sqlite3, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return {}, "no rocprofv3 on this box"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return {}, "this run is itself under a profiler"
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="lz4hip_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-configs", "--no-live-traffic",
                   "--blocks", str(args.blocks), "--block-size", str(args.block_size), "--litmax", str(args.litmax), "--win", str(args.win),
                   "--decode-lanes", str(args.decode_lanes)]
            # (its own process group: a pass that runs into the time-out is killed WITH the bench child rocprofv3 started -- nothing of it
            # may still be on the GPU when the parent measures)
            proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                raise
            r = proc
            dbs = [os.path.join(w, f) for w, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not dbs:
                return {}, "the %s pass failed (rc %d)" % (counter, r.returncode)
            per = {}
            con = sqlite3.connect(dbs[0])
            for k, c, v in con.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%lz4hip%'"):
                if c == counter:
                    per.setdefault(k.split("(")[0].replace("void ", "").replace("lz4hip::", "").strip(), []).append(v)
            con.close()
            for k, vs in per.items():
                vs = sorted(vs)
                vals.setdefault(k, {})[counter] = vs[len(vs) // 2] if len(vs) % 2 else 0.5 * (vs[len(vs) // 2 - 1] + vs[len(vs) // 2])
        except Exception as e:   # a time-out, an unreadable database: the committed passes stand in
            return {}, "the %s pass: %s" % (counter, type(e).__name__)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {k: int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024) for k, c in vals.items() if "FETCH_SIZE" in c and "WRITE_SIZE" in c}
    return out, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes made by this run (two child runs of the headline launches, median over their launches)"


def kernel_traffic(tr, world, *names):
    """HBM bytes per launch of the named kernels (summed) from the same PMC passes; None unless every one was measured, at N = 1"""
    k = (tr or {}).get("kernels") or {}
    if world != 1 or not names or any(n not in k for n in names):
        return None
    return int(sum(k[n] for n in names))


def text_traffic(tr, world, n_blocks, leg, what):
    """HBM bytes per launch of a real-text leg (profiles/traffic.json "text": PMC passes of tools/gpu_text_legs.py, which runs those
    legs alone); None unless measured, at N = 1 and the bench's default batch"""
    if world != 1 or n_blocks != 65536:
        return None
    return (((tr or {}).get("text") or {}).get(leg) or {}).get(what)


def cpu_bench(args, env=None):
    """runs oracle/cpu_bench (the reference's liblz4 through dlopen, or the C port) and returns its JSON line"""
    from oracle import oracle as O
    O.build_port()
    odir = os.path.join(ROOT, "oracle")
    exe = os.path.join(odir, "cpu_bench")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(os.path.join(odir, "cpu_bench.c")):
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-o", exe, os.path.join(odir, "cpu_bench.c"), "-ldl", "-lpthread"])
    port_so = os.path.join(odir, "liblz4oracle.so")
    ref = O.ref_path()
    kind, lib = ("reference", ref) if ref else ("port", port_so)
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.check_output([exe, kind, lib, port_so] + [str(a) for a in args], env=e)
    r = json.loads(out.decode().strip().splitlines()[-1])
    r["_kind"], r["_lib"] = kind, (os.path.relpath(lib, ROOT) if lib.startswith(ROOT) else lib)
    return r


def decode_kernel_name(n_blocks, safe=True, big_blocks=False):
    """the decoder instantiation launch_decompress (kernels.hip) picks: by batch size, and -- batches of 16384 .. 40959 blocks -- on the
    device by the blocks' compressed sizes (decode_route_kernel: blocks of >= 512 KiB go to the ring loop, csrc/lz4_decode_ring.h)"""
    s = "true" if safe else "false"
    if n_blocks <= 5 * 256:    # up to 5 blocks per CU: scanner, planner and copier wavefronts per block (csrc/lz4_decode_trio.h)
        w, kw = (1, 65536) if n_blocks <= 256 else (2, 65536) if n_blocks <= 512 else (4, 32768) if n_blocks <= 1024 else (5, 16384)
        return "decode_trio_kernel<%d, %d, 2048, %s, 1>" % (w, kw, s)
    if n_blocks <= 16 * 256:   # up to 16 blocks per CU: a wavefront per block, several sequences per trip (csrc/lz4_decode_wave.h)
        w, kw, ks = (8, 16384, 2048) if n_blocks <= 2048 else (16, 8192, 1024)
        return "decode_wave_kernel<%d, %d, %d, %s, 5>" % (w, kw, ks, s)
    if n_blocks >= 40960:
        return "decode_kernel<4, %s, 0, true>" % s
    if 16384 <= n_blocks < 40960 and big_blocks:
        return "decode_ring_kernel<4, 2048, %s>" % s
    return "decode_deep_kernel<8, %s>" % s   # the deep interior loop, csrc/lz4_decode_deep.h


def routed_decoder(amd, dev_index, n_blocks, safe=True):
    """the decoder a launch of more than 16 blocks per CU really ran: decode_route_kernel's decision for the device's last routed launch
    (lz4hip_last_decode_route: 0 = the lane-group default of the batch size, 1 = ring loop, 2 = wave kernel) as a kernel name, and the
    sampled sequence density behind it"""
    s = "true" if safe else "false"
    if n_blocks <= 16 * 256:
        return decode_kernel_name(n_blocks, safe), None
    route, hops, nbytes, _avg, near, offs, outb, _ = amd.last_decode_route(dev_index)
    name = "decode_wave_kernel<16, 8192, 1024, %s, 5>" % s if route == 2 else "decode_ring_kernel<4, 2048, %s>" % s if route == 1 else \
        "decode_deep_kernel<8, %s>" % s if route == 3 else decode_kernel_name(n_blocks, safe)
    return name, {"route": route, "sampled_sequences_per_256B": round(256.0 * hops / nbytes, 1) if nbytes else None,
                  "sampled_output_bytes_per_sequence": round(outb / offs, 2) if offs else None,
                  "sampled_offsets_within_6KB": round(near / offs, 3) if offs else None}


def spread(r, *keys):
    """best-of-N / median-of-N of the named rates of a cpu_bench line (1.0 = every repetition the same; the harness repeats whole
    passes for >= CPU_BENCH_MIN_MS (300; headline 1000) per repetition on a persistent pinned thread pool, oracle/cpu_bench.c)"""
    return {k: round(r[k] / r[k + "_median"], 3) for k in keys if r.get(k + "_median")}


def cpu_quota():
    """CPUs' worth of time the box's cgroup allows this process (cpu.max / cfs quota; None = unlimited or unknown): os.cpu_count()
    counts the host's threads, and a box that is a slice of a node runs 256 baseline threads on whatever its quota is"""
    try:
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p_), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p_, 2)
    except Exception:
        return None


def cpu_entry(fn):
    try:
        r = fn()
        if isinstance(r, dict):
            r.setdefault("cgroup_cpu_quota", cpu_quota())
            r.setdefault("host_threads", os.cpu_count())
        return r
    except Exception as e:  # a baseline is a reported extra, never a reason to lose the GPU line
        return {"value": None, "unit": "GB/s", "cores": len(os.sched_getaffinity(0)), "kind": "port", "sample": "failed: %r" % (e,)}


def relaunch(n):
    """`bench.py --gpus N` started without a launcher: become N ranks under torch.distributed.run"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


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
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the configs[2..4] / decompress_fast sub-objects")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from profiles/traffic.json only (no PMC child runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch(args.gpus)

    import torch
    import torch.distributed as dist

    amd = importlib.import_module("lz4-java_amd")
    shard = importlib.import_module("lz4-java_amd.shard")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    live, live_note = {}, None
    if args.gpus == 1 and world == 1 and not args.no_live_traffic:
        # (behind `import torch` -- a fresh box pages the image in once, for the children too -- and before this process has a context on the GPU)
        live, live_note = live_traffic(args, timeout_s=180)
        if not live:
            print("bench.py: headline traffic from profiles/traffic.json (%s)" % live_note, file=sys.stderr, flush=True)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    joined = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        one = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(one)                      # every rank that joined the RCCL group adds itself
        joined = int(one.item())
        if joined != args.gpus:
            raise SystemExit("bench.py: %d rank(s) joined the nccl group, %d were asked for" % (joined, args.gpus))
    if args.decode_lanes:
        amd.set_option("decode_lanes", args.decode_lanes)

    i64, i32, u8 = torch.int64, torch.int32, torch.uint8
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            tt = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return x

    def all_ok(ok):
        if world > 1:
            okt = torch.tensor([1 if ok else 0], device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            return bool(okt.item())
        return ok

    def timed(fn, reps, warm=1):
        """reps launches of fn between barriers: (wall seconds per launch, max over ranks; HIP-event seconds per launch on this rank)"""
        for _ in range(warm):
            fn()
        barrier()
        a, b = ev(), ev()
        t0 = time.perf_counter()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        barrier()
        return max_over_ranks((time.perf_counter() - t0) / reps), a.elapsed_time(b) / reps * 1e-3

    def batch(n, blk, cap):
        return dict(so=torch.arange(n, dtype=i64, device=dev) * blk, sl=torch.full((n,), blk, dtype=i32, device=dev),
                    co=torch.arange(n, dtype=i64, device=dev) * cap, cc=torch.full((n,), cap, dtype=i32, device=dev),
                    clen=torch.zeros(n, dtype=i32, device=dev), dlen=torch.zeros(n, dtype=i32, device=dev))

    # ------------------------------------------------------------------------------------------------------------------
    # headline: configs[1]
    # ------------------------------------------------------------------------------------------------------------------
    n, blk = args.blocks, args.block_size
    cap = amd.maxCompressedLength(blk)
    src = torch.empty(n * blk, dtype=u8, device=dev)
    comp = torch.empty(n * cap, dtype=u8, device=dev)
    back = torch.zeros(n * blk, dtype=u8, device=dev)
    B = batch(n, blk, cap)
    so, sl, co, cc, clen, dlen = B["so"], B["sl"], B["co"], B["cc"], B["clen"], B["dlen"]
    assert shard.block_range(n * world, world, rank) == (rank * n, (rank + 1) * n)
    # rank r owns block indices [r*n, (r+1)*n): contiguous ranges, no data exchange
    amd.DeviceBatch.gen_blocks(src, blk, blk, n, first_idx=rank * n, litmax=args.litmax, win=args.win)
    torch.cuda.synchronize()

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

    for _ in range(args.warmup):
        step()
    barrier()
    evs = [[ev() for _ in range(4)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)

    # ---- checks (outside the timed region): the work was real ----
    ok = bool(torch.equal(back, src)) and bool((clen > 0).all()) and bool(torch.equal(dlen, sl))
    csum = int(clen.sum().item())
    t_c = sum(e[0].elapsed_time(e[1]) for e in evs) / args.steps * 1e-3   # s per compress launch (HIP events on the launch stream)
    t_d = sum(e[2].elapsed_time(e[3]) for e in evs) / args.steps * 1e-3
    t_g = sum(e[1].elapsed_time(e[2]) for e in evs) / args.steps * 1e-3   # the size all-gather as the launch stream sees it (0 at N = 1: no collective)
    dev_index = dev.index or 0
    ok = all_ok(ok)
    nbytes = float(n) * blk

    def roof(kernel, alg_bytes, secs, traffic=None):
        return {"kernel": kernel, "bound": "hbm", "achieved": round(alg_bytes / secs / 1e9, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(alg_bytes / secs / 1e9 / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(secs * 1e3, 4)}

    extra = {}
    tr = measured_traffic(n, blk) or {}
    try:
        cores = len(os.sched_getaffinity(0)) or 1   # the CPUs this process may run on (a box can be a slice of its node: os.cpu_count() counts the node's)
    except Exception:
        cores = os.cpu_count() or 1
    if cpu_quota():                                  # ... and the CPU time its cgroup grants: more threads than that only get throttled
        cores = max(1, min(cores, int(cpu_quota() + 0.999)))
    want_cpu = world == 1 and rank == 0 and not args.no_cpu_baseline
    if not args.no_extra_configs:
        # ---- decompress_fast on the headline blocks (LZ4FastDecompressor: LZ4JNI.c:169) ----
        back.zero_()
        wall, tk = timed(lambda: amd.DeviceBatch.decompress_fast(comp, co, cc, back, so, sl, dlen), 3)
        okf = all_ok(bool(torch.equal(back, src)) and bool(torch.equal(dlen, clen)))
        extra["decompress_fast"] = {"workload": "the headline blocks, LZ4_decompress_fast", "value": round(world * nbytes / wall / 1e9, 3), "unit": "GB/s",
                                    "verified": okf, "roofline": roof(decode_kernel_name(n, False), nbytes + csum, tk, kernel_traffic(tr, world, decode_kernel_name(n, False)))}
        ok = ok and okf

        # ---- the headline's own bytes against the reference library: 24 blocks of this run (the suite compares 4096 of the same
        # batch, tests/test_gpu_scale.py; the round-4 verdict asked for the bench to compare some itself) ----
        from oracle import oracle as O
        refh = None
        if O.ref_path():
            chk = O.ref()
            clh = clen.cpu().tolist()
            okh, refh = True, 0
            for i in list(range(0, n, max(1, n // 23)))[:24]:
                want = chk.compress_fast(src[i * blk:(i + 1) * blk].cpu().numpy().tobytes())
                okh = okh and clh[i] == len(want) and comp[i * cap:i * cap + clh[i]].cpu().numpy().tobytes() == want
                refh += 1
            okh = all_ok(okh)
            extra["headline_blocks_vs_reference"] = {"blocks": refh, "verified": okh, "what": "compressed bytes of the headline batch == LZ4_compress_default of the reference library"}
            ok = ok and okh
        # ---- small launches of the headline's blocks: one block (the Java single-call path: one block per call,
        # LZ4JNISafeDecompressor.java:34-43) and 512 / 2048 blocks (a reader's batch); a wavefront per block (lz4_decode_wave.h) ----
        small = {}
        for ns in (1, 512, 2048):
            if ns > n:
                continue
            back[:ns * blk].zero_()
            ws_, tks_ = timed(lambda: amd.DeviceBatch.decompress_safe(comp, co[:ns], clen[:ns], back, so[:ns], sl[:ns], dlen[:ns]), 20, warm=3)
            oks = all_ok(bool(torch.equal(back[:ns * blk], src[:ns * blk])))
            small["%d" % ns] = {"ms_per_launch": round(tks_ * 1e3, 4), "GBps": round(ns * blk / tks_ / 1e9, 3), "verified": oks,
                                "kernel": decode_kernel_name(ns)}
            ok = ok and oks
        extra["decode_small_launches"] = {"workload": "LZ4_decompress_safe of the first 1 / 512 / 2048 headline blocks per launch (HIP-event time per launch)",
                                          "unit": "GB/s", "launches": small}

        # ---- end to end: what a JNI caller reaches (LZ4JNI.c:53-84: pin, ONE call, release) is the host-pointer batch API ----
        # lz4hip_compress_fast_batch / lz4hip_decompress_safe_batch on a sample of the headline blocks lying in PAGEABLE host memory:
        # staging, H2D, kernels, D2H all inside the timed call.  Never `value`.
        if rank == 0:
            import numpy as np
            ne = min(n, 16384)
            h_src = bytearray(src[:ne * blk].cpu().numpy().tobytes())
            h_dst, h_back = bytearray(ne * cap), bytearray(ne * blk)
            so_h, sl_h = np.arange(ne, dtype=np.uint64) * blk, np.full(ne, blk, dtype=np.int32)
            do_h, dc_h = np.arange(ne, dtype=np.uint64) * cap, np.full(ne, cap, dtype=np.int32)
            tc, td = [], []
            for _ in range(4):
                t0 = time.perf_counter()
                sizes = amd.LZ4HIPBatch.compress(h_src, so_h, sl_h, h_dst, do_h, dc_h)
                t1 = time.perf_counter()
                res = amd.LZ4HIPBatch.decompressSafe(h_dst, do_h, sizes, h_back, so_h, sl_h)
                t2 = time.perf_counter()
                tc.append(t1 - t0); td.append(t2 - t1)
            oke = bytes(h_back) == bytes(h_src) and bool((np.asarray(res) == blk).all()) and \
                bool((np.asarray(sizes) == clen[:ne].cpu().numpy()).all())
            tc, td = sorted(tc[1:]), sorted(td[1:])          # (the first call allocates the staging pool)
            eb = float(ne) * blk
            extra["end_to_end"] = {"workload": "%d of the headline blocks in pageable host memory through lz4hip_compress_fast_batch / "
                                               "lz4hip_decompress_safe_batch (staging + PCIe both ways + kernels), best / median of 3 calls" % ne,
                                   "unit": "GB/s", "verified": oke,
                                   "compress_GBps": round(eb / tc[0] / 1e9, 3), "compress_GBps_median": round(eb / tc[1] / 1e9, 3),
                                   "decompress_GBps": round(eb / td[0] / 1e9, 3), "decompress_GBps_median": round(eb / td[1] / 1e9, 3),
                                   "roundtrip_GBps": round(eb / (tc[0] + td[0]) / 1e9, 3),
                                   "note": "compare with cpu_baseline (all threads, one_thread) of the headline: the reference works on host memory in place"}
            ok = ok and oke
            del h_src, h_dst, h_back

        # ---- device-resident container encode: the data blocks of an LZ4 Frame for the headline bytes, compressed AND
        # assembled on the device (raw fallback, size scan, size words, payload compaction), nothing crossing PCIe ----
        total_t = torch.zeros(1, dtype=i64, device=dev)
        wf, tkf = timed(lambda: amd.DeviceBatch.container_blocks(0, src, blk, comp, total_t), 2)
        tot = int(total_t.item())
        w0 = int.from_bytes(comp[:4].cpu().numpy().tobytes(), "little")
        okc = all_ok(tot == csum + 4 * n and w0 == int(clen[0]))
        # (every block's size word + payload: checked against the batch call's sizes; byte identity with the host-assembled frames and
        # the lz4 CLI: tests/test_gpu_streams.py)
        extra["frame_blocks_dev"] = {"workload": "the headline bytes as ONE buffer -> the data blocks of an LZ4 Frame with 64 KiB blocks, compressed and "
                                                 "assembled on the device (lz4hip_container_blocks_dev)", "value": round(world * nbytes / wf / 1e9, 3),
                                     "unit": "GB/s", "verified": okc, "bytes_out": tot}
        ok = ok and okc

        # ---- device-resident container decode (READ side): the frame body just assembled is walked (size words),
        # decoded and its blocks' sizes checked on the device (lz4hip_container_decode_dev) ----
        import ctypes as C
        L = amd.lib()
        wsb = L.lz4hip_container_decode_workspace_bytes(n)
        ws_t = torch.empty(wsb, dtype=u8, device=dev)
        sizes_t = torch.zeros(n, dtype=i32, device=dev)
        info_t = torch.zeros(5, dtype=i64, device=dev)
        back.zero_()

        def frame_read():
            rc = L.lz4hip_container_decode_dev(0, 0, comp.data_ptr(), tot, blk, back.data_ptr(), blk, n, sizes_t.data_ptr(), info_t.data_ptr(),
                                               ws_t.data_ptr(), wsb, dev.index or 0, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, L.lz4hip_last_error()
        wr, tkr = timed(frame_read, 2)
        inf = [int(x) for x in info_t.cpu()]
        okr = all_ok(inf[0] == n and inf[1] == tot and inf[2] == 1 and inf[3] == n * blk and bool(torch.equal(back, src)))
        extra["frame_read_dev"] = {"workload": "the frame body above (%d blocks) -> walked, decoded and checked on the device (lz4hip_container_decode_dev); "
                                               "stop reason 1 = the body ended at a block boundary" % n,
                                   "value": round(world * nbytes / wr / 1e9, 3), "unit": "GB/s", "verified": okr, "ms": round(tkr * 1e3, 3),
                                   "note": "round 5: the size words are walked in parallel (a speculatively validated candidate per region of the body, a lane per region, "
                                           "a stitch in stream order; the one-lane serial walk -- 36.7 ms for these blocks -- is the fallback)"}
        ok = ok and okr
        # ---- the same for lz4-java's LZ4Block container: the headline bytes as 64 KiB LZ4Block blocks (21-byte headers,
        # XXH32 checks) assembled on the device, then walked IN PARALLEL (the headers carry a magic: a candidate per region, a lane per
        # region, a stitch -- kernels.hip container_walk_par_kernel), decoded by the fast decoder and check-summed on the device ----
        wfb, _ = timed(lambda: amd.DeviceBatch.container_blocks(1, src, blk, comp, total_t), 1)
        totb = int(total_t.item())
        back.zero_()

        def block_read():
            rc = L.lz4hip_container_decode_dev(1, 0, comp.data_ptr(), totb, blk, back.data_ptr(), blk, n, sizes_t.data_ptr(), info_t.data_ptr(),
                                               ws_t.data_ptr(), wsb, dev.index or 0, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, L.lz4hip_last_error()
        wrb, tkrb = timed(block_read, 2)
        infb = [int(x) for x in info_t.cpu()]
        okrb = all_ok(totb == csum + 21 * n and infb[0] == n and infb[1] == totb and infb[2] == 1 and infb[3] == n * blk and bool(torch.equal(back, src)))
        extra["block_read_dev"] = {"workload": "the headline bytes as an LZ4Block stream (%d blocks of 64 KiB, device-assembled) -> headers found and walked in parallel, "
                                               "blocks decoded (LZ4_decompress_fast into the headers' original lengths) and XXH32-checked on the device "
                                               "(lz4hip_container_decode_dev kind 1)" % n,
                                   "value": round(world * nbytes / wrb / 1e9, 3), "unit": "GB/s", "verified": okrb, "ms": round(tkrb * 1e3, 3)}
        ok = ok and okrb
        del ws_t, sizes_t, info_t

        # ---- real text: BASELINE configs[0] names a 64 KiB Silesia/dickens block; the only real data of the reference are
        # src/test-resources/calgary/* (LZ4Test.java:335-348), so: every block = 64 KiB of Calgary book1 (English prose, the same
        # class) from a different offset.  Compressed bytes of a sample of blocks are compared with the reference library's. ----
        bpath = os.path.join(ROOT, "tests", "golden", "book1_200000.bin")
        if os.path.exists(bpath):
            import numpy as np
            book = np.frombuffer(open(bpath, "rb").read(), dtype=np.uint8)
            span = len(book) - blk
            bdev = torch.from_numpy(book.copy()).to(dev)
            offs = (torch.arange(n, dtype=i64, device=dev) + rank * n) * 7919 % span
            ar = torch.arange(blk, dtype=i64, device=dev)
            for c0 in range(0, n, 1024):
                c1 = min(n, c0 + 1024)
                src[c0 * blk:c1 * blk] = bdev[(offs[c0:c1, None] + ar[None, :]).reshape(-1)]
            del ar
            torch.cuda.synchronize()
            wc, tkc = timed(lambda: amd.DeviceBatch.compress_fast(src, so, sl, comp, co, cc, clen), 2)
            csb = int(clen.sum().item())
            back.zero_()
            wd, tkd = timed(lambda: amd.DeviceBatch.decompress_safe(comp, co, clen, back, so, sl, dlen), 2)
            okb = bool(torch.equal(back, src)) and bool(torch.equal(dlen, sl))
            from oracle import oracle as O
            if O.ref_path():     # the reference's own liblz4 (oracle/_ref travels with the repo): bytes of 48 blocks
                chk = O.ref()
                cl = clen.cpu().tolist()
                for i in list(range(0, n, max(1, n // 47)))[:48]:
                    want = chk.compress_fast(book[int(offs[i]):int(offs[i]) + blk].tobytes())
                    okb = okb and cl[i] == len(want) and comp[i * cap:i * cap + cl[i]].cpu().numpy().tobytes() == want
            okb = all_ok(okb)
            dk_name, dk_route = routed_decoder(amd, dev_index, n)
            # the same text in smaller launches: which decoder each gets (pair loop / wave loop / routed by density) and what it delivers
            by_n = {}
            for nb_ in (2048, 8192, 16384):
                if nb_ >= n:
                    continue
                back[:nb_ * blk].zero_()
                wq, tq = timed(lambda: amd.DeviceBatch.decompress_safe(comp, co[:nb_], clen[:nb_], back, so[:nb_], sl[:nb_], dlen[:nb_]), 3)
                kq, rq = routed_decoder(amd, dev_index, nb_)
                okq = all_ok(bool(torch.equal(back[:nb_ * blk], src[:nb_ * blk])))
                by_n[str(nb_)] = {"decompress_GBps": round(world * float(nb_) * blk / wq / 1e9, 3), "ms_per_launch": round(tq * 1e3, 4), "kernel": kq, "route": rq, "verified": okq}
                okb = okb and okq
            extra["real_book1"] = {"workload": "%d x 64 KiB slices of Calgary book1 per GPU (stand-in for configs[0]'s Silesia/dickens block), fast compress + "
                                               "safe decompress, ratio %.3f; compressed bytes of 48 blocks vs the reference library" % (n, nbytes / csb),
                                   "unit": "GB/s", "verified": okb,
                                   "compress_GBps": round(world * nbytes / wc / 1e9, 3), "decompress_GBps": round(world * nbytes / wd / 1e9, 3),
                                   # (HBM bytes of these launches: PMC passes of the text legs by themselves, tools/gpu_text_legs.py)
                                   "roofline_compress": roof("compress_fast_v2w_cu_kernel + compress_fast_ms_cu_kernel", nbytes + csb, tkc, text_traffic(tr, world, n, "real_book1", "compress")),
                                   "roofline_decode": roof(dk_name, nbytes + csb, tkd, text_traffic(tr, world, n, "real_book1", "decode")),
                                   "decode_route": dk_route, "decode_by_blocks_per_launch": by_n}
            ok = ok and okb
            if want_cpu:
                def fb():
                    r = cpu_bench([min(n, 64 * cores), blk, cores, 3, 0, args.litmax, args.win], {"CPU_BENCH_FILE": bpath})
                    return {"compress_GBps": round(r["compress_GBps"], 3), "decompress_safe_GBps": round(r["decompress_safe_GBps"], 3),
                            "compress_GBps_median": round(r["compress_GBps_median"], 3), "decompress_safe_GBps_median": round(r["decompress_safe_GBps_median"], 3),
                            "best_over_median": spread(r, "compress_GBps", "decompress_safe_GBps"),
                            "unit": "GB/s", "cores": cores, "kind": r["_kind"],
                            "sample": "%d x 64 KiB slices of book1, %d threads, best (and median) of 3 repetitions of >= %d ms each" % (r["n_blocks"], cores, r.get("min_ms", 0))}
                extra["real_book1"]["cpu_baseline"] = cpu_entry(fb)
            # ---- the same text in 4 MiB blocks (the Frame format's largest block, the reference's default): every block is made of
            # 70000-byte slices of book1 -- longer than the match window, and the same text comes back no closer than 130000 bytes,
            # so a block is English text throughout.  Blocks of 65547 bytes .. 4 MiB run on the ten-chain kernel with packed table
            # entries (no routing to the window-parallel core there: ten lean chains are the faster ones on text too). ----
            bT, nT = 4 << 20, max(1, 2560 // world)
            capT = amd.maxCompressedLength(bT)
            sT = torch.empty(nT * bT, dtype=u8, device=dev)
            slc = 70000
            per = (bT + slc - 1) // slc
            arT = torch.arange(slc, dtype=i64, device=dev)
            blkT = (torch.arange(nT, dtype=i64, device=dev) + rank * nT) * 7919
            sTv = sT.view(nT, bT)
            for j in range(per):   # slice j of every block at once
                wdt = min(slc, bT - j * slc)
                oT = (blkT + j * 70000) % (len(book) - slc)
                sTv[:, j * slc:j * slc + wdt] = bdev[oT[:, None] + arT[None, :wdt]]
            del blkT, sTv
            cT = torch.empty(nT * capT, dtype=u8, device=dev)
            BT = batch(nT, bT, capT)
            wT, tkT = timed(lambda: amd.DeviceBatch.compress_fast(sT, BT["so"], BT["sl"], cT, BT["co"], BT["cc"], BT["clen"]), 2)
            csT = int(BT["clen"].sum().item())
            okT = True
            if O.ref_path():
                clT = BT["clen"].cpu().tolist()
                for i in sorted(set([0, nT // 2, nT - 1])):
                    want = O.ref().compress_fast(sT[i * bT:(i + 1) * bT].cpu().numpy().tobytes())
                    okT = okT and clT[i] == len(want) and cT[i * capT:i * capT + clT[i]].cpu().numpy().tobytes() == want
            okT = all_ok(okT)
            extra["real_book1_4MiB"] = {"workload": "%d x 4 MiB blocks of book1 text per GPU, fast compress (byU32, packed entries, ten chains per CU), ratio %.3f; "
                                                    "compressed bytes of 3 blocks vs the reference library" % (nT, nT * bT / csT),
                                        "value": round(world * float(nT) * bT / wT / 1e9, 3), "unit": "GB/s", "verified": okT,
                                        "roofline": roof("compress_fast_v2wp_cu_kernel", float(nT) * bT + csT, tkT, text_traffic(tr, world, n, "real_book1_4MiB", "compress") if nT == 2560 else None)}
            ok = ok and okT
            del sT, cT, BT, arT
            del bdev, offs
        del comp, back
        torch.cuda.empty_cache()

        # ---- configs[2]: safe decode of 4 MiB blocks (gen_block(4 MiB, win 4096): fast ratio ~2.1) ----
        # BASELINE configs[2] is 16384 blocks in all: one GPU takes them in ONE launch (3 x 64 GiB of buffers: input, compressed slots,
        # output), N GPUs shard them.  Halved until the buffers fit what the device has free (never on a 288 GB MI355X at N = 1).
        b3 = 4 << 20
        cap3 = amd.maxCompressedLength(b3)
        n3 = max(1, 16384 // world)
        free_b = torch.cuda.mem_get_info(dev)[0]
        while n3 > 256 and n3 * (2 * b3 + cap3) * 1.04 > free_b:
            n3 //= 2
        s3 = torch.empty(n3 * b3, dtype=u8, device=dev)
        amd.DeviceBatch.gen_blocks(s3, b3, b3, n3, first_idx=(1 << 24) + rank * n3, litmax=args.litmax, win=4096)
        c3 = torch.empty(n3 * cap3, dtype=u8, device=dev)
        B3 = batch(n3, b3, cap3)
        # the setup of configs[2] is a measurement of its own: fast compress of 4 MiB blocks (byU32 table, 5-byte hash -- the LZ4Frame
        # default block size, LZ4FrameOutputStream.java:169-171); bit-exactness of these bytes: tests/test_gpu_scale.py
        wall3c, tk3c = timed(lambda: amd.DeviceBatch.compress_fast(s3, B3["so"], B3["sl"], c3, B3["co"], B3["cc"], B3["clen"]), 1, warm=0)
        cs3 = int(B3["clen"].sum().item())
        # (blocks of 65547 bytes .. 4 MiB: the ten-chain kernel with packed table entries, kernels.hip compress_fast_v2wp_cu_kernel)
        extra["compress_4MiB"] = {"workload": "%d x 4 MiB blocks per GPU, App.F win 4096, LZ4_compress_default (byU32), ratio %.3f" % (n3, n3 * b3 / cs3),
                                  "value": round(world * float(n3) * b3 / wall3c / 1e9, 3), "unit": "GB/s", "verified": None,
                                  "roofline": roof("compress_fast_v2wp_cu_kernel", float(n3) * b3 + cs3, tk3c, kernel_traffic(tr, world, "compress_fast_v2wp_cu_kernel"))}
        bk3 = torch.zeros(n3 * b3, dtype=u8, device=dev)
        wall, tk = timed(lambda: amd.DeviceBatch.decompress_safe(c3, B3["co"], B3["clen"], bk3, B3["so"], B3["sl"], B3["dlen"]), 2)
        ok3 = all_ok(bool(torch.equal(bk3, s3)))
        # its bytes decode back to the input AND a sample of the blocks equals the reference library's output byte for byte
        ref3 = None
        k3_name, k3_route = routed_decoder(amd, dev_index, n3)
        from oracle import oracle as O
        cl3 = B3["clen"].cpu().tolist()
        if O.ref_path():
            chk = O.ref()
            ref3 = 0
            for i in sorted(set([0, n3 - 1] + list(range(0, n3, max(1, n3 // 6)))))[:8]:
                want = chk.compress_fast(s3[i * b3:(i + 1) * b3].cpu().numpy().tobytes())
                ok3 = ok3 and cl3[i] == len(want) and c3[i * cap3:i * cap3 + cl3[i]].cpu().numpy().tobytes() == want
                ref3 += 1
            ok3 = all_ok(ok3)
        extra["compress_4MiB"]["verified"] = ok3
        extra["compress_4MiB"]["blocks_vs_reference"] = ref3
        extra["configs2_decode_4MiB"] = {"workload": "%d x 4 MiB blocks per GPU in one launch (BASELINE configs[2]: 16384 blocks in all, sharded over the ranks), "
                                                     "App.F win 4096, LZ4_decompress_safe of fast-compressed blocks, ratio %.3f" % (n3, n3 * b3 / cs3),
                                         "value": round(world * float(n3) * b3 / wall / 1e9, 3), "unit": "GB/s", "verified": ok3,
                                         "kernel_ran": k3_name, "route": k3_route,
                                         "roofline": roof(k3_name, float(n3) * b3 + cs3, tk,
                                                          kernel_traffic(tr, world, k3_name) if n3 == (tr.get("configs2_blocks") or 16384) else None)}
        ok = ok and ok3
        # what the one-GPU shard legs of the committed N = 1 line predicted for this rank count (profiles/shard_prediction.json, written from
        # the line by tools/collect_profiles.sh): beside the measured figure, so that a shortfall at N > 1 can be told from the line alone
        try:
            pred = json.load(open(os.path.join(ROOT, "profiles", "shard_prediction.json")))
            extra["configs2_decode_4MiB"]["predicted_per_gpu_GBps"] = (pred.get("configs2_decode_per_gpu_GBps") or {}).get(str(world))
            extra["configs2_decode_4MiB"]["prediction_source"] = pred.get("source")
        except Exception:
            pass
        # ---- the 8-GPU shard of configs[2] ON THIS GPU: `--gpus 8` gives every rank 16384 / 8 = 2048 blocks (n3 above) and the ranks
        # exchange no data, so one GPU running 2048 blocks IS what each of eight would run; x 8 is the PREDICTED aggregate, not a
        # measurement of eight GPUs.  (Round-4 verdict: the lane-group decoders take 63 ms for a launch of 4 MiB blocks whether it
        # carries 2048 or 8192 -- 135 GB/s per GPU at 2048; round 5: a wavefront per block, several sequences per trip.) ----
        if world == 1 and n3 >= 2048:
            nS = 2048
            bkS = bk3[:nS * b3]
            bkS.zero_()
            wS, tkS = timed(lambda: amd.DeviceBatch.decompress_safe(c3, B3["co"][:nS], B3["clen"][:nS], bkS, B3["so"][:nS], B3["sl"][:nS], B3["dlen"][:nS]), 3)
            okS = all_ok(bool(torch.equal(bkS, s3[:nS * b3])))
            csS = int(B3["clen"][:nS].sum().item())
            # ... and on streams that ARE the reference library's bytes: 64 of the shard's blocks compressed by liblz4 on the host, every
            # one decoded 32 times into 2048 distinct slots (the same launch shape, the same kernel)
            refS = None
            if O.ref_path():
                import concurrent.futures as cf
                import numpy as np
                chk = O.ref()
                hostb = [s3[i * b3:(i + 1) * b3].cpu().numpy().tobytes() for i in range(64)]
                with cf.ThreadPoolExecutor(cores) as ex:          # (ctypes releases the GIL in the call)
                    rstreams = list(ex.map(chk.compress_fast, hostb))
                okref = all(len(r_) == cl3[i] and c3[i * cap3:i * cap3 + cl3[i]].cpu().numpy().tobytes() == r_ for i, r_ in enumerate(rstreams[:8]))   # (the engine's bytes for these blocks are the same bytes)
                offs_h, p_ = [], 0
                for r_ in rstreams:
                    offs_h.append(p_); p_ += len(r_) + 5
                packed = torch.from_numpy(np.frombuffer(b"".join(r_ + b"\0" * 5 for r_ in rstreams), dtype=np.uint8).copy()).to(dev)
                coR = torch.tensor([offs_h[i % 64] for i in range(nS)], dtype=i64, device=dev)
                clR = torch.tensor([len(rstreams[i % 64]) for i in range(nS)], dtype=i32, device=dev)
                bkS.zero_()
                wR, tkR = timed(lambda: amd.DeviceBatch.decompress_safe(packed, coR, clR, bkS, B3["so"][:nS], B3["sl"][:nS], B3["dlen"][:nS]), 3)
                okR = bool(torch.equal(B3["dlen"][:nS], B3["sl"][:nS]))
                v64 = s3[:64 * b3].view(64, b3)
                for j in range(nS // 64):
                    okR = okR and bool(torch.equal(bkS[j * 64 * b3:(j + 1) * 64 * b3].view(64, b3), v64))
                okR = all_ok(okR and okref)
                refS = {"value": round(float(nS) * b3 / wR / 1e9, 3), "unit": "GB/s", "verified": okR, "streams": 64,
                        "what": "64 blocks of the shard compressed by the reference library on the host (LZ4_compress_default), each decoded into 32 of the 2048 slots"}
                okS = okS and okR
                del packed, coR, clR
            extra["configs2_shard8"] = {"workload": "2048 x 4 MiB blocks in one launch: what EACH rank of `--gpus 8` decodes of BASELINE configs[2] (16384 blocks "
                                                    "sharded over 8 GPUs, no data exchanged), measured on one GPU; LZ4_decompress_safe",
                                        "value": round(float(nS) * b3 / wS / 1e9, 3), "unit": "GB/s", "verified": okS,
                                        "predicted_8gpu_aggregate_GBps": round(8.0 * nS * b3 / wS / 1e9, 1),
                                        "note": "predicted aggregate = 8 x this GPU's rate (independent shards); not a multi-GPU measurement",
                                        "reference_compressed_streams": refS,
                                        "roofline": roof(decode_kernel_name(nS, True, True), float(nS) * b3 + csS, tkS, kernel_traffic(tr, world, decode_kernel_name(nS, True, True)))}
            # the shards of 2 and 4 ranks as well (8192 / 4096 blocks per launch): the prediction bench.py --gpus N prints beside its measurement
            shards = {"8": round(float(nS) * b3 / wS / 1e9, 3), "1": extra["configs2_decode_4MiB"]["value"]}
            for wv_, nb_ in ((2, 8192), (4, 4096)):
                if nb_ > n3:
                    continue
                bkq = bk3[:nb_ * b3]
                bkq.zero_()
                wq, tq = timed(lambda: amd.DeviceBatch.decompress_safe(c3, B3["co"][:nb_], B3["clen"][:nb_], bkq, B3["so"][:nb_], B3["sl"][:nb_], B3["dlen"][:nb_]), 2)
                okq = all_ok(bool(torch.equal(bkq, s3[:nb_ * b3])))
                okS = okS and okq
                shards[str(wv_)] = round(float(nb_) * b3 / wq / 1e9, 3)
                extra["configs2_shard%d" % wv_] = {"workload": "%d x 4 MiB blocks in one launch: the shard of each of %d ranks" % (nb_, wv_), "value": shards[str(wv_)], "unit": "GB/s",
                                                   "verified": okq, "kernel": routed_decoder(amd, dev_index, nb_)[0], "predicted_aggregate_GBps": round(wv_ * shards[str(wv_)], 1)}
            extra["configs2_shards_per_gpu_GBps"] = shards
            wSc, tkSc = timed(lambda: amd.DeviceBatch.compress_fast(s3, B3["so"][:nS], B3["sl"][:nS], c3, B3["co"][:nS], B3["cc"][:nS], B3["clen"][:nS]), 2)
            # (deterministic: the same sizes as the full launch wrote, and 8 blocks' bytes against the reference library once more)
            okSc = B3["clen"][:nS].cpu().tolist() == cl3[:nS]
            if O.ref_path():
                for i in sorted(set([0, nS - 1] + list(range(0, nS, nS // 6))))[:8]:
                    okSc = okSc and c3[i * cap3:i * cap3 + cl3[i]].cpu().numpy().tobytes() == O.ref().compress_fast(s3[i * b3:(i + 1) * b3].cpu().numpy().tobytes())
            okSc = all_ok(okSc)
            extra["configs2_shard8_compress"] = {"workload": "the same 2048 x 4 MiB shard, LZ4_compress_default (byU32)",
                                                 "value": round(float(nS) * b3 / wSc / 1e9, 3), "unit": "GB/s", "verified": okSc,
                                                 "blocks_vs_reference": 8 if O.ref_path() else 0,
                                                 "predicted_8gpu_aggregate_GBps": round(8.0 * nS * b3 / wSc / 1e9, 1),
                                                 "roofline": roof("compress_fast_v2wp_cu_kernel", float(nS) * b3 + csS, tkSc, kernel_traffic(tr, world, "compress_fast_v2wp_cu_kernel@shard8"))}
            ok = ok and okS and okSc
        if want_cpu:
            def f3():
                r = cpu_bench([min(256, 2 * cores), b3, cores, 3, 1 << 24, args.litmax, 4096])
                return {"value": round(r["decompress_safe_GBps"], 3), "median": round(r["decompress_safe_GBps_median"], 3),
                        "per_core": round(r["decompress_safe_GBps"] / cores, 4), "compress_GBps": round(r["compress_GBps"], 3),
                        "compress_GBps_median": round(r["compress_GBps_median"], 3),
                        "best_over_median": spread(r, "compress_GBps", "decompress_safe_GBps"),
                        "unit": "GB/s", "cores": cores, "kind": r["_kind"],
                        "sample": "%d x 4 MiB blocks (same generator), %d threads, best (and median) of 3, LZ4_decompress_safe / LZ4_compress_default" % (r["n_blocks"], cores)}
            extra["configs2_decode_4MiB"]["cpu_baseline"] = cpu_entry(f3)
        del s3, c3, bk3, B3
        torch.cuda.empty_cache()

        # ---- configs[3]: LZ4 HC level 9 of 1 MiB blocks ----
        n4, b4 = 4096, 1 << 20
        cap4 = amd.maxCompressedLength(b4)
        s4 = torch.empty(n4 * b4, dtype=u8, device=dev)
        amd.DeviceBatch.gen_blocks(s4, b4, b4, n4, first_idx=(2 << 24) + rank * n4, litmax=args.litmax, win=4096)
        c4 = torch.empty(n4 * cap4, dtype=u8, device=dev)
        B4 = batch(n4, b4, cap4)
        wall, tk = timed(lambda: amd.DeviceBatch.compress_hc(s4, B4["so"], B4["sl"], c4, B4["co"], B4["cc"], B4["clen"], 9), 2)
        cs4 = int(B4["clen"].sum().item())
        bk4 = torch.zeros(n4 * b4, dtype=u8, device=dev)
        amd.DeviceBatch.decompress_safe(c4, B4["co"], B4["clen"], bk4, B4["so"], B4["sl"], B4["dlen"])
        ok4 = all_ok(bool(torch.equal(bk4, s4)) and bool((B4["clen"] > 0).all()))
        extra["configs3_hc9_1MiB"] = {"workload": "%d x 1 MiB blocks per GPU, App.F win 4096, LZ4_compress_HC level 9, ratio %.3f" % (n4, n4 * b4 / cs4),
                                      "value": round(world * float(n4) * b4 / wall / 1e9, 3), "unit": "GB/s", "verified": ok4,
                                      # compress reads N, writes C.  (The two-kernel scheme also writes and re-reads a u16 chain
                                      # delta per input byte -- 4 N of workspace traffic, reported apart, not algorithmic bytes.)
                                      "roofline": roof("hc_build_kernel + hc_parse_kernel", float(n4) * b4 + cs4, tk,
                                                       kernel_traffic(tr, world, "hc_build_kernel", "hc_parse_kernel")),
                                      "workspace_bytes_per_launch": int(4 * n4 * b4),
                                      "note": "levels 10-12 (liblz4's optimal parser) are functional only: ~1 GB/s, not benchmarked here"}
        ok = ok and ok4
        if want_cpu:
            def f4():
                r = cpu_bench([min(512, 2 * cores), b4, cores, 2, 2 << 24, args.litmax, 4096], {"LZ4_HC_LEVEL": "9"})
                return {"value": round(r["compress_GBps"], 3), "median": round(r["compress_GBps_median"], 3), "per_core": round(r["compress_GBps"] / cores, 4),
                        "best_over_median": spread(r, "compress_GBps"),
                        "unit": "GB/s", "cores": cores, "kind": r["_kind"],
                        "sample": "%d x 1 MiB blocks (same generator), %d threads, best (and median) of 2, LZ4_compress_HC level 9" % (r["n_blocks"], cores)}
            extra["configs3_hc9_1MiB"]["cpu_baseline"] = cpu_entry(f4)
        del c4, bk4, B4
        torch.cuda.empty_cache()

        # ---- configs[4]: XXH32 / XXH64 of 4 KiB buffers (the 4 GiB of configs[3]'s input as 1 Mi slices) ----
        n5, b5 = 1 << 20, 4096
        off5 = torch.arange(n5, dtype=i64, device=dev) * b5
        len5 = torch.full((n5,), b5, dtype=i32, device=dev)
        h32 = torch.zeros(n5, dtype=torch.int32, device=dev)
        h64 = torch.zeros(n5, dtype=torch.int64, device=dev)
        # (sub-millisecond launches: a few of them first, or the first timed kernel runs before the clocks are back up -- 5.4 vs 6.1 TB/s)
        w32, t32 = timed(lambda: amd.DeviceBatch.xxh32(s4, off5, len5, 0x9747b28c, h32), 20, warm=5)
        w64, t64 = timed(lambda: amd.DeviceBatch.xxh64(s4, off5, len5, 0x9747b28c, h64), 20, warm=5)
        # a sample of the hashes against the checker (the reference's own library where oracle/_ref travels, else the C restatement);
        # the full check of every hash: tests/test_gpu_scale.py
        from oracle import oracle as O
        chk = O.ref() if O.ref_path() else O.port()
        host = s4[:64 * b5].cpu().numpy().tobytes()
        a32, a64 = h32[:64].cpu().tolist(), h64[:64].cpu().tolist()
        ok5 = all((a32[i] & 0xFFFFFFFF) == chk.xxh32(host[i * b5:(i + 1) * b5], 0x9747b28c) and
                  (a64[i] & 0xFFFFFFFFFFFFFFFF) == chk.xxh64(host[i * b5:(i + 1) * b5], 0x9747b28c) for i in range(64))
        ok5 = all_ok(ok5)
        extra["configs4_xxhash_4KiB"] = {"workload": "%d x 4 KiB buffers per GPU, seed 0x9747b28c" % n5, "unit": "GB/s", "verified": ok5,
                                         "xxh32": {"value": round(world * float(n5) * b5 / w32 / 1e9, 3), "roofline": roof("xxh_multi_kernel<unsigned int, 4>", float(n5) * (b5 + 4), t32, kernel_traffic(tr, world, "xxh_multi_kernel<unsigned int, 4>"))},
                                         "xxh64": {"value": round(world * float(n5) * b5 / w64 / 1e9, 3), "roofline": roof("xxh_multi_kernel<unsigned long, 4>", float(n5) * (b5 + 8), t64, kernel_traffic(tr, world, "xxh_multi_kernel<unsigned long, 4>"))}}
        ok = ok and ok5
        if want_cpu:
            def f5():
                r = cpu_bench([min(n5, 4096 * cores), b5, cores, 3, 0, args.litmax, 4096], {"XXH_MODE": "1"})
                return {"xxh32_GBps": round(r["xxh32_GBps"], 3), "xxh64_GBps": round(r["xxh64_GBps"], 3),
                        "xxh32_GBps_median": round(r["xxh32_GBps_median"], 3), "xxh64_GBps_median": round(r["xxh64_GBps_median"], 3),
                        "best_over_median": spread(r, "xxh32_GBps", "xxh64_GBps"),
                        "unit": "GB/s", "cores": cores, "kind": r["_kind"],
                        "sample": "%d x 4 KiB buffers, %d threads, best of 3, XXH32 / XXH64 one-shot" % (r["n_blocks"], cores)}
            extra["configs4_xxhash_4KiB"]["cpu_baseline"] = cpu_entry(f5)
        del s4

    # ---- N > 1: every rank's own figures on rank 0's line (a shortfall is attributable to the decoder, the launcher or RCCL from the line alone) ----
    per_rank = None
    if world > 1:
        c2 = (extra.get("configs2_decode_4MiB") or {})
        mine = torch.tensor([t_c * 1e3, t_g * 1e3, t_d * 1e3, float((c2.get("roofline") or {}).get("avg_launch_ms") or 0.0), float((c2.get("route") or {}).get("route") or 0)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r_, "compress_ms": round(float(v_[0]), 3), "gather_ms": round(float(v_[1]), 4), "decompress_ms": round(float(v_[2]), 3),
                     "configs2_decode_ms": round(float(v_[3]), 3), "configs2_route": int(v_[4])} for r_, v_ in enumerate(allr)]
    if rank == 0:
        ratio = nbytes / csum
        value = world * nbytes * args.steps / dt / 1e9
        out = {
            "metric": "uncompressed GB/s (compress + decompress) per GPU, 64 KiB blocks",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": joined, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d x %d B blocks per GPU, SURVEY App.F gen_block(seed 0x4C5A3447, litmax %d, win %d); "
                                   "step = LZ4 fast compress + int32 size gather + LZ4 safe decompress, device-resident"
                                   % (n, blk, args.litmax, args.win),
                       "blocks_per_gpu": n, "block_bytes": blk, "ratio": round(ratio, 4), "parallelism": "blocks sharded x%d" % world},
            "verified": ok,
            "compress_GBps": round(nbytes / t_c / 1e9, 3), "decompress_GBps": round(nbytes / t_d / 1e9, 3),
            # compress: reads N, writes C (1 + 1/ratio B/B); decompress: reads C, writes N
            "roofline": roof("compress_fast_v2w_cu_kernel", nbytes + csum, t_c, live.get("compress_fast_v2w_cu_kernel") or kernel_traffic(tr, world, "compress_fast_v2w_cu_kernel")),
            "roofline_decode": roof(decode_kernel_name(n), nbytes + csum, t_d, live.get(decode_kernel_name(n)) or kernel_traffic(tr, world, decode_kernel_name(n))),
        }
        out["gather_ms"] = round(t_g * 1e3, 4)     # the int32 size all-gather per step (HIP events on the launch stream; no collective at N = 1)
        if per_rank is not None:
            out["per_rank"] = per_rank
        if tr:
            out["traffic_source"] = tr.get("source")
            out["traffic_stale"] = tr.get("kernel_source_hash") != kernel_source_hash()   # true: kernels changed since the PMC passes
        if live_note:
            # the headline's two `traffic` values are this run's own PMC passes when they worked; the other legs' come from the committed file
            hk = ("compress_fast_v2w_cu_kernel", decode_kernel_name(n))
            out["traffic_live"] = {"headline": bool(live.get(hk[0]) and live.get(hk[1])), "note": live_note,
                                   "bytes_per_launch": {k: live.get(k) for k in hk},
                                   "committed_file": {k: kernel_traffic(tr, world, k) for k in hk}}
        if extra:
            out["configs"] = extra
        if want_cpu:
            def f1():
                sample = min(n, 64 * cores)
                r = cpu_bench([sample, blk, cores, 5, 0, args.litmax, args.win], {"CPU_BENCH_MIN_MS": "1000"})
                r1 = cpu_bench([min(n, 256), blk, 1, 3, 0, args.litmax, args.win])     # one host thread: the per-core figure
                return {"value": round(r["roundtrip_GBps"], 3), "median": round(r["roundtrip_GBps_median"], 3), "unit": "GB/s", "cores": cores, "kind": r["_kind"],
                        "sample": "%d x %d B blocks (same generator/seed), %d pinned threads of a persistent pool, best (and median) of 5 repetitions of >= %d ms "
                                  "(whole passes over the sample); LZ4_compress_default + LZ4_decompress_safe; "
                                  "no JVM/JNI overhead; one_thread: %d blocks, best of 3" % (sample, blk, cores, r.get("min_ms", 0), min(n, 256)),
                        "best_over_median": spread(r, "compress_GBps", "decompress_safe_GBps", "roundtrip_GBps"),
                        "compress_GBps": round(r["compress_GBps"], 3), "decompress_safe_GBps": round(r["decompress_safe_GBps"], 3),
                        "decompress_fast_GBps": round(r["decompress_fast_GBps"], 3),
                        "compress_GBps_median": round(r["compress_GBps_median"], 3), "decompress_safe_GBps_median": round(r["decompress_safe_GBps_median"], 3),
                        "per_core": {"compress_GBps": round(r["compress_GBps"] / cores, 4), "decompress_safe_GBps": round(r["decompress_safe_GBps"] / cores, 4)},
                        "one_thread": {"compress_GBps": round(r1["compress_GBps"], 4), "decompress_safe_GBps": round(r1["decompress_safe_GBps"], 4),
                                       "roundtrip_GBps": round(r1["roundtrip_GBps"], 4)},
                        "lib": r["_lib"]}
            out["cpu_baseline"] = cpu_entry(f1)
            try:   # the single-call path side by side: one 64 KiB block on one host core against one launch on the device
                one = out["cpu_baseline"]["one_thread"]["decompress_safe_GBps"]
                out["cpu_baseline"]["one_thread"]["us_per_64KiB_block_decode"] = round(blk / one / 1e3, 2)
                out["configs"]["decode_small_launches"]["launches"]["1"]["host_one_thread_us_per_block"] = round(blk / one / 1e3, 2)
            except Exception:
                pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("verification failed: decode(encode(x)) != x")


if __name__ == "__main__":
    main()
