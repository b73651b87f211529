#!/usr/bin/env python3
"""Builds profiles/traffic.json from the PMC pass databases written by tools/traffic_passes.sh.
usage: traffic_json.py <passes dir> <blocks_per_gpu> <block_bytes> <source tag>"""
import glob, hashlib, json, os, sqlite3, sys
d, n, blk, tag = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
text_dir = sys.argv[5] if len(sys.argv) > 5 else None   # passes of tools/gpu_text_legs.py (the real-text legs alone)
vals = {}
full = {}   # every lz4hip kernel by its full (template) name: "decode_kernel<8, true, 2, false>", ...
def norm(k):
    k = k.split("(")[0].replace("void ", "").replace("lz4hip::", "").strip()
    return k
for db in glob.glob(os.path.join(d, "*", "pmc_results.db")):
    con = sqlite3.connect(db)
    rows = {}
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    order = "dispatch_id" if "dispatch_id" in cols else "rowid"
    for k, c, v in con.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%%lz4hip%%' order by %s" % order):
        rows.setdefault((k, c), []).append(v)
    # Kernels that serve several workloads of the bench line (fast compress: headline, end_to_end chunks, frame blocks, book1, 4 MiB
    # blocks; the 4-lane staged safe decoder: headline and book1) are represented by their FIRST launches in dispatch order -- the
    # warm-up and timed steps of the headline (tools/traffic_passes.sh runs --warmup 1 --steps 2: three launches).  The others
    # have one workload: MEDIAN over their dispatches.
    HEADLINE_FIRST = ("compress_fast_v2w_cu_kernel", "compress_fast_ms_cu_kernel", "decode_kernel<4, true, 0, true>")
    # The 8-lane deep decoder serves the chunks of the end_to_end leg (1024 blocks each, many launches) AND the configs[2] launch
    # (16384 x 4 MiB, the biggest by far): it is represented by its BIGGEST dispatch.
    # compress_fast_v2wp_cu_kernel (blocks of 65547 bytes .. 4 MiB) is launched with every fast compress and returns at once when the
    # batch has no such block: its real launch is the biggest one (compress_4MiB).
    # decode_wave_kernel<8, 16384, ..> (a wavefront per block): the configs2_shard8 launch (2048 x 4 MiB) and small launches of 64 KiB
    # blocks: its biggest dispatch is the shard
    BIGGEST = ("decode_deep_kernel<8, true>", "decode_ring_kernel<4, 2048, true>", "compress_fast_v2wp_cu_kernel", "decode_wave_kernel<8, 16384", "decode_wave_kernel<16, 8192", "decode_pair_kernel", "decode_trio_kernel")
    for (k, c), vs in rows.items():
        if any(h in k for h in HEADLINE_FIRST):
            vs = vs[:3]
        vs = sorted(vs)
        v = vs[len(vs) // 2] if len(vs) % 2 else 0.5 * (vs[len(vs) // 2 - 1] + vs[len(vs) // 2])
        if any(h in k for h in BIGGEST):
            v = vs[-1]
        full.setdefault(norm(k), {})[c] = v
        if "compress_fast_v2wp_cu_kernel" in k:
            # ... and its configs2_shard8_compress launches (2048 of the 16384 blocks): the dispatches at about an eighth of the biggest
            # (real_book1_4MiB -- 2560 blocks of text -- moves two thirds of it; the launches without such a block next to nothing)
            sh = [x for x in vs if 0.08 * vs[-1] <= x <= 0.2 * vs[-1]]
            if sh:
                full.setdefault(norm(k) + "@shard8", {})[c] = sh[len(sh) // 2]
        if "decode_kernel" in k and "decode_kernel<4, true, 0, true>" not in k:
            continue   # the legacy "decode_kernel" key below is the headline launch (65536 x 64 KiB: 4 lanes, safe, staged) only
        key = None
        for name in ("compress_fast_v2w_cu_kernel", "compress_fast_v2_cu_kernel", "compress_fast_cu_kernel", "compress_fast_ms_cu_kernel", "decode_kernel", "hc_parse_kernel", "hc_build_kernel", "xxh_multi_kernel"):
            if name in k:
                key = name
                break
        if key:
            vals.setdefault(key, {})[c] = v
def kernel_source_hash():   # same function as bench.py: marks which kernel sources the passes measured
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lz4-java_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


out = {"blocks_per_gpu": n, "block_bytes": blk, "source": tag, "kernel_source_hash": kernel_source_hash(),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 "
                 "(gfx950: FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); median over the launches of a kernel (kernels shared by several workloads: over the headline's launches, the first three in dispatch order; decode_deep_kernel<8, true>: its biggest launch, configs[2])",
       "raw": vals}
for key, c in vals.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out[key] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
# cross-check of the doubling rule: read bytes from the request-size counters of the same passes (128/64/32-byte requests)
out["kernels_read_bytes_by_request_size"] = {
    k: int(128 * c["TCC_EA0_RDREQ_128B_sum"] + 64 * c["TCC_EA0_RDREQ_64B_sum"] + 32 * c["TCC_EA0_RDREQ_32B_sum"])
    for k, c in full.items() if all(x in c for x in ("TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_32B_sum"))}
out["kernels"] = {k: int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024) for k, c in full.items() if "FETCH_SIZE" in c and "WRITE_SIZE" in c}
if text_dir:
    # the text legs' launches, from passes of their own command: every kernel by its BIGGEST dispatch (the five-pair and the packed
    # kernel are both launched with every fast compress and the one that has no block of its kind returns at once)
    tk = {}
    for db in glob.glob(os.path.join(text_dir, "*", "pmc_results.db")):
        con = sqlite3.connect(db)
        for k, c, v in con.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%%lz4hip%%'"):
            e = tk.setdefault(norm(k), {})
            e[c] = max(e.get(c, 0), v)
    tb = {k: int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024) for k, c in tk.items() if "FETCH_SIZE" in c and "WRITE_SIZE" in c}
    def pick(*subs):
        vs = [v for k, v in tb.items() if any(s_ in k for s_ in subs)]
        return int(sum(vs)) if vs else None
    out["text"] = {"real_book1": {"compress": pick("compress_fast_v2w_cu_kernel", "compress_fast_ms_cu_kernel"), "decode": pick("decode_kernel<4, true, 0, true>", "decode_wave_kernel<16, 8192, 1024, true, 5>", "decode_route_kernel")},   # (routed by density since round 6: the wave kernel does the work, the staged loop's launch returns at once)
                   "real_book1_4MiB": {"compress": pick("compress_fast_v2wp_cu_kernel")},
                   "kernels": tb, "source": "tools/gpu_text_legs.py under tools/traffic_passes.sh <dir> 2 text"}
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out["kernels"]))
