#!/usr/bin/env python3
"""Builds profiles/traffic.json from the PMC pass databases written by tools/traffic_passes.sh.
usage: traffic_json.py <passes dir> <blocks_per_gpu> <block_bytes> <source tag>"""
import glob, hashlib, json, os, sqlite3, sys
d, n, blk, tag = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
vals = {}
for db in glob.glob(os.path.join(d, "*", "pmc_results.db")):
    con = sqlite3.connect(db)
    for k, c, v in con.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%lz4hip%' group by kernel_name, counter_name"):
        key = None
        for name in ("compress_fast_v2_cu_kernel", "compress_fast_cu_kernel", "compress_fast_ms_cu_kernel", "decode_kernel", "hc_parse_kernel", "hc_build_kernel", "xxh_multi_kernel"):
            if name in k:
                key = name
                break
        if key:
            vals.setdefault(key, {})[c] = v
def kernel_source_hash():   # same function as bench.py: marks which kernel sources the passes measured
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lz4-java_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


out = {"blocks_per_gpu": n, "block_bytes": blk, "source": tag, "kernel_source_hash": kernel_source_hash(),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 "
                 "(gfx950: FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); average per launch",
       "raw": vals}
for key, c in vals.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out[key] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: out[k] for k in out if k.endswith("kernel")}))
