#!/usr/bin/env python3
"""Regenerates tests/golden/*: run HERE (where /root/reference exists), outputs are committed.

* calgary slices (the only real-world data in the reference: src/test-resources/calgary/{book1,geo,pic},
  used by LZ4Test.java:335-348) -- small prefixes only, as input fixtures for the GPU box where
  /root/reference does not exist;
* golden.json: for every named input the outputs of the REFERENCE ITSELF (the prebuilt
  liblz4-java.so = liblz4 1.9.3 + xxhash 0.6.5 that LZ4Factory.nativeInstance() loads): compressed
  size + sha256 of LZ4_compress_default and LZ4_compress_HC(9), XXH32/XXH64 with seeds 0 and
  0x9747b28c, plus decoder return codes on the reference's malformed vectors (LZ4Test.java:350-419).
  SURVEY.md Appendix E is a subset of this table.
"""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O

CAL = "/root/reference/src/test-resources/calgary/"
R = O.ref()
assert "oracle/_ref" in R.path, "golden vectors must come from the reference's own library"


def main():
    book1, geo, pic = (open(CAL + n, "rb").read() for n in ("book1", "geo", "pic"))
    slices = {"book1_200000.bin": book1[:200000], "geo_65536.bin": geo[:65536], "pic_65536.bin": pic[:65536]}
    for name, data in slices.items():
        open(os.path.join(HERE, name), "wb").write(data)
    inputs = {
        "book1": book1, "geo": geo, "pic": pic,
        "book1[:65536]": book1[:65536], "geo[:65536]": geo[:65536], "pic[:65536]": pic[:65536],
        "book1[:65546]": book1[:65546], "book1[:65547]": book1[:65547], "book1[:200000]": book1[:200000],
        "zeros65536": bytes(65536), "len12": book1[:12], "len13": book1[:13], "len0": b"",
        "selftest": b"abcd      abcdefghij", "readme": b"12345345234572",
        "gen_block(65536,0)": O.gen_block(65536, 0), "gen_block(65536,1)": O.gen_block(65536, 1),
        "gen_block(1<<20,3,win=4096)": O.gen_block(1 << 20, 3, win=4096),
    }
    table = {}
    for name, data in inputs.items():
        fast = R.compress_fast(data)
        hc = R.compress_hc(data, 9)
        hc10, hc12 = R.compress_hc(data, 10), R.compress_hc(data, 12)
        table[name] = {
            "n": len(data), "md5": hashlib.md5(data).hexdigest(),
            "fast_size": len(fast), "fast_sha256": hashlib.sha256(fast).hexdigest(),
            "hc9_size": len(hc), "hc9_sha256": hashlib.sha256(hc).hexdigest(),
            "hc10_size": len(hc10), "hc10_sha256": hashlib.sha256(hc10).hexdigest(),
            "hc12_size": len(hc12), "hc12_sha256": hashlib.sha256(hc12).hexdigest(),
            "xxh32_seed0": "%08x" % R.xxh32(data, 0), "xxh64_seed0": "%016x" % R.xxh64(data, 0),
            "xxh32_seed9747b28c": "%08x" % R.xxh32(data, 0x9747b28c), "xxh64_seed9747b28c": "%016x" % R.xxh64(data, 0x9747b28c),
        }
        if len(data) <= 20:
            table[name]["fast_hex"] = fast.hex()
    vectors = []
    v0 = bytes([16, 42, 0, 0, 128] + [42] * 8)                      # LZ4Test.java:353 (offset 0)
    v1 = bytes([96, 42, 43, 44, 45, 46, 47, 5, 0])                  # LZ4Test.java:366 (ends with a match)
    cases = [(v0, 20, 13), (v1, 20, 10)]
    for i in range(1, 5):                                           # LZ4Test.java:393-397
        cases.append((v1 + bytes([i << 4] + [0] * i), 20, 20))
    cases.append((bytes([0]), 0, 0))
    cases.append((bytes([0]), 1, 0))
    for vec, cap, fast_len in cases:
        rs, ds = R.decompress_safe_raw(vec, cap)
        rf, _ = R.decompress_fast_raw(vec, fast_len)
        vectors.append({"hex": vec.hex(), "safe_cap": cap, "safe_ret": rs, "safe_out_hex": ds[:max(rs, 0)].hex(),
                        "fast_len": fast_len, "fast_ret": rf})
    bounds = {str(n): R.compress_bound(n) for n in (0, 1, 254, 255, 256, 65536, 1 << 20, 4 << 20, 0x7E000000, 0x7E000001)}
    json.dump({"generator": "tests/golden/make_golden.py", "reference_lib": "liblz4-java.so linux/amd64 (liblz4 1.9.3, xxhash 0.6.5)",
               "inputs": table, "malformed": vectors, "compress_bound": bounds,
               "gen_block_65536_0_sha256": hashlib.sha256(O.gen_block(65536, 0)).hexdigest()},
              open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(table), "inputs,", len(vectors), "malformed vectors")


if __name__ == "__main__":
    main()
