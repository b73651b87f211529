#!/usr/bin/env python3
"""Regenerates tests/golden/fast_decode_contract.json: the pinned behaviour of the BOUNDED fast decoder
(lz4hip_decompress_fast*, include/lz4hip.h "fast decoder contract") on valid, truncated, corrupted and random input.

liblz4's LZ4_decompress_fast (LZ4JNI.c:169) trusts the stream and reads wherever it points; the HIP engine never reads past the
source slot (src_cap) and never before the destination.  On VALID streams the two agree (checked here against the reference
library for every valid case); on anything else the engine's answer is DEFINED by oracle/lz4_oracle.c
lz4o_decompress_fast_bounded -- liblz4's decode loop with every out-of-slot read turned into the error at that input position --
and this file freezes it: 600 cases {stream, src_cap, dst_len} -> {return code, sha256 of the dst_len output bytes when >= 0}.
Run HERE (where oracle/_ref exists); the output is committed."""
import hashlib, json, os, random, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import oracle as O
from conftest import rnd_inputs

R = O.ref()


def main():
    book1 = open(os.path.join(HERE, "book1_200000.bin"), "rb").read()
    corpus = {"book1[:200000]": book1, "pic[:65536]": open(os.path.join(HERE, "pic_65536.bin"), "rb").read(),
              "geo[:65536]": open(os.path.join(HERE, "geo_65536.bin"), "rb").read()}
    rng = random.Random(20260924)
    cases = []
    for v in rnd_inputs(O, corpus, 4242, 600, max_n=3000):
        c = bytearray(R.compress_fast(v))
        mode, n = rng.randrange(7), len(v)
        if mode == 1:
            n = max(0, len(v) + rng.choice([-1, 1, -5, 5, -12, 12, 64]))
        elif mode == 2 and c:
            for _ in range(rng.randrange(1, 4)):
                c[rng.randrange(len(c))] = rng.randrange(256)
        elif mode == 3 and len(c) > 1:
            c = c[:rng.randrange(1, len(c))]
        elif mode == 4:
            c = c + rng.randbytes(rng.randrange(1, 20))
        elif mode == 5:
            c, n = bytearray(rng.randbytes(rng.randrange(1, 40))), rng.randrange(0, 200)
        c = bytes(c)
        cap = max(len(c) + rng.choice([0, 0, 0, 3, 16, -1, -4]), 0)
        r, d = O.decompress_fast_bounded(c, cap, n)
        e = {"hex": c.hex(), "src_cap": cap, "dst_len": n, "ret": r}
        if r >= 0:
            e["sha256"] = hashlib.sha256(d[:n]).hexdigest()
        if mode == 0 and cap >= len(c):   # valid stream, whole slot readable: the reference library itself agrees
            rr, rd = R.decompress_fast_raw(c, n)
            assert rr == r and rd[:n] == d[:n], "port and reference disagree on a valid stream"
            e["reference_agrees"] = True
        cases.append(e)
    json.dump({"what": __doc__.split("\n")[0], "generator": "tests/golden/make_fast_decode_contract.py", "cases": cases},
              open(os.path.join(HERE, "fast_decode_contract.json"), "w"), indent=0)
    print("wrote %d cases (%d valid, %d negative)" % (len(cases), sum(1 for e in cases if e.get("reference_agrees")), sum(1 for e in cases if e["ret"] < 0)))


if __name__ == "__main__":
    main()
