#!/usr/bin/env python3
"""What a narrower fingerprint costs the byU16 finder (64 KiB blocks): the headline's limit is FIVE chains per CU -- five 32 KB
tables of {position 16 bits, fingerprint 16 bits} -- and round 4 measured throughput linear in the chains a CU holds
(profiles/r04_compress_study.txt).  A table of pos16[8192] + fp8[8192] is 24 KB = six chains (there are no 16-bit LDS atomics:
the commit would be plain stores + a re-read that takes the place of the atomic's returned value).  The price is false tentative
hits: probes whose bucket holds a position with the same k-bit fingerprint but different bytes; each is one more candidate round
trip (if the loop handles it in place) or a loop exit (today).
This script replays liblz4's greedy parse (acceleration 1, byU16) in plain Python and counts, per search, the false tentative probes
in front of the real hit for fingerprints of 16 / 12 / 10 / 8 / 6 / 4 bits (the top k of the kernel's 16: bits 3..18 of the bucket product), plus the searches that end without a hit within 63 probes.
usage: fp_width.py -> a table; recorded in profiles/r04_compress_study.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O

WIDTHS = (16, 12, 10, 8, 6, 4)


def prod(b, p):
    v = b[p] | (b[p + 1] << 8) | (b[p + 2] << 16) | (b[p + 3] << 24)
    return (v * 2654435761) & 0xFFFFFFFF


def study(name, data):
    n = len(data)
    mflimit = n - 12
    matchlimit = n - 5
    pr = [prod(data, p) for p in range(n - 3)]
    table = [0] * 8192          # every bucket: position 0 (liblz4's zeroed table)
    steps = probes = 0
    false_t = {k: 0 for k in WIDTHS}           # false tentative probes in front of the hit, all searches
    steps_with = {k: 0 for k in WIDTHS}        # searches with at least one
    ip = 1
    anchor = 0
    while True:
        searchMatchNb = 1 << 6
        fwd = ip
        found = False
        ft = {k: 0 for k in WIDTHS}
        while True:
            ipc = fwd
            step = searchMatchNb >> 6; searchMatchNb += 1
            fwd = ipc + step
            if fwd > mflimit:
                break
            h = pr[ipc] >> 19
            cand = table[h]
            table[h] = ipc
            probes += 1
            if data[cand:cand + 4] == data[ipc:ipc + 4]:
                found = True
                break
            x = ((pr[ipc] >> 3) ^ (pr[cand] >> 3)) & 0xFFFF      # the kernel's 16 fingerprint bits (bits 3..18 of the product)
            for k in WIDTHS:
                if x >> (16 - k) == 0:      # the TOP k of the 16 bits (the low bits of a multiplicative hash see only the low input bits:
                    ft[k] += 1              # with the low k, geo's float data has 31 % of its searches meet a false tentative at 12 bits)
        if not found:
            break
        steps += 1
        for k in WIDTHS:
            false_t[k] += ft[k]
            steps_with[k] += 1 if ft[k] else 0
        ip = ipc; match = cand
        while ip > anchor and match > 0 and data[ip - 1] == data[match - 1]:
            ip -= 1; match -= 1
        while True:
            ml = 4
            while ip + ml < matchlimit and data[ip + ml] == data[match + ml]:
                ml += 1
            ip += ml; anchor = ip
            if ip >= mflimit:
                break
            table[pr[ip - 2] >> 19] = ip - 2
            h = pr[ip] >> 19
            cand = table[h]
            table[h] = ip
            if data[cand:cand + 4] == data[ip:ip + 4]:
                match = cand
                steps += 1
                continue
            break
        if ip >= mflimit:
            break
        ip += 1
    print("%-22s searches %6d  probes per search %5.1f   false tentative probes per search (searches with one):" % (name, steps, probes / max(steps, 1)))
    print("    " + "   ".join("fp%-2d %.4f (%4.1f %%)" % (k, false_t[k] / max(steps, 1), 100.0 * steps_with[k] / max(steps, 1)) for k in WIDTHS))
    return {k: false_t[k] / max(steps, 1) for k in WIDTHS}


if __name__ == "__main__":
    g = os.path.join(ROOT, "tests", "golden")
    book = open(os.path.join(g, "book1_200000.bin"), "rb").read()
    sets = [("App. F 64 KiB", O.gen_block(65536, 0)), ("App. F 64 KiB (#2)", O.gen_block(65536, 12345)), ("book1[:65536]", book[:65536]),
            ("geo[:65536]", open(os.path.join(g, "geo_65536.bin"), "rb").read()), ("pic[:65536]", open(os.path.join(g, "pic_65536.bin"), "rb").read())]
    res = {}
    for name, d in sets:
        res[name] = study(name, d)
    print()
    print("chains x 1 / (1 + false tentatives per search) -- each false tentative priced at one more step of the chain, commit + re-read at 2 % --")
    for name in res:
        f = res[name]
        print("  %-20s five chains fp16: %.3f   six chains fp8: %.3f   six chains fp6 (22-bit entries do not pack): n/a" % (name, 5 / (1 + f[16]), 6 * 0.98 / (1 + f[8])))
