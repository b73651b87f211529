#!/usr/bin/env python3
"""Which ten bits make the fingerprint of the packed byU32 entries (csrc/lz4_fast_core.h PK: {position 22 bits, fingerprint 10 bits})?
Replays liblz4's greedy parse with the byU32 table (4096 buckets = bits 28..39 of the 5-byte product, candidates within 65535
bytes) and counts the false tentative probes per search for several choices of ten bits of (four bytes) * 2654435761:
bits 16..25 (what the kernel takes: (prod >> 16) & 0x3FF), bits 22..31 (the best mixed ones), bits 6..15, and 16 bits (>> 16) as
in the 64-bit entries.  usage: fp_bits_u32.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O

CHOICES = {"16 bits (>>16)": lambda p: p >> 16, "bits 16..25": lambda p: (p >> 16) & 0x3FF, "bits 22..31": lambda p: p >> 22, "bits 6..15": lambda p: (p >> 6) & 0x3FF}
P5 = 889523592379


def study(name, data):
    n = len(data)
    mflimit = n - 12
    matchlimit = n - 5
    pr = [0] * (n - 7); hb = [0] * (n - 7)
    for p in range(n - 7):
        v = data[p] | (data[p + 1] << 8) | (data[p + 2] << 16) | (data[p + 3] << 24)
        pr[p] = (v * 2654435761) & 0xFFFFFFFF
        v5 = v | (data[p + 4] << 32)
        hb[p] = (((v5 << 24) * P5) & 0xFFFFFFFFFFFFFFFF) >> 52
    table = [0] * 4096
    steps = probes = 0
    ft_all = {k: 0 for k in CHOICES}; with_ = {k: 0 for k in CHOICES}
    ip = 1; anchor = 0
    while True:
        searchMatchNb = 1 << 6
        fwd = ip; found = False
        ft = {k: 0 for k in CHOICES}
        while True:
            ipc = fwd
            step = searchMatchNb >> 6; searchMatchNb += 1
            fwd = ipc + step
            if fwd > mflimit or ipc >= n - 7:
                break
            h = hb[ipc]; cand = table[h]; table[h] = ipc; probes += 1
            if cand + 65535 < ipc:
                continue                                   # too far: no hit whatever the bytes (the kernel's distance rule)
            if data[cand:cand + 4] == data[ipc:ipc + 4]:
                found = True
                break
            for k, f in CHOICES.items():
                if f(pr[ipc]) == f(pr[cand]):
                    ft[k] += 1
        if not found:
            break
        steps += 1
        for k in CHOICES:
            ft_all[k] += ft[k]; with_[k] += 1 if ft[k] else 0
        ip = ipc; match = cand
        while ip > anchor and match > 0 and data[ip - 1] == data[match - 1]:
            ip -= 1; match -= 1
        while True:
            ml = 4
            while ip + ml < matchlimit and data[ip + ml] == data[match + ml]:
                ml += 1
            ip += ml; anchor = ip
            if ip >= mflimit or ip >= n - 7:
                break
            table[hb[ip - 2]] = ip - 2
            h = hb[ip]; cand = table[h]; table[h] = ip
            if cand + 65535 >= ip and data[cand:cand + 4] == data[ip:ip + 4]:
                match = cand; steps += 1
                continue
            break
        if ip >= mflimit or ip >= n - 7:
            break
        ip += 1
    print("%-26s searches %6d  probes per search %5.1f" % (name, steps, probes / max(steps, 1)))
    print("    false tentative probes per search (searches with one): " + "   ".join("%s %.4f (%4.1f %%)" % (k, ft_all[k] / max(steps, 1), 100.0 * with_[k] / max(steps, 1)) for k in CHOICES))


if __name__ == "__main__":
    g = os.path.join(ROOT, "tests", "golden")
    book = open(os.path.join(g, "book1_200000.bin"), "rb").read()
    geo = open(os.path.join(g, "geo_65536.bin"), "rb").read(); pic = open(os.path.join(g, "pic_65536.bin"), "rb").read()
    for name, d in (("App. F 192 KiB win 4096", O.gen_block(196608, 1 << 24, win=4096)), ("App. F 192 KiB win 65535", O.gen_block(196608, 5)),
                    ("book1[:200000]", book), ("geo x 2 (131072)", geo + geo), ("pic x 2", pic + pic)):
        study(name, d)
