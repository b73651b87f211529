#!/usr/bin/env python3
"""Go / no-go study for a fast-compress design without the 32 KB LDS table per chain:
a pre-pass writes, for every position p of a block, delta[p] = distance to the previous position with the same LZ4 hash (what
hc_build_kernel already computes for HC); the greedy parse then needs no table -- only to know which positions liblz4 has INSERTED
(an 8 KB bitmap per 64 KiB block, so ~20 chains per CU instead of 5): the table entry liblz4 would find for a probe at p is the
most recent INSERTED position on p's chain.  The price is the walk: every position liblz4 did NOT insert (the insides of matches)
that shares the hash lies on the chain in front of it, and each hop is a dependent load.
This script replays liblz4's greedy parse (acceleration 1, byU16 table) in plain Python on 64 KiB blocks, records
which positions are inserted, and counts for every PROBE how many chain nodes are visited before the entry liblz4 uses (hops = visited
non-inserted nodes; 0 = the first node is the entry).  A 64-lane step costs the slowest lane's walk, so the distribution of the
MAXIMUM over the probes of a step (all probes up to and including the one that hits) is what matters.
usage: chain_hops.py  -> prints a table; results recorded in profiles/r04_compress_study.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O

def h4(b, p):   # LZ4_hash4, byU16: 13 bits
    v = b[p] | (b[p + 1] << 8) | (b[p + 2] << 16) | (b[p + 3] << 24)
    return ((v * 2654435761) & 0xFFFFFFFF) >> 19

def study(name, data):
    n = len(data)
    mflimit = n - 12
    matchlimit = n - 5
    table = {}            # hash -> position (liblz4's table)
    inserted = bytearray(n)
    prev_same = [-1] * n  # previous position with the same hash (every position: the chain of the pre-pass)
    last = {}
    for p in range(0, n - 3):
        h = h4(data, p)
        prev_same[p] = last.get(h, -1)
        last[h] = p
    hops_per_probe = []   # visited non-inserted nodes before the node liblz4's table holds (or the chain's end)
    step_max = []         # max of the above over the probes of one search (one "step" of the GPU finder)
    ip = 0
    anchor = 0
    table[h4(data, 0)] = 0; inserted[0] = 1
    ip = 1
    steps = seqs = 0
    def walk(p):
        q = prev_same[p]; hops = 0
        while q >= 0 and not inserted[q]:
            hops += 1; q = prev_same[q]
        return hops, q
    while True:
        # find a match (LZ4_compress_generic, byU16: no distance check needed in a 64 KiB block)
        searchMatchNb = 1 << 6
        fwd = ip
        cur_max = 0
        found = False
        while True:
            ipc = fwd
            step = searchMatchNb >> 6; searchMatchNb += 1
            fwd = ipc + step
            if fwd > mflimit:
                break
            h = h4(data, ipc)
            hops, q = walk(ipc)
            want = table.get(h, -1)
            assert q == want, (name, ipc, q, want)   # the walk finds exactly liblz4's table entry
            hops_per_probe.append(hops); cur_max = max(cur_max, hops)
            table[h] = ipc; inserted[ipc] = 1
            if want >= 0 and data[want:want + 4] == data[ipc:ipc + 4]:
                found = True
                break
        if not found:
            break
        steps += 1; step_max.append(cur_max)
        ip = ipc; match = want
        while ip > anchor and match > 0 and data[ip - 1] == data[match - 1]:
            ip -= 1; match -= 1
        while True:
            # extend
            ml = 4
            while ip + ml < matchlimit and data[ip + ml] == data[match + ml]:
                ml += 1
            ip += ml; anchor = ip; seqs += 1
            if ip >= mflimit:
                break
            table[h4(data, ip - 2)] = ip - 2; inserted[ip - 2] = 1
            h = h4(data, ip)
            hops, q = walk(ip)
            want = table.get(h, -1)
            assert q == want
            hops_per_probe.append(hops)
            table[h] = ip; inserted[ip] = 1
            if want >= 0 and data[want:want + 4] == data[ip:ip + 4]:
                match = want; steps += 1; step_max.append(hops)
                continue
            break
        if ip >= mflimit:
            break
        ip += 1
    hp = sorted(hops_per_probe); sm = sorted(step_max)
    ins = sum(inserted) / n
    def pct(a, q): return a[min(len(a) - 1, int(q * len(a)))]
    print("%-18s seq %5d  inserted %4.1f %% of positions | hops per probe: mean %.2f  p50 %d  p90 %d  p99 %d  max %d | slowest probe of a step: mean %.2f  p50 %d  p90 %d  p99 %d"
          % (name, seqs, 100 * ins, sum(hp) / len(hp), pct(hp, .5), pct(hp, .9), pct(hp, .99), hp[-1], sum(sm) / max(len(sm), 1), pct(sm, .5), pct(sm, .9), pct(sm, .99)))

book = open(os.path.join(ROOT, "tests/golden/book1_200000.bin"), "rb").read()
for name, d in (("App. F 64 KiB", O.gen_block(65536, 0)), ("App. F 64 KiB #2", O.gen_block(65536, 7)), ("book1[:65536]", book[:65536]),
                ("geo[:65536]", open(os.path.join(ROOT, "tests/golden/geo_65536.bin"), "rb").read()),
                ("pic[:65536]", open(os.path.join(ROOT, "tests/golden/pic_65536.bin"), "rb").read())):
    study(name, d)
