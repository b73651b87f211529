#!/usr/bin/env python3
"""Raw link rates of the box: pinned H2D, D2H, both at once (two streams), for 16 / 128 / 512 MiB transfers."""
import torch, time
dev = torch.device("cuda:0")
for mb in (16, 128, 512):
    n = mb << 20
    h1 = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    d1 = torch.empty(n, dtype=torch.uint8, device=dev); d2 = torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def run(f, reps=6):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        return best
    def h2d():
        with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
    def d2h():
        with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
    def both(): h2d(); d2h()
    a, b, c = run(h2d), run(d2h), run(both)
    print("%4d MiB: H2D %.1f GB/s  D2H %.1f GB/s  both at once %.1f + %.1f GB/s" % (mb, n / a / 1e9, n / b / 1e9, n / c / 1e9, n / c / 1e9))
