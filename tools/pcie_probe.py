import torch, time
n=256<<20
h=torch.empty(n,dtype=torch.uint8).pin_memory(); d=torch.empty(n,dtype=torch.uint8,device='cuda')
for name,fn in (("H2D",lambda: d.copy_(h,non_blocking=True)),("D2H",lambda: h.copy_(d,non_blocking=True))):
    fn(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    print(name, "%.1f GB/s"%(n/dt/1e9))
s1,s2=torch.cuda.Stream(),torch.cuda.Stream()
h2=torch.empty(n,dtype=torch.uint8).pin_memory(); d2=torch.empty(n,dtype=torch.uint8,device='cuda')
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s1): d.copy_(h,non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2,non_blocking=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
print("both directions at once: %.1f GB/s each"%(n/dt/1e9))
import numpy as np
a=np.empty(n,dtype=np.uint8); b=np.empty(n,dtype=np.uint8); a[:]=1
t=time.perf_counter(); b[:]=a; dt=time.perf_counter()-t; print("1-thread host memcpy %.1f GB/s"%(n/dt/1e9))
