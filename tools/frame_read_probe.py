"""Developer probe: the device-side container READ path on a frame body of n 64 KiB blocks assembled on the device: frame_read_probe.py <n>"""
import importlib, sys, torch
sys.path.insert(0,'/root/repo')
amd = importlib.import_module("lz4-java_amd")
dev=torch.device("cuda:0"); n=int(sys.argv[1]); blk=65536; cap=amd.maxCompressedLength(blk)
src=torch.empty(n*blk,dtype=torch.uint8,device=dev); amd.DeviceBatch.gen_blocks(src,blk,blk,n)
comp=torch.empty(n*cap,dtype=torch.uint8,device=dev); total_t=torch.zeros(1,dtype=torch.int64,device=dev)
amd.DeviceBatch.container_blocks(0, src, blk, comp, total_t); torch.cuda.synchronize(); tot=int(total_t.item())
L=amd.lib(); wsb=L.lz4hip_container_decode_workspace_bytes(n)
ws=torch.empty(wsb,dtype=torch.uint8,device=dev); sizes=torch.zeros(n,dtype=torch.int32,device=dev); info=torch.zeros(5,dtype=torch.int64,device=dev)
back=torch.zeros(n*blk,dtype=torch.uint8,device=dev)
for _ in range(3):
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    rc=L.lz4hip_container_decode_dev(0,0,comp.data_ptr(),tot,blk,back.data_ptr(),blk,n,sizes.data_ptr(),info.data_ptr(),ws.data_ptr(),wsb,0,torch.cuda.current_stream().cuda_stream)
    b.record(); torch.cuda.synchronize()
    print("container_decode_dev: %.3f ms = %.1f GB/s of output" % (a.elapsed_time(b), n*blk/a.elapsed_time(b)/1e6))
print(rc, tot, info.cpu().tolist(), int((sizes!=blk).sum()), bool(torch.equal(back,src)))
bad=(back.view(n,blk)!=src.view(n,blk)).any(dim=1).nonzero().flatten()[:5].tolist(); print("bad blocks",bad, sizes[:4].tolist())
