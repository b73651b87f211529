#!/usr/bin/env python3
"""What the decoder's ring loop does on the GPU (needs a -DLZ4HIP_RING_DBG build: tools/build_variant.sh rdbg -DLZ4HIP_RING_DBG,
LZ4HIP_LIBRARY=.../variants/rdbg.so): ring_stats.py <workload> <lanes> <ring>  -- trips / stalls / exits per block"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
amd = importlib.import_module("lz4-java_amd")
L = amd.lib()
L.lz4hip_dbg_ring_stats.restype = C.c_int; L.lz4hip_dbg_ring_stats.argtypes = [C.POINTER(C.c_ulonglong)]
sys.argv = [sys.argv[0], sys.argv[1], "%s:3:0:%s" % (sys.argv[2], sys.argv[3])]
out = (C.c_ulonglong * 8)()
L.lz4hip_dbg_ring_stats(out)
exec(open(os.path.join(ROOT, "tools", "ring_matrix.py")).read())
torch.cuda.synchronize()
L.lz4hip_dbg_ring_stats(out)
t, st, fr, seeds, wait, entries = out[0], out[1], out[2], out[3], out[4], out[5]
print("wave-trips %d (block-trips / wave-trips = %.2f)" % (out[6], t / max(out[6], 1)))
print("3 launches: trips %d, stalled %d (%.1f %%), frozen %d, waiting for another block %d (%.1f %%), (re-)seeds %d, loop entries %d"
      % (t, st, 100.0 * st / max(t, 1), fr, wait, 100.0 * wait / max(t, 1), seeds, entries))
