#!/usr/bin/env python3
"""Which stage of the trio loop waits for which (needs a -DLZ4HIP_RING_DBG build: tools/build_variant.sh rdbg -DLZ4HIP_RING_DBG, then
LZ4HIP_LIBRARY=lz4-java_amd/variants/rdbg.so): trio_stats.py <workload>  -- naps by reason over the launches of tools/ring_matrix.py <workload> 64:8:0:0"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
amd = importlib.import_module("lz4-java_amd")
L = amd.lib()
L.lz4hip_dbg_ring_stats.restype = C.c_int; L.lz4hip_dbg_ring_stats.argtypes = [C.POINTER(C.c_ulonglong)]
sys.argv = [sys.argv[0], sys.argv[1], "64:8:0:0"]
out = (C.c_ulonglong * 8)()
L.lz4hip_dbg_ring_stats(out)
exec(open(os.path.join(ROOT, "tools", "ring_matrix.py")).read())
torch.cuda.synchronize()
L.lz4hip_dbg_ring_stats(out)
n = [int(x) for x in out]
print("naps (s_sleep 1 each) over 4 launches: scanner waits for the planner (scan queue full) %d, scanner waits for stream room (the copier's IPDONE) %d, "
      "planner waits for the scanner (no window) %d, planner waits for the copier (mailbox full) %d, copier waits for the planner (no message) %d, copier waits for SACK %d, other %d"
      % (n[0], n[1], n[2], n[3], n[4], n[5], n[7]))
