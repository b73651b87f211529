#!/usr/bin/env python3
"""What the decoder's parallel wave loop does on the GPU (needs a -DLZ4HIP_RING_DBG build: tools/build_variant.sh rdbg -DLZ4HIP_RING_DBG, then
LZ4HIP_LIBRARY=lz4-java_amd/variants/rdbg.so): wave_stats.py <workload>  -- trips, sequences per trip, rounds, one-sequence steps per block"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
amd = importlib.import_module("lz4-java_amd")
L = amd.lib()
L.lz4hip_dbg_ring_stats.restype = C.c_int; L.lz4hip_dbg_ring_stats.argtypes = [C.POINTER(C.c_ulonglong)]
sys.argv = [sys.argv[0], sys.argv[1], "64:5:0:0"]
out = (C.c_ulonglong * 8)()
L.lz4hip_dbg_ring_stats(out)
exec(open(os.path.join(ROOT, "tools", "ring_matrix.py")).read())
torch.cuda.synchronize()
L.lz4hip_dbg_ring_stats(out)
trips, seqs, rounds, single, hungry, T, two, entries = (int(x) for x in out)
print("3 launches: loop entries %d, trips %d, sequences decoded in trips %d (%.2f per trip; %.2f starts found per trip), copy rounds %d (%.2f per trip), "
      "trips with two windows %d, one-sequence steps %d, trips that found the stream ring short %d"
      % (entries, trips, seqs, seqs / max(trips, 1), T / max(trips, 1), rounds, rounds / max(trips, 1), two, single, hungry))
