#!/usr/bin/env python
"""Per-wave phase times of the boxed path of the cost-volume backward kernel (library built with -DSMVS_BWD_TIMING):
    AB_SRC=costvol_bwd.hip tools/ab_build.sh bwdtiming -DSMVS_BWD_TIMING && SMVS_LIB_PATH=gpurun_ab/bwdtiming.so python tools/wave_timing_bwd.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satmvs_amd import _lib
import bench
_lib.load()
lib = ctypes.CDLL(os.environ["SMVS_LIB_PATH"])
dev = torch.device("cuda:0")
V, C, D, H, W = bench.WORKLOADS["cfg2_rpc_3view_768x384x64_c32"]
feats, rpc, depth = bench.make_inputs(V, C, D, D, 0, H, W, dev)
g = torch.randn((1, C, D, H, W), device=dev)
gref = torch.zeros_like(feats[0]); gsrc = [torch.zeros_like(f) for f in feats[1:]]
st = _lib.current_stream(dev)
srcs, gs = _lib.ptr_array(feats[1:]), _lib.ptr_array(gsrc)
def step():
    _lib.call("smvs_costvol_bwd", 0, _lib.ptr(g), _lib.ptr(feats[0]), srcs, V - 1, _lib.ptr(rpc), _lib.ptr(depth), 1, _lib.ptr(gref), gs, 1, C, D, H, W, st)
for _ in range(5): step()
buf = (ctypes.c_ulonglong * 8)()
lib.smvs_debug_timing_bwd(buf, 1)
n = 10
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n): step()
b.record(); torch.cuda.synchronize()
lib.smvs_debug_timing_bwd(buf, 1)
w = buf[7]
print("%.3f ms per launch (with stamps); boxed waves %d (%d launches)" % (a.elapsed_time(b) / n, w, n))
for name, i in (("geometry + set-up", 0), ("top-of-channel waits", 1), ("flush", 2), ("issue of loads / boxes", 3), ("plane loop", 4), ("whole wave", 5)):
    print("%-24s %9.0f clocks per wave" % (name, buf[i] / max(w, 1)))
print("plane-view pairs whose 32-lane rows are single runs of cells: %.1f of %d per boxed wave; waves where all are: %.1f %%"
      % ((buf[6] & 0xffffffff) / max(w, 1), 16, 100.0 * (buf[6] >> 32) / max(w, 1)))
