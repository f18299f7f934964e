#!/usr/bin/env python
"""Per-wave phase times of the staged cost-volume kernel (library built with -DSMVS_TIMING):
    SMVS_LIB_PATH=gpurun_ab/timing.so [SMVS_WORKLOAD=cfg4_rpc_5view_1536x768x8_c32] python tools/wave_timing.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satmvs_amd import _lib
import bench
_lib.load()
lib = ctypes.CDLL(os.environ["SMVS_LIB_PATH"])
dev = torch.device("cuda:0")
wl = os.environ.get("SMVS_WORKLOAD", "cfg2_rpc_3view_768x384x64_c32")
V, C, D, H, W = bench.WORKLOADS[wl]
feats, rpc, depth = bench.make_inputs(V, C, D, D, 0, H, W, dev)
if wl in bench.SIDE_HEIGHTS:
    lo, hi = bench.SIDE_HEIGHTS[wl]
    depth = torch.linspace(lo, hi, D, dtype=torch.float32).view(1, D, 1, 1).expand(1, D, H, W).contiguous().to(dev)
_lib.set_arith("exact")          # the stamps live in costvol.hip's copy of the kernels (the exact instances)
out = torch.empty((1, C, D, H, W), dtype=torch.float32, device=dev)
st = _lib.current_stream(dev)
# SMVS_BENCH_TRIVARIATE=1: smvs_rpc_costvol_fwd (no folded plane coefficients), else fold + smvs_rpc_costvol_fwd_pc
step = bench.tile_build(_lib, feats, rpc, depth, out, V, C, D, H, W, st, plane_coef=os.environ.get("SMVS_BENCH_TRIVARIATE") != "1")[1]
for _ in range(20): step()
buf = (ctypes.c_ulonglong * 12)()
lib.smvs_debug_timing(buf, 1)
n = 20
for _ in range(n): step()
lib.smvs_debug_timing(buf, 1)
w = buf[7]
names = ["geometry", "box+setup", "pair loop", " vmcnt waits", " store issue", " dma issue", "", "", " heights+check", " scales+ref view", " source views+taps"]
print("waves %d staged + %d fallback (%d launches)" % (w, buf[6], n))
for i in (0, 8, 9, 10, 1, 2, 3, 4, 5):
    print("%-20s %9.0f clocks per wave" % (names[i], buf[i] / w))
