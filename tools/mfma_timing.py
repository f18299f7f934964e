#!/usr/bin/env python
"""Phase times inside the 9-tap MFMA convolution body during one stage of the RED plane loop (library built with
-DSMVS_TIMING: tools/ab_build.sh timing -DSMVS_TIMING):  SMVS_LIB_PATH=gpurun_ab/timing.so python tools/mfma_timing.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satmvs_amd import rpc_synth, _lib
from satmvs_amd.modules.module import slice_RED_Regularization
from satmvs_amd.networks.casred import compute_depth_when_pred
_lib.load()
lib = ctypes.CDLL(os.environ["SMVS_LIB_PATH"])
dev = torch.device("cuda:0")
V = 3
for C, H, W, D, s in ((32, 96, 192, 48, 4), (16, 192, 384, 32, 2)):
    torch.manual_seed(0)
    reg = slice_RED_Regularization(C, 8).to(dev).eval()
    feats = [torch.randn(1, C, H, W, device=dev) for _ in range(V)]
    proj = torch.from_numpy(rpc_synth.rescale_rpc(rpc_synth.make_view_rpcs(V, 384, 768, seed=0)[None], s)).to(dev)
    dv = torch.linspace(0, 400, D, device=dev).view(1, D, 1, 1).expand(1, D, H, W).contiguous()
    buf = (ctypes.c_ulonglong * 64)()
    with torch.no_grad():
        for _ in range(3):
            compute_depth_when_pred(feats, proj, dv, D, reg, "rpc", False)
        lib.smvs_debug_mfma_timing(buf)
        for _ in range(3):
            compute_depth_when_pred(feats, proj, dv, D, reg, "rpc", False)
        lib.smvs_debug_mfma_timing(buf)
    print("stage %dx%d" % (W, H))
    for k in range(8):
        n = buf[k * 8]
        if n:
            print("  Cout %3d: waves(0) %6d | first operands %7.0f | K loop %7.0f | LDS reduce %6.0f | epilogue %6.0f | total %7.0f clocks" % (
                k * 32, n, buf[k*8+1] / n, buf[k*8+2] / n, buf[k*8+3] / n, buf[k*8+4] / n, buf[k*8+5] / n))
