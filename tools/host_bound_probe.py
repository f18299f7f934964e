#!/usr/bin/env python
"""Is the RED plane loop bound by the host's enqueue rate or by the GPU?  Times one stage of compute_depth_when_pred:
host return of the native call vs GPU completion."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd import rpc_synth
from satmvs_amd.modules.module import slice_RED_Regularization
from satmvs_amd.networks.casred import compute_depth_when_pred
dev = torch.device("cuda:0")
V = 3
for C, H, W, D, s in ((32, 96, 192, 48, 4), (16, 192, 384, 32, 2), (8, 384, 768, 8, 1)):
    torch.manual_seed(0)
    reg = slice_RED_Regularization(C, 8).to(dev).eval()
    feats = [torch.randn(1, C, H, W, device=dev) for _ in range(V)]
    proj = torch.from_numpy(rpc_synth.rescale_rpc(rpc_synth.make_view_rpcs(V, 384, 768, seed=0)[None], s)).to(dev)
    dv = torch.linspace(0, 400, D, device=dev).view(1, D, 1, 1).expand(1, D, H, W).contiguous()
    with torch.no_grad():
        for _ in range(3):
            compute_depth_when_pred(feats, proj, dv, D, reg, "rpc", False)
        torch.cuda.synchronize()
        th, tg = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            compute_depth_when_pred(feats, proj, dv, D, reg, "rpc", False)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            th.append(t1 - t0); tg.append(t2 - t0)
    print("stage %dx%d C=%d D=%d: host returns after %.2f ms, GPU done after %.2f ms  (%.0f / %.0f us per plane)" % (
        W, H, C, D, min(th) * 1e3, min(tg) * 1e3, min(th) / D * 1e6, min(tg) / D * 1e6))
