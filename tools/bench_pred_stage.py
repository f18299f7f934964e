#!/usr/bin/env python
"""Per-stage timing of the native pred loop (variance plane -> RED step -> regression update):
host enqueue time vs GPU completion time per plane, for the three cascade stage shapes of a 768x384 image."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd import rpc_synth
from satmvs_amd.modules.module import slice_RED_Regularization
from satmvs_amd.networks.casred import compute_depth_when_pred

dev = torch.device("cuda:0")
V = 3
B = int(os.environ.get("SMVS_BENCH_BATCH", "1"))
torch.manual_seed(0)
for name, (C, H, W, D, s) in {"stage1": (32, 96, 192, 48, 4), "stage2": (16, 192, 384, 32, 2), "stage3": (8, 384, 768, 8, 1)}.items():
    reg = slice_RED_Regularization(C, 8).to(dev).eval()
    feats = [torch.randn(B, C, H, W, device=dev) for _ in range(V)]
    import numpy as np
    rpc = rpc_synth.rescale_rpc(np.stack([rpc_synth.make_view_rpcs(V, 384, 768, seed=b) for b in range(B)]), s)
    proj = torch.from_numpy(rpc).to(dev)
    dv = torch.linspace(0, 400, D, device=dev).view(1, D, 1, 1).expand(B, D, H, W).contiguous()
    with torch.no_grad():
        for _ in range(2):
            compute_depth_when_pred(feats, proj, dv, D, reg, "rpc", False)
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter(); host = 0.0
        for _ in range(n):
            h0 = time.perf_counter()
            compute_depth_when_pred(feats, proj, dv, D, reg, "rpc", False)
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
    print("%s B=%d C%d %dx%d D=%d: %.1f us/plane GPU-complete, %.1f us/plane host enqueue" % (
        name, B, C, W, H, D, tot / n / D * 1e6, host / n / D * 1e6))
