#!/usr/bin/env python
"""One cascade stage of the native pred loop, a few repetitions (for rocprofv3 traces)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd import rpc_synth
from satmvs_amd.modules.module import slice_RED_Regularization
from satmvs_amd.networks.casred import compute_depth_when_pred
dev = torch.device("cuda:0")
C, H, W, D, s, V = 32, 96, 192, 48, 4, 3
torch.manual_seed(0)
reg = slice_RED_Regularization(C, 8).to(dev).eval()
feats = [torch.randn(1, C, H, W, device=dev) for _ in range(V)]
proj = torch.from_numpy(rpc_synth.rescale_rpc(rpc_synth.make_view_rpcs(V, 384, 768, seed=0)[None], s)).to(dev)
dv = torch.linspace(0, 400, D, device=dev).view(1, D, 1, 1).expand(1, D, H, W).contiguous()
with torch.no_grad():
    for _ in range(4):
        compute_depth_when_pred(feats, proj, dv, D, reg, "rpc", False)
    torch.cuda.synchronize()
