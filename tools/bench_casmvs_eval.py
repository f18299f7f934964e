#!/usr/bin/env python
"""CascadeMVSNet and UCSNet (3-D CostRegNet regulariser) in eval mode, 3-view 768x384, ndepths 48/32/8."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd import rpc_synth
from satmvs_amd.networks.casmvs import CascadeMVSNet
from satmvs_amd.networks.ucs import UCSNet

dev = torch.device("cuda:0")
H, W, V = 384, 768, 3
imgs = torch.randn(1, V, 3, H, W, device=dev)
rpc = rpc_synth.make_view_rpcs(V, H, W, seed=0)[None]
proj = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev),
        "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev), "stage3": torch.from_numpy(rpc).to(dev)}
dv = torch.tensor([[0.0, 400.0]], device=dev)
for name, make in (("CascadeMVSNet", lambda: CascadeMVSNet("rpc", min_interval=2.5, ndepths=[48, 32, 8])),
                   ("UCSNet", lambda: UCSNet("rpc", stage_configs=[48, 32, 8]))):
    torch.manual_seed(0)
    net = make().to(dev).eval()
    with torch.no_grad():
        for _ in range(2):
            out = net(imgs, proj, dv)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            out = net(imgs, proj, dv)
        torch.cuda.synchronize()
    print("%s eval 3-view %dx%d, 48/32/8 planes: %.1f ms per forward, depth mean %.3f" % (
        name, W, H, (time.perf_counter() - t0) / n * 1e3, float(out["stage3"]["depth"].mean())))
