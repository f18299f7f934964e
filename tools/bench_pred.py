#!/usr/bin/env python
"""End-to-end timing of the RED inference cascade (Infer_CascadeREDNet.forward) on one MI355X:
3-view 768x384, ndepths 48/32/8, random weights, synthetic RPCs.  Run twice:
   python tools/bench_pred.py                 # native plane loop (variance + RED + regression in HIP)
   SMVS_RED_TORCH=1 python tools/bench_pred.py   # same cost-volume kernel, stock PyTorch RED composite
"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd import rpc_synth
from satmvs_amd.networks.casred import Infer_CascadeREDNet

dev = torch.device("cuda:0")
H, W, V = 384, 768, 3
B = int(os.environ.get("SMVS_BENCH_BATCH", "1"))      # tiles per forward (the kernels take the batch in grid.z)
torch.manual_seed(0)
net = Infer_CascadeREDNet("rpc", ndepths=[48, 32, 8]).to(dev).eval()
imgs = torch.randn(B, V, 3, H, W, device=dev)
rpc = np.stack([rpc_synth.make_view_rpcs(V, H, W, seed=b) for b in range(B)])
proj = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev),
        "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev), "stage3": torch.from_numpy(rpc).to(dev)}
dv = torch.tensor([[0.0, 400.0]] * B, device=dev)
with torch.no_grad():
    for _ in range(2):
        out = net(imgs, proj, dv)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        out = net(imgs, proj, dv)
    torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
planes_vox = sum(d * (H // s) * (W // s) for d, s in zip([48, 32, 8], [4, 2, 1]))
print("Infer_CascadeREDNet %d tile(s) x 3-view %dx%d, 48/32/8 planes: %.1f ms per forward (%s RED), %.1f Mvox/s through the whole cascade, depth mean %.3f" % (
    B, W, H, ms, "stock-PyTorch" if os.environ.get("SMVS_RED_TORCH") == "1" else "native", B * planes_vox / ms / 1e3, float(out["depth"].mean())))
