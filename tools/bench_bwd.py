#!/usr/bin/env python
"""Forward + backward of the fused variance cost volume at the metric shape (training path)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd import rpc_synth
from satmvs_amd.modules.warping import variance_cost_volume

dev = torch.device("cuda:0")
# heights: the headline span for cfg2; the cascade's own spacings for the stage shapes (stage 1: the whole range on 48 planes,
# stage 2: 32 planes 5 m apart, stage 3: 8 planes 2.5 m apart) and, last, a spacing whose boxes overflow (register scheme)
VIEWS = {"5-view C32 768x384x32": 5, "4-view C32 768x384x32": 4, "2-view C32 768x384x32": 2, "7-view C16 384x192x16": 7}
for name, (C, H, W, D, s, lo, hi) in {"cfg2 C32 768x384x64": (32, 384, 768, 64, 1, 0.0, 400.0),
                                      "stage1 C32 192x96x48": (32, 96, 192, 48, 4, 0.0, 400.0),
                                      "stage2 C16 384x192x32": (16, 192, 384, 32, 2, 150.0, 305.0),
                                      "stage3 C8 768x384x8": (8, 384, 768, 8, 1, 190.0, 207.5),
                                      "overflow C8 768x384x8, 57 m planes": (8, 384, 768, 8, 1, 0.0, 400.0),
                                      "5-view C32 768x384x32": (32, 384, 768, 32, 1, 0.0, 200.0),
                                      "4-view C32 768x384x32": (32, 384, 768, 32, 1, 0.0, 200.0),
                                      "2-view C32 768x384x32": (32, 384, 768, 32, 1, 0.0, 200.0),
                                      "7-view C16 384x192x16": (16, 192, 384, 16, 2, 100.0, 200.0)}.items():
    V = VIEWS.get(name, 3)
    torch.manual_seed(0)
    feats = [torch.randn(1, C, H, W, device=dev, requires_grad=True) for _ in range(V)]
    proj = torch.from_numpy(rpc_synth.rescale_rpc(rpc_synth.make_view_rpcs(V, 384, 768, seed=0)[None], s)).to(dev)
    dv = torch.linspace(lo, hi, D, device=dev).view(1, D, 1, 1).expand(1, D, H, W).contiguous()
    vol = variance_cost_volume(feats, proj, dv, "rpc", False)
    g = torch.randn_like(vol)
    for _ in range(2):
        vol = variance_cost_volume(feats, proj, dv, "rpc", False)
        vol.backward(g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        vol = variance_cost_volume(feats, proj, dv, "rpc", False)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for _ in range(n):
        vol = variance_cost_volume(feats, proj, dv, "rpc", False)
        vol.backward(g)
    torch.cuda.synchronize(); tfb = (time.perf_counter() - t0) / n * 1e3
    print("%-36s forward %.3f ms   forward+backward %.3f ms   (backward %.3f ms)" % (name, tf, tfb, tfb - tf))
