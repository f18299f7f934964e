#!/usr/bin/env python
"""Time CostRegNet.forward (eval): native HIP vs the stock PyTorch composite (MIOpen)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd.modules.module import CostRegNet

dev = torch.device("cuda:0")
torch.manual_seed(0)
flops_per = lambda cin, cout, vox: 2.0 * 27 * cin * cout * vox
for name, (C, D, H, W) in {"stage1 C32 48x96x192": (32, 48, 96, 192), "stage2 C16 32x192x384": (16, 32, 192, 384),
                           "stage3 C8 8x384x768": (8, 8, 384, 768)}.items():
    net = CostRegNet(C, 8).to(dev).eval()
    x = torch.randn(1, C, D, H, W, device=dev)
    v = D * H * W
    fl = (flops_per(C, 8, v) + flops_per(8, 16, v / 8) + flops_per(16, 16, v / 8) + flops_per(16, 32, v / 64) +
          flops_per(32, 32, v / 64) + flops_per(32, 64, v / 512) + flops_per(64, 64, v / 512) +
          flops_per(64, 32, v / 512) + flops_per(32, 16, v / 64) + flops_per(16, 8, v / 8) + flops_per(8, 1, v))
    res = {}
    for mode in ("native", "torch"):
        os.environ["SMVS_COSTREG_TORCH"] = "1" if mode == "torch" else "0"
        with torch.no_grad():
            for _ in range(2):
                y = net(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                y = net(x)
            torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / n * 1e3
    print("%-24s native %.2f ms (%.1f TFLOP/s)   torch %.2f ms   x%.2f" % (name, res["native"], fl / res["native"] / 1e9, res["torch"], res["torch"] / res["native"]))
