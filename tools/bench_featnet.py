#!/usr/bin/env python
"""FeatureNet on 3 views of a 768x384 image: native single call (smvs_featnet_fwd) vs the PyTorch/MIOpen composite."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd.modules.module import FeatureNet

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = FeatureNet(base_channels=8, stride=4, num_stage=3, arch_mode="unet").to(dev).eval()
imgs = torch.randn(1, 3, 3, 384, 768, device=dev)
res = {}
for mode in ("native", "torch"):
    if mode == "torch":
        os.environ["SMVS_FEATNET_TORCH"] = "1"
    with torch.no_grad():
        for _ in range(3):
            net.forward_views(imgs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            net.forward_views(imgs)
        torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / n * 1e3
print("FeatureNet 3 views 768x384: native %.3f ms   torch %.3f ms   x%.2f" % (res["native"], res["torch"], res["torch"] / res["native"]))
