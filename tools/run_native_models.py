#!/usr/bin/env python
"""Native regulariser / feature kernels only (no PyTorch composite, no MIOpen), a few forwards each: the command the MFMA
counter passes and the kernel trace of tools/profile_r03.sh run.  Prints the analytic convolution FLOPs of each model so
that counter-derived and analytic rates can be compared."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd import rpc_synth
from satmvs_amd.modules.module import CostRegNet
from satmvs_amd.networks.casred import Infer_CascadeREDNet

dev = torch.device("cuda:0")
torch.manual_seed(0)
n = int(os.environ.get("SMVS_RUNS", "5"))
fl3 = lambda cin, cout, vox: 2.0 * 27 * cin * cout * vox
for name, (C, D, H, W) in {"costreg stage1 C32 48x96x192": (32, 48, 96, 192), "costreg stage2 C16 32x192x384": (16, 32, 192, 384),
                           "costreg stage3 C8 8x384x768": (8, 8, 384, 768)}.items():
    net = CostRegNet(C, 8).to(dev).eval()
    x = torch.randn(1, C, D, H, W, device=dev)
    v = D * H * W
    mfma_fl = fl3(16, 32, v / 64) + fl3(32, 32, v / 64) + fl3(32, 64, v / 512) + fl3(64, 64, v / 512)   # conv3..conv6 (MFMA kernel)
    with torch.no_grad():
        for _ in range(2):
            net(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            net(x)
        torch.cuda.synchronize()
    print("%-30s %.3f ms per forward; conv3..conv6 on mfma_conv_kernel<27,..>: %.3f GFLOP per forward" % (name, (time.perf_counter() - t0) / n * 1e3, mfma_fl / 1e9))
H, W, V = 384, 768, 3
net = Infer_CascadeREDNet("rpc", ndepths=[48, 32, 8]).to(dev).eval()
imgs = torch.randn(1, V, 3, H, W, device=dev)
rpc = rpc_synth.make_view_rpcs(V, H, W, seed=0)[None]
proj = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev), "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev),
        "stage3": torch.from_numpy(rpc).to(dev)}
dv = torch.tensor([[0.0, 400.0]], device=dev)
with torch.no_grad():
    for _ in range(2):
        net(imgs, proj, dv)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        net(imgs, proj, dv)
    torch.cuda.synchronize()
# MFMA jobs of a RED plane (conv_jobs_kernel kind 2 + encoder conv2/conv3 on mfma_conv_kernel<9,..>), per plane of size h x w at C input channels
def red_mfma_flops(C, h, w):
    f = lambda cin, cout, px: 2.0 * 9 * cin * cout * px
    lv = [(h >> g) * (w >> g) for g in range(4)]
    xin = [C, 16, 32, 64]; hid = [8, 16, 32, 64]
    tot = f(16, 32, lv[2]) + f(32, 64, lv[3])                                            # encoder conv2, conv3
    for g in range(4):
        if 2 * hid[g] >= 32: tot += f(xin[g] + hid[g], 2 * hid[g], lv[g])                # gate convolutions with >= 32 outputs
        if hid[g] >= 32: tot += f(xin[g] + hid[g], hid[g], lv[g])                         # candidate convolutions with >= 32 outputs
    return tot
tot = sum(d * red_mfma_flops(c, H // s, W // s) for d, s, c in zip([48, 32, 8], [4, 2, 1], [32, 16, 8]))
print("Infer_CascadeREDNet 3-view %dx%d 48/32/8: %.2f ms per forward; MFMA-routed convolutions of the RED planes: %.3f GFLOP per forward" % (W, H, (time.perf_counter() - t0) / n * 1e3, tot / 1e9))
