#!/usr/bin/env python
"""Time one plane of the RED regulariser: native HIP step vs the stock PyTorch composite (MIOpen)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd.modules.module import slice_RED_Regularization


def _REDCore_step(reg, cost, s1, s2, s3, s4):
    """The PyTorch composite (what training uses), forced regardless of autograd state."""
    neg = -cost
    e1 = reg.conv1(neg); e2 = reg.conv2(e1); e3 = reg.conv3(e2)
    r4, s4 = reg.conv_gru4(e3, s4)
    u3 = reg.upconv3(r4)
    r3, s3 = reg.conv_gru3(e2, s3)
    u2 = reg.upconv2(u3 + r3)
    r2, s2 = reg.conv_gru2(e1, s2)
    u1 = reg.upconv1(u2 + r2)
    r1, s1 = reg.conv_gru1(neg, s1)
    return reg.upconv2d(u1 + r1), s1, s2, s3, s4

dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, (C, H, W) in {"stage1 96x192 C32": (32, 96, 192), "stage2 192x384 C16": (16, 192, 384),
                        "stage3 384x768 C8": (8, 384, 768)}.items():
    reg = slice_RED_Regularization(C, 8).to(dev).eval()
    x = torch.randn(1, C, H, W, device=dev)
    res = {}
    for mode in ("native", "torch"):
        st = reg.initial_states(1, H, W, dev)
        with torch.no_grad():
            fn = reg.native_step if mode == "native" else (lambda c, *s: _REDCore_step(reg, c, *s))
            for _ in range(3):
                out = fn(x, *st); st = list(out[1:])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                out = fn(x, *st); st = list(out[1:])
            torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / n * 1e3
    print("%-22s native %.3f ms/plane   torch %.3f ms/plane   x%.2f" % (name, res["native"], res["torch"], res["torch"] / res["native"]))
