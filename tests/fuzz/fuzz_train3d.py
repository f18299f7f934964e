#!/usr/bin/env python
"""Randomised parity of the 3-D training operators (round 5): smvs_conv3d_fwd + its adjoint + smvs_conv3d_wgrad behind
train_fns._conv3d, and smvs_batchnorm_train_fwd / _bwd behind train_fns._bn3d_relu, against float64 evaluations of the same torch
layers on the CPU.  Random layer kind (Conv3d stride 1 / 2, ConvTranspose3d stride 2), batch 1-2, 1-40 channels each side (incl. odd
counts and the 32 / 64-channel MFMA layers), D / H / W up to 10 / 40 / 150 (ragged: not multiples of the 64-column strips or the 62-column
tiles), BatchNorm over 3-D and 2-D blocks with and without ReLU.
    python tests/fuzz/fuzz_train3d.py [cases] [seed]        last line: "<n> cases, worst relative error <e>"    (MISMATCH lines on failure)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from satmvs_amd.modules import train_fns as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
worst = 0.0


def rel(got, ref):
    return float((got.double().cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)


for case in range(n_cases):
    torch.manual_seed(seed * 100003 + case)
    if case % 4 == 3:                                                  # BatchNorm + ReLU
        dims = [int(rng.integers(1, 7)), int(rng.integers(1, 30)), int(rng.integers(1, 90))][int(rng.integers(0, 2)):]
        B, C, relu = int(rng.integers(1, 4)), int(rng.integers(1, 70)), bool(rng.integers(0, 2))
        if B * int(np.prod(dims)) < 2:
            dims[-1] += 2
        bn = (torch.nn.BatchNorm3d if len(dims) == 3 else torch.nn.BatchNorm2d)(C, momentum=0.1).to(dev).train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5); bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2.0)
        bn64 = (torch.nn.BatchNorm3d if len(dims) == 3 else torch.nn.BatchNorm2d)(C, momentum=0.1).double().train()
        bn64.load_state_dict({k: (v.double() if v.dtype.is_floating_point else v).cpu() for k, v in bn.state_dict().items()})
        x = (torch.randn(B, C, *dims, device=dev) * float(rng.uniform(0.2, 3.0)) + float(rng.uniform(-2, 2))).requires_grad_(True)
        y = T._bn3d_relu(bn, x, relu)
        if y is None or "BatchNormRelu" not in type(y.grad_fn).__name__:
            print("MISMATCH case %d: BatchNorm did not take the native path" % case)
            continue
        gy = torch.randn_like(y)
        y.backward(gy)
        x64 = x.detach().double().cpu().requires_grad_(True)
        y64 = bn64(x64)
        y64 = torch.relu(y64) if relu else y64
        y64.backward(gy.double().cpu())
        errs = {"y": rel(y.detach(), y64.detach()), "dx": rel(x.grad, x64.grad), "dgamma": rel(bn.weight.grad, bn64.weight.grad),
                "dbeta": rel(bn.bias.grad, bn64.bias.grad), "running_mean": rel(bn.running_mean, bn64.running_mean),
                "running_var": rel(bn.running_var, bn64.running_var)}
        tol = {"y": 2e-5, "dx": 1e-4, "dgamma": 1e-4, "dbeta": 1e-4, "running_mean": 1e-5, "running_var": 1e-5}
        desc = "bn%dd B%d C%d %s relu=%d" % (len(dims), B, C, dims, relu)
    else:
        kind = ("conv_s1", "conv_s2", "convT_s2")[int(rng.integers(0, 3))]
        B = int(rng.integers(1, 3))
        cin, cout = (int(rng.choice([1, 2, 3, 5, 8, 16, 17, 32, 40, 64])) for _ in range(2))
        D, H, W = int(rng.integers(1, 6)), int(rng.integers(1, 21)), int(rng.integers(1, 76))
        if kind == "conv_s2":
            D, H, W = 2 * D, 2 * H, 2 * W
        if kind == "conv_s1":
            conv = torch.nn.Conv3d(cin, cout, 3, stride=1, padding=1, bias=False)
        elif kind == "conv_s2":
            conv = torch.nn.Conv3d(cin, cout, 3, stride=2, padding=1, bias=False)
        else:
            conv = torch.nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False)
        conv64 = type(conv)(cin, cout, 3, **({"stride": 2, "padding": 1, "output_padding": 1, "bias": False} if kind == "convT_s2" else
                                             {"stride": 1 if kind == "conv_s1" else 2, "padding": 1, "bias": False})).double()
        conv64.weight.data.copy_(conv.weight.detach().double())
        conv = conv.to(dev)
        x = torch.randn(B, cin, D, H, W, device=dev, requires_grad=True)
        y = T._conv3d(conv, x)
        if "Conv3dNative" not in type(y.grad_fn).__name__:
            print("MISMATCH case %d: %s did not take the native path (%s)" % (case, kind, type(y.grad_fn).__name__))
            continue
        gy = torch.randn_like(y)
        y.backward(gy)
        x64 = x.detach().double().cpu().requires_grad_(True)
        y64 = conv64(x64)
        y64.backward(gy.double().cpu())
        errs = {"y": rel(y.detach(), y64.detach()), "dx": rel(x.grad, x64.grad), "dw": rel(conv.weight.grad, conv64.weight.grad)}
        tol = {"y": 2e-5, "dx": 2e-5, "dw": 5e-5}
        desc = "%s B%d %d->%d %s" % (kind, B, cin, cout, (D, H, W))
    for k, e in errs.items():
        if not np.isfinite(e) or e > tol[k]:
            print("MISMATCH case %d (%s): %s %.3g > %.0e" % (case, desc, k, e, tol[k]))
        worst = max(worst, e / tol[k] * 1e-5)
print("%d cases, worst relative error %.3g" % (n_cases, worst))
