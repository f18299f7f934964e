#!/usr/bin/env python
"""Random shapes / view counts / geometries / height spans through smvs_costvol_bwd against autograd through the per-view warp
operators (the comparison of tests/test_hip_parity.py::test_costvol_backward_matrix):  python tests/fuzz/fuzz_costvol_bwd.py [n] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from satmvs_amd.modules import warping
import test_hip_parity as T

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for it in range(n):
    V = int(rng.integers(2, 9)); C = int(rng.integers(1, 13)); D = int(rng.integers(1, 21))
    H = int(rng.integers(6, 120)); W = int(rng.integers(8, 220)); B = int(rng.integers(1, 3))
    geo = "rpc" if rng.random() < 0.7 else "pinhole"
    jitter = bool(rng.random() < 0.6)
    feats, gp, depth = T._inputs(B, V, C, D, H, W, seed=int(rng.integers(0, 10000)), jitter=jitter, geo=geo)
    if rng.random() < 0.3:                                      # a span that overflows the boxes
        lo, hi = (0.0, float(rng.uniform(2000, 40000))) if geo == "rpc" else (300.0, float(rng.uniform(900, 3000)))
        depth = np.broadcast_to(np.linspace(lo, hi, D, dtype=np.float32).reshape(1, D, 1, 1), (B, D, H, W)).copy()
    gpt, dt = torch.from_numpy(gp).to(dev), torch.from_numpy(depth).to(dev)
    fs = [torch.from_numpy(f).to(dev).requires_grad_(True) for f in feats]
    var = warping.variance_cost_volume(fs, gpt, dt, geo)
    gout = torch.randn_like(var)
    var.backward(gout)
    fs2 = [torch.from_numpy(f).to(dev).requires_grad_(True) for f in feats]
    s = fs2[0].unsqueeze(2).repeat(1, 1, D, 1, 1); q = s ** 2
    for v in range(1, V):
        w = warping.rpc_warping(fs2[v], gpt[:, v], gpt[:, 0], dt, None) if geo == "rpc" else warping.homo_warping(fs2[v], gpt[:, v], gpt[:, 0], dt)
        s = s + w; q = q + w ** 2
    (q / V - (s / V) ** 2).backward(gout)
    for v, (a, b) in enumerate(zip(fs, fs2)):
        scale = float(b.grad.abs().max())
        err = float((a.grad - b.grad).abs().max()) / max(scale, 1e-20)
        worst = max(worst, err)
        if not (err <= 1e-4) or not torch.isfinite(a.grad).all():
            print("MISMATCH it=%d B=%d V=%d C=%d D=%d H=%d W=%d geo=%s jitter=%s view=%d err=%.3g scale=%.3g" % (it, B, V, C, D, H, W, geo, jitter, v, err, scale))
print("%d cases, worst relative error %.3g" % (n, worst))
