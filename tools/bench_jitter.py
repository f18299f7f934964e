#!/usr/bin/env python
"""The headline launch with per-pixel jittered heights (plane + N(0, sigma) per pixel: what cascade stages 2-3 hand over), sigma from argv:
    python tools/bench_jitter.py [sigma_m ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from satmvs_amd import _lib

dev = torch.device("cuda:0")
V, C, D, H, W = bench.WORKLOADS["cfg2_rpc_3view_768x384x64_c32"]
stream = _lib.current_stream(dev)
for sigma in [float(a) for a in sys.argv[1:]] or [0.0, 0.5, 2.0, 8.0]:
    feats, rpc, depth = bench.make_inputs(V, C, D, D, 0, H, W, dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    depth = (depth.cpu() + sigma * torch.randn((1, D, H, W), generator=g)).contiguous().to(dev)
    out = torch.empty((1, C, D, H, W), dtype=torch.float32, device=dev)
    srcs = _lib.ptr_array(feats[1:])

    def step():
        _lib.call("smvs_rpc_costvol_fwd", _lib.ptr(feats[0]), srcs, V - 1, _lib.ptr(rpc), _lib.ptr(depth), 1, _lib.ptr(out), 1, C, D, H, W, 0, D, D, 0, stream)
    for _ in range(200):
        step()
    _, ms = bench.time_steps(step, 100)
    print("sigma %.1f m: %.4f ms per launch, frac %.4f" % (sigma, ms, bench.algorithmic_bytes_per_voxel(V, C, D) * D * H * W / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS), flush=True)
    del feats, out, depth
