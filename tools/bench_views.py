#!/usr/bin/env python
"""Cost-volume build at 2..8 views (3-view 768x384 tile shape, 8 planes, C=32): which kernel serves many-view tiles."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from satmvs_amd import _lib
dev = torch.device("cuda:0")
st = _lib.current_stream(dev)
for V in (3, 5, 6, 7, 8):
    C, D, H, W = 32, 8, 384, 768
    feats, rpc, depth = bench.make_inputs(V, C, D, D, 0, H, W, dev)
    depth = torch.linspace(0.0, 400.0 * 7 / 63, D).view(1, D, 1, 1).expand(1, D, H, W).contiguous().to(dev)   # 8 of 64 planes over 0..400 m
    out = torch.empty((1, C, D, H, W), device=dev)
    srcs = _lib.ptr_array(feats[1:])
    def step():
        _lib.call("smvs_rpc_costvol_fwd", _lib.ptr(feats[0]), srcs, V - 1, _lib.ptr(rpc), _lib.ptr(depth), 1, _lib.ptr(out), 1, C, D, H, W, 0, D, D, 0, st)
    for _ in range(5): step()
    _, ms = bench.time_steps(step, 30)
    print("V=%d: %.3f ms per 8 planes of 768x384, C=32  (%s)" % (V, ms, bench.kernel_name(V, C, D)))
