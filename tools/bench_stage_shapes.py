#!/usr/bin/env python
"""Cost-volume build at the cascade's stage-2 / stage-3 launch shapes (one chunk of 8 planes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from satmvs_amd import _lib
dev = torch.device("cuda:0")
st = _lib.current_stream(dev)
for name, (C, D, H, W, lo, hi) in {"stage2 C16 384x192 x8": (16, 8, 192, 384, 180.0, 215.0), "stage3 C8 768x384 x8": (8, 8, 384, 768, 190.0, 207.5),
                                   "stage2 C16 384x192 x32": (16, 32, 192, 384, 120.0, 280.0)}.items():
    V = 3
    feats, rpc, _ = bench.make_inputs(V, C, D, D, 0, H, W, dev)
    depth = torch.linspace(lo, hi, D).view(1, D, 1, 1).expand(1, D, H, W).contiguous().to(dev)
    out = torch.empty((1, C, D, H, W), device=dev)
    srcs = _lib.ptr_array(feats[1:])
    def step():
        _lib.call("smvs_rpc_costvol_fwd", _lib.ptr(feats[0]), srcs, V - 1, _lib.ptr(rpc), _lib.ptr(depth), 1, _lib.ptr(out), 1, C, D, H, W, 0, D, D, 0, st)
    for _ in range(50): step()
    _, ms = bench.time_steps(step, 200)
    print("%s %-26s %.4f ms" % (os.environ.get("SMVS_LIB_PATH", "default").split("/")[-1], name, ms))
