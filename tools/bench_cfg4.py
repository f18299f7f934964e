#!/usr/bin/env python
"""cfg4 (5-view 1536x768, C=32) cost-volume launches of the loaded library: one GPU's 8-plane shard and the whole 64-plane
sweep, alternating, a few rounds.  SMVS_LIB_PATH=gpurun_ab/<cand>.so python tools/bench_cfg4.py [views]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from satmvs_amd import _lib
_lib.load()
dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 5
C, H, W = 32, 768, 1536
stream = _lib.current_stream(dev)
res = {}
for name, D, lo, hi in (("shard8", 8, 0.0, 400.0 * 7 / 63), ("sweep64", 64, 0.0, 400.0)):
    feats, rpc, _ = bench.make_inputs(V, C, D, D, 0, H, W, dev)
    depth = torch.linspace(lo, hi, D, dtype=torch.float32).view(1, D, 1, 1).expand(1, D, H, W).contiguous().to(dev)
    out = torch.empty((1, C, D, H, W), dtype=torch.float32, device=dev)
    srcs = _lib.ptr_array(feats[1:])
    def step():
        _lib.call("smvs_rpc_costvol_fwd", _lib.ptr(feats[0]), srcs, V - 1, _lib.ptr(rpc), _lib.ptr(depth), 1,
                  _lib.ptr(out), 1, C, D, H, W, 0, D, D, 0, stream)
    for _ in range(10):
        step()
    ms = []
    for _ in range(3):
        _, m = bench.time_steps(step, 30 if D == 8 else 10)
        ms.append(m)
    bpv = bench.algorithmic_bytes_per_voxel(V, C, D)
    m = min(ms)
    print("%s V=%d: %.4f ms (%s)  roofline_frac %.4f  checksum %.6e" % (name, V, m, " ".join("%.4f" % x for x in ms),
          bpv * D * H * W / (m * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, float(out.double().sum())))
    del feats, out, depth
    torch.cuda.empty_cache()
