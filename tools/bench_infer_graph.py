#!/usr/bin/env python
"""Infer_CascadeREDNet forward (3-view 768x384, planes 48/32/8): eager three-stream plane pipeline vs eager single stream vs the
whole forward captured in one HIP graph (single-stream mode, smvs_red_set_streams(0)) and replayed.
    python tools/bench_infer_graph.py [batch]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd import _lib, rpc_synth
from satmvs_amd.networks.casred import Infer_CascadeREDNet

dev = torch.device("cuda:0")
H, W, nd = 384, 768, [48, 32, 8]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
net = Infer_CascadeREDNet("rpc", ndepths=nd).to(dev).eval()
imgs = torch.randn(B, 3, 3, H, W, device=dev)
rpc = np.stack([rpc_synth.make_view_rpcs(3, H, W, seed=b) for b in range(B)])
pm = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev), "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev),
      "stage3": torch.from_numpy(rpc).to(dev)}
dv = torch.tensor([[0.0, 400.0]] * B, device=dev)
lib = _lib.load()


def timed(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


with torch.no_grad():
    ref = net(imgs, pm, dv)["stage3"]["depth"].clone()
    print("B=%d eager, three-stream pipeline : median %.2f ms (min %.2f, max %.2f)" % ((B,) + timed(lambda: net(imgs, pm, dv))))
    lib.smvs_red_set_streams(0)
    try:
        one = net(imgs, pm, dv)["stage3"]["depth"]
        assert torch.equal(one, ref)
        print("B=%d eager, single stream         : median %.2f ms (min %.2f, max %.2f)" % ((B,) + timed(lambda: net(imgs, pm, dv))))
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net(imgs, pm, dv)
            with torch.cuda.graph(graph, stream=side):
                out = net(imgs, pm, dv)
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out["stage3"]["depth"], ref)
        print("B=%d HIP graph replay             : median %.2f ms (min %.2f, max %.2f)" % ((B,) + timed(graph.replay)))
    finally:
        lib.smvs_red_set_streams(2)
