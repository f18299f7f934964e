#!/usr/bin/env python
"""smvs_conv3x3_wgrad against torch's (MIOpen's) weight gradient for every ConvGRU convolution shape of the 768x384 cascade:
    python tools/bench_wgrad.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd import _lib
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = False


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot_n = tot_t = 0.0
for stage, C, H, W, planes in (("stage1", 32, 96, 192, 48), ("stage2", 16, 192, 384, 32), ("stage3", 8, 384, 768, 8)):
    xin = [C, 16, 32, 64]
    for g in range(4):
        hc = 8 << g
        h, w = H >> g, W >> g
        for name, cout in (("gate", 2 * hc), ("cand", hc)):
            cin = xin[g] + hc
            x = torch.randn(1, cin, h, w, device=dev)
            dy = torch.randn(1, cout, h, w, device=dev)
            wt = torch.randn(cout, cin, 3, 3, device=dev)
            dw = torch.zeros_like(wt); db = torch.zeros(cout, device=dev)
            st = _lib.current_stream(dev)

            def native():
                _lib.call("smvs_conv3x3_wgrad", _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db), 1, cin, cout, h, w, st)

            def ref():
                torch.ops.aten.convolution_backward(dy, x, wt, [cout], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True])
            tn, tr = timeit(native), timeit(ref)
            gf = 2.0 * 9 * h * w * cin * cout / 1e9
            print("%s level %d %s  Cin %3d Cout %3d %3dx%3d  %6.3f GFLOP  native %7.1f us (%5.1f TFLOP/s)   torch %7.1f us   x%.2f   (%d calls per step)" % (
                stage, g + 1, name, cin, cout, h, w, gf, tn, gf / tn * 1e3, tr, tr / tn, planes))
            tot_n += tn * planes; tot_t += tr * planes
print("per training step (all planes): native %.2f ms, torch %.2f ms" % (tot_n / 1e3, tot_t / 1e3))
