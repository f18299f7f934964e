#!/usr/bin/env python
"""Parameter probe for tests/test_full_size_red_conditioned.py (trained-like RED weights at 768x384): per setting the gains, the mean
confidences and (native-f64, composite-f64, native-composite) per stage.   python tools/conditioned_probe.py [tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SMVS_ARITH", "exact")
import numpy as np
import torch
import test_full_size_red_conditioned as T
import test_full_size_cascade as FS
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = False
tag = sys.argv[1] if len(sys.argv) > 1 else "redinf"
imgs, pm, dv, truth = T.photo_consistent_inputs(dev)
for shrink, gamma, ub, minc in ((0.25, 1.0, -3.0, 0.5), (0.25, 0.5, -3.0, 0.5), (0.1, 1.0, -3.0, 0.5), (0.5, 1.0, -3.0, 0.5), (0.25, 1.0, -1.0, 0.5), (0.25, 1.0, -3.0, 0.7)):
    torch.manual_seed(43)
    net = FS.build_net(tag, "rpc").to(dev).eval()
    FS.randomise_batchnorm(net, 44)
    T.trained_like(net, shrink, gamma, ub)
    gains, confs = T.make_peaky(net, imgs, pm, dv, min_conf=minc)
    err, a, b = FS.native_vs_composite(net, imgs, pm, dv)
    f64 = FS.red_stages_against_float64(net, imgs, pm, dv, "rpc")
    herr = np.abs(a["stage3"]["depth"][0].cpu().numpy() - truth)[32:-32, 32:-32]
    print("shrink %.2f gamma %.2f ubias %.1f minconf %.1f | gains %s conf %s | free-running %s | f64 %s | surface median %.2f m p90 %.2f m" % (
        shrink, gamma, ub, minc, {k: int(v) for k, v in gains.items()}, {k: "%.2f" % v for k, v in confs.items()}, {k: "%.2g" % v for k, v in err.items()},
        {k: tuple("%.2g" % x for x in v) for k, v in f64.items()}, float(np.median(herr)), float(np.percentile(herr, 90))), flush=True)
