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
feats, pm, dv, truth = T.photo_consistent_inputs(dev); imgs = torch.zeros((1, 3, 3, 384, 768), device=dev)
for shrink, gamma, ub, ped, minc in ((0.1, 4.0, -3.0, 1.0, 0.5), (0.25, 4.0, -3.0, 1.0, 0.5), (0.1, 4.0, -3.0, 1.0, 0.7)):
    torch.manual_seed(43)
    net = FS.build_net(tag, "rpc").to(dev).eval()
    FS.randomise_batchnorm(net, 44); net.feature.forward_views = lambda im: feats
    T.trained_like(net, shrink, gamma, ub, ped)
    gains, confs = T.make_peaky(net, imgs, pm, dv, min_conf=minc)
    err, a, b = FS.native_vs_composite(net, imgs, pm, dv)
    det = {}
    f64 = FS.red_stages_against_float64(net, imgs, pm, dv, "rpc", det)
    for k, d in det.items():
        e = (d["native"].double() - d["float64"]).abs()[0]
        ec = (d["native"] - d["composite"]).abs()[0]
        q = lambda t, p: float(torch.quantile(t.flatten().float()[::7], p))
        iy, ix = divmod(int(e.argmax()), e.shape[1])
        print("   %s |native-f64|: p50 %.2e p99 %.2e p99.9 %.2e max %.2e at (%d,%d) of %s conf64 there %.3f native %.3f composite %.3f f64 %.3f; frac>1e-3: %.2e | |native-composite| p99.9 %.2e frac>1e-3 %.2e" % (
            k, q(e, 0.5), q(e, 0.99), q(e, 0.999), float(e.max()), iy, ix, tuple(e.shape), float(d["p64"][0, iy, ix]), float(d["native"][0, iy, ix]),
            float(d["composite"][0, iy, ix]), float(d["float64"][0, iy, ix]), float((e > 1e-3).float().mean()), q(ec, 0.999), float((ec > 1e-3).float().mean())), flush=True)
    herr = np.abs(a["stage3"]["depth"][0].cpu().numpy() - truth)[32:-32, 32:-32]
    print("shrink %.2f gamma %.2f ubias %.1f pedestal %.1f minconf %.2f | gains %s conf %s | free-running %s | f64 %s | surface median %.2f m p90 %.2f m" % (
        shrink, gamma, ub, ped, minc, {k: int(v) for k, v in gains.items()}, {k: "%.2f" % v for k, v in confs.items()}, {k: "%.2g" % v for k, v in err.items()},
        {k: tuple("%.2g" % x for x in v) for k, v in f64.items()}, float(np.median(herr)), float(np.percentile(herr, 90))), flush=True)
