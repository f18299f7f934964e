#!/usr/bin/env python
"""Random shapes through the smaller operators of the path against the CPU oracle: rpc_warping / homo_warping (bits), softmax and
window regressions, streaming regression, in-kernel height hypotheses (bits).   python tests/fuzz/fuzz_ops.py [n] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc
from satmvs_amd.modules import module as M
from satmvs_amd.modules import warping
from satmvs_amd.modules.depth_range import GeneratedHeights
import test_hip_parity as T

orc.build()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0
for it in range(n):
    C = int(rng.integers(1, 20)); D = int(rng.integers(1, 18)); H = int(rng.integers(3, 90)); W = int(rng.integers(3, 150)); B = int(rng.integers(1, 3))
    jitter = bool(rng.random() < 0.6)
    # ---- warps (bit comparison)
    for geo in ("rpc", "pinhole"):
        feats, gp, depth = T._inputs(B, 2, C, D, H, W, seed=int(rng.integers(0, 10000)), jitter=jitter, geo=geo)
        if geo == "rpc":
            got = warping.rpc_warping(t(feats[1]), t(gp[:, 1]), t(gp[:, 0]), t(depth), None).cpu().numpy()
            want = orc.rpc_warping(feats[1], gp[:, 1], gp[:, 0], depth)
        else:
            got = warping.homo_warping(t(feats[1]), t(gp[:, 1]), t(gp[:, 0]), t(depth)).cpu().numpy()
            want = orc.homo_warping(feats[1], gp[:, 1], gp[:, 0], depth)
        nb = int((got != want).sum())
        if nb > max(1, 1e-4 * got.size):
            bad += 1; print("MISMATCH warp %s it=%d B=%d C=%d D=%d H=%d W=%d jitter=%s: %d of %d" % (geo, it, B, C, D, H, W, jitter, nb, got.size))
    # ---- regressions
    reg = (rng.standard_normal((B, D, H, W)) * 3).astype(np.float32)
    dvals = depth if depth.ndim == 4 else depth
    with torch.no_grad():
        d1, c1 = M.softmax_depth_regression(t(reg), t(dvals))
        d2, c2, v2 = M.window_depth_regression(t(reg), t(dvals), lamb=1.5)
    od, oc = orc.softmax_regress(reg, dvals)
    wd, wc, wv = orc.window_regress(reg, dvals, lamb=1.5)
    e = [np.abs(d1.cpu().numpy() - od).max(), np.abs(c1.cpu().numpy() - oc).max(), np.abs(d2.cpu().numpy() - wd).max(), np.abs(v2.cpu().numpy() - wv).max()]
    if e[0] > 1e-3 or e[1] > 1e-5 or e[2] > 1e-3 or e[3] > 2e-3 or (np.abs(c2.cpu().numpy() - wc) > 1e-5).mean() > 0.01:
        bad += 1; print("MISMATCH regress it=%d B=%d D=%d H=%d W=%d: %s" % (it, B, D, H, W, e))
    acc = M.StreamingRegression(B, H, W, dev)
    oacc = orc.StreamRegress(B, H, W)
    for d in range(D):
        acc.step(t(reg[:, d]), t(dvals), d)
        oacc.step(reg[:, d], dvals, d)
    sd, sc = acc.result()
    osd, osc = oacc.final()
    if np.abs(sd.cpu().numpy() - osd).max() > 1e-4 or np.abs(sc.cpu().numpy() - osc).max() > 1e-6:
        bad += 1; print("MISMATCH streaming it=%d" % it, np.abs(sd.cpu().numpy() - osd).max())
    # ---- in-kernel hypotheses (bit comparison)
    sh, sw = int(rng.integers(2, 40)), int(rng.integers(2, 60))
    scale = int(rng.choice([1, 2]))
    prev = (200.0 + 50.0 * rng.standard_normal((B, sh, sw))).astype(np.float32)
    nd = int(rng.integers(2, 12)); interval = float(rng.choice([1.25, 2.5, 5.0, 10.0]))
    img_hw, stage_hw = (sh * 2 * scale, sw * 2 * scale), (sh * 2, sw * 2)
    gen = GeneratedHeights(t(prev), nd, interval, img_hw, stage_hw)
    got = gen.materialize().cpu().numpy()
    want = orc.height_hypotheses(prev, nd, interval, img_hw, stage_hw)
    if not np.array_equal(got, want):
        bad += 1; print("MISMATCH hypotheses it=%d prev %s nd=%d interval=%g img=%s stage=%s: %d differ" % (it, prev.shape, nd, interval, img_hw, stage_hw, int((got != want).sum())))
print("%d rounds, %d mismatching checks" % (n, bad))
