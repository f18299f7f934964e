#!/usr/bin/env python
"""Random shapes / view counts / channel counts / geometries / height spans through the fused variance-volume kernels against the
CPU oracle:  python tests/fuzz/fuzz_costvol_fwd.py [n] [seed] [exact|fused|pc]
exact (default): bit comparison, as tests/test_hip_parity.py::test_costvol_vs_oracle; fused: the library's default arithmetic at its
contract, |delta| <= 1e-5 max(1, |v|) with the same NaN pattern, as tests/test_fused_arith.py; pc: rpc cases through smvs_rpc_plane_coef +
smvs_rpc_costvol_fwd_pc (exact arithmetic) on (B,D) planes, broadcast planes, planes with a few jittered / NaN voxels and fully jittered
heights -- compared like `exact` (the folded cubics move a coordinate by ~1e-13 px, so a voxel whose tap coordinate straddles a float32
rounding boundary may differ: counted, and bounded per case like the other modes)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc
from satmvs_amd.modules import warping
import test_hip_parity as T

from satmvs_amd import _lib
orc.build()
mode = sys.argv[3] if len(sys.argv) > 3 else "exact"
_lib.set_arith("exact" if mode == "pc" else mode)
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tot_bad, worst = 0, 0.0
for it in range(n):
    V = int(rng.integers(2, 9)); C = int(rng.choice([1, 3, 8, 10, 16, 24, 32, 40])); D = int(rng.integers(1, 20))
    H = int(rng.integers(3, 100)); W = int(rng.integers(3, 200)); B = int(rng.integers(1, 3))
    geo = "rpc" if (mode == "pc" or rng.random() < 0.7) else "pinhole"
    jitter = bool(rng.random() < 0.6)
    if mode == "pc":
        V = min(V, 5)                                           # (the staged kernels: up to 4 source views take the folded cubics)
        kind = int(rng.integers(0, 4))
        jitter = kind == 3
    feats, gp, depth = T._inputs(B, V, C, D, H, W, seed=int(rng.integers(0, 10000)), jitter=jitter, geo=geo)
    if mode == "pc" and kind in (1, 2):
        depth = np.ascontiguousarray(np.broadcast_to(depth[:, :, None, None], (B, D, H, W))).copy()
        if kind == 2:
            for _ in range(int(rng.integers(1, 6))):
                depth[rng.integers(B), rng.integers(D), rng.integers(H), rng.integers(W)] += float(rng.choice([1.5, -0.75, np.nan]))
    r = rng.random() if mode != "pc" else 1.0
    if r < 0.25:                                                # boxes overflow
        lo, hi = (0.0, float(rng.uniform(2000, 40000))) if geo == "rpc" else (300.0, float(rng.uniform(900, 3000)))
        depth = np.broadcast_to(np.linspace(lo, hi, D, dtype=np.float32).reshape(1, D, 1, 1), (B, D, H, W)).copy()
    elif r < 0.35 and depth.ndim == 4:                          # some hypotheses are NaN / far outside
        depth = depth.copy()
        depth[..., ::3, ::5] = np.nan
        depth[..., 1::4, 2::7] = 1e9
    d0 = int(rng.integers(0, D)); d1 = int(rng.integers(d0 + 1, D + 1))
    want = orc.costvol_variance(feats, gp, depth, geo, d_begin=d0, d_end=d1)[:, :, d0:d1]
    if mode == "pc":
        got = T._build_raw(dev, feats, gp, depth, True, d0, d1)[0].cpu().numpy()
    else:
        got = warping.variance_cost_volume([torch.from_numpy(f).to(dev) for f in feats], torch.from_numpy(gp).to(dev), torch.from_numpy(depth).to(dev), geo,
                                           d_begin=d0, d_end=d1).cpu().numpy()
    if mode == "fused":
        with np.errstate(invalid="ignore"):
            ok = (np.abs(got.astype(np.float64) - want.astype(np.float64)) <= 1e-5 * np.maximum(1.0, np.abs(want.astype(np.float64)))) | (np.isnan(got) & np.isnan(want))
        nout = int((~ok).sum())
        tot_bad += nout
        if nout:
            print("MISMATCH it=%d B=%d V=%d C=%d D=%d[%d:%d] H=%d W=%d geo=%s jitter=%s: %d of %d outside the contract" % (it, B, V, C, D, d0, d1, H, W, geo, jitter, nout, got.size))
        continue
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    nbad = int((~same).sum())
    tot_bad += nbad
    if nbad:
        diff = float(np.nanmax(np.abs(got.astype(np.float64) - want.astype(np.float64))))
        worst = max(worst, diff)
        if nbad > max(1, 1e-4 * got.size) or not diff <= 2e-4:
            print("MISMATCH it=%d B=%d V=%d C=%d D=%d[%d:%d] H=%d W=%d geo=%s jitter=%s: %d of %d differ, max %.3g" % (it, B, V, C, D, d0, d1, H, W, geo, jitter, nbad, got.size, diff))
if mode == "fused":
    print("%d cases (fused): %d voxels outside 1e-5 max(1,|v|)" % (n, tot_bad))
else:
    print("%d cases: %d differing voxels in total, largest difference %.3g" % (n, tot_bad, worst))
