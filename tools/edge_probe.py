import sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from oracle import oracle as orc
orc.build()
from satmvs_amd.modules import warping
import test_hip_parity as T
dev = torch.device('cuda:0')
for cfg in [dict(B=2,V=5,C=16,D=5,H=33,W=72), dict(B=2,V=5,C=16,D=5,H=33,W=70), dict(B=1,V=3,C=16,D=5,H=33,W=70), dict(B=1,V=3,C=16,D=5,H=33,W=72), dict(B=1,V=3,C=16,D=8,H=33,W=68), dict(B=1,V=2,C=32,D=8,H=40,W=36)]:
    feats, rpc, depth = T._inputs(cfg['B'], cfg['V'], cfg['C'], cfg['D'], cfg['H'], cfg['W'], seed=3, jitter=True)
    want = orc.costvol_variance(feats, rpc, depth, 'rpc')
    got = warping.variance_cost_volume([T._t(f, dev) for f in feats], T._t(rpc, dev), T._t(depth, dev), 'rpc').cpu().numpy()
    bad = np.argwhere(got != want)
    print(cfg, 'nbad', len(bad), 'of', got.size, 'first', bad[:5].tolist() if len(bad) else '')
    if len(bad):
        xs = np.bincount(bad[:, 4], minlength=cfg['W']); ys = np.bincount(bad[:, 3], minlength=cfg['H'])
        print('  by x', np.nonzero(xs)[0].tolist()[:40]); print('  by y', np.nonzero(ys)[0].tolist()[:40])
