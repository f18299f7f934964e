import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd.modules import warping
from oracle import oracle as orc
g = np.load('tests/golden/costvol.npz')
dev = torch.device('cuda:0')
feats = [torch.from_numpy(f).to(dev) for f in g['feats']]
var = warping.variance_cost_volume(feats, torch.from_numpy(g['rpc']).to(dev), torch.from_numpy(g['depth']).to(dev), 'rpc').cpu().numpy()
want = g['variance_rpc']
bad = np.argwhere(var != want)
print('nbad', len(bad), 'max', np.abs(var-want).max())
if len(bad):
    for ax, nm in enumerate('bcdyx'):
        u, c = np.unique(bad[:, ax], return_counts=True)
        print(nm, dict(zip(u.tolist(), c.tolist())))
    b = bad[0]; print(b, var[tuple(b)], want[tuple(b)])
