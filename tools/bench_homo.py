import sys, os, json
sys.path.insert(0, '.')
import torch, numpy as np, bench
from satmvs_amd import _lib
from satmvs_amd.modules import warping
dev = torch.device("cuda:0")
V, C, D, H, W = 3, 32, 64, 384, 768
g = torch.Generator(device="cpu").manual_seed(0)
feats = [torch.randn((1, C, H, W), generator=g).to(dev) for _ in range(V)]
proj = np.zeros((1, V, 4, 4))
for v in range(V):
    f = 1.1 * W
    K = np.array([[f, 0, W / 2.0, 0], [0, f, H / 2.0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1]]); E = np.eye(4)
    E[:3, 3] = [25.0 * v * (-1) ** v, 3.0 * v, 0.5 * v]; proj[0, v] = K @ E
projt = torch.from_numpy(proj).to(dev)
depth = torch.linspace(400.0, 700.0, D).view(1, D, 1, 1).expand(1, D, H, W).contiguous().to(dev)
step = lambda: warping.variance_cost_volume(feats, projt, depth, "pinhole")
for _ in range(200): step()
_, ms = bench.time_steps(step, 300)
print(os.environ.get("SMVS_LIB_PATH", "default").split("/")[-1], "cfg5 pinhole volume: %.4f ms" % ms)
