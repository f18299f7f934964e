#!/usr/bin/env python
"""The training step of tools/bench_train_step.py captured in one HIP graph (forward, loss, backward, RMSprop) and replayed:
the eager step is bound by the host (32 k launches per step), the graph by the GPU.
    python tools/bench_train_graph.py [steps] [casred|casmvs|ucs] [rpc|pinhole] [ndepths, e.g. 64,32,8]"""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd import rpc_synth
from satmvs_amd.networks.casred import CascadeREDNet

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = os.environ.get("SMVS_CUDNN_BENCHMARK", "0") == "1"      # train.py:21 sets it; MIOpen then searches per shape
H, W, nd = 384, 768, [48, 32, 8]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.manual_seed(0)
model = sys.argv[2] if len(sys.argv) > 2 else "casred"
geo = sys.argv[3] if len(sys.argv) > 3 else "rpc"                              # the reference's train.py defaults: --model casmvs --geo_model pinhole --ndepths 64,32,8
if len(sys.argv) > 4:
    nd = [int(v) for v in sys.argv[4].split(",")]
if model == "casred":
    net = CascadeREDNet(geo, min_interval=2.5, ndepths=nd)
elif model == "casmvs":
    from satmvs_amd.networks.casmvs import CascadeMVSNet
    net = CascadeMVSNet(geo, min_interval=2.5, ndepths=nd)
else:
    from satmvs_amd.networks.ucs import UCSNet
    net = UCSNet(geo, stage_configs=nd)
net = net.to(dev).train()
opt = torch.optim.RMSprop(net.parameters(), lr=1e-3, alpha=0.9, capturable=True)
imgs = torch.randn(1, 3, 3, H, W, device=dev)
rpc = rpc_synth.make_view_rpcs(3, H, W, seed=0)[None]
pm = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev), "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev),
      "stage3": torch.from_numpy(rpc).to(dev)}
dv = torch.tensor([[0.0, 400.0]], device=dev)
if geo == "pinhole":                                                           # K.E matrices as dataset/virdataset.py hands them over, intrinsics rows scaled per stage
    import numpy as np
    full = np.zeros((1, 3, 4, 4))
    for v in range(3):
        f = 1.1 * W
        K = np.array([[f, 0, W / 2.0, 0], [0, f, H / 2.0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1]])
        E = np.eye(4)
        E[:3, 3] = [25.0 * v * (-1) ** v, 3.0 * v, 0.5 * v]
        full[0, v] = K @ E

    def _scaled(s):
        m = full.copy()
        m[:, :, :2, :] /= s
        return torch.from_numpy(m).to(dev)
    pm = {"stage1": _scaled(4), "stage2": _scaled(2), "stage3": _scaled(1)}
    dv = torch.tensor([[400.0, 700.0]], device=dev)
gt = {s: torch.full((1, H // k, W // k), 550.0 if geo == "pinhole" else 200.0, device=dev) for s, k in (("stage1", 4), ("stage2", 2), ("stage3", 1))}


def loss_fn(out, gt):
    return sum(w * F.smooth_l1_loss(out[s]["depth"], gt[s], reduction="mean") for s, w in (("stage1", 0.5), ("stage2", 1.0), ("stage3", 2.0)))


from satmvs_amd.train_graph import GraphedTrainStep
step = GraphedTrainStep(net, opt, loss_fn)
t0 = time.perf_counter()
loss, _ = step(imgs, pm, dv, gt)
torch.cuda.synchronize()
print("captured (3 warm-up steps + capture + first replay) in %.1f s" % (time.perf_counter() - t0))
ts = []
for _ in range(steps):
    t0 = time.perf_counter()
    loss, _ = step(imgs, pm, dv, gt)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(model + " " + geo + " graphed training step (satmvs_amd.train_graph), 3-view 768x384, planes %s: median %.1f ms (min %.1f, max %.1f), loss %.4f"
      % (nd, ts[len(ts) // 2], ts[0], ts[-1], float(loss)))
