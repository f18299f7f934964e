#!/usr/bin/env python
"""The training step of tools/bench_train_step.py captured in one HIP graph (forward, loss, backward, RMSprop) and replayed:
the eager step is bound by the host (32 k launches per step), the graph by the GPU.
    python tools/bench_train_graph.py [steps] [casred|casmvs|ucs]"""
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
if model == "casred":
    net = CascadeREDNet("rpc", min_interval=2.5, ndepths=nd)
elif model == "casmvs":
    from satmvs_amd.networks.casmvs import CascadeMVSNet
    net = CascadeMVSNet("rpc", min_interval=2.5, ndepths=nd)
else:
    from satmvs_amd.networks.ucs import UCSNet
    net = UCSNet("rpc", stage_configs=nd)
net = net.to(dev).train()
opt = torch.optim.RMSprop(net.parameters(), lr=1e-3, alpha=0.9, capturable=True)
imgs = torch.randn(1, 3, 3, H, W, device=dev)
rpc = rpc_synth.make_view_rpcs(3, H, W, seed=0)[None]
pm = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev), "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev),
      "stage3": torch.from_numpy(rpc).to(dev)}
dv = torch.tensor([[0.0, 400.0]], device=dev)
gt = {s: torch.full((1, H // k, W // k), 200.0, device=dev) for s, k in (("stage1", 4), ("stage2", 2), ("stage3", 1))}


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
print(model + " graphed training step (satmvs_amd.train_graph), 3-view 768x384, planes %s: median %.1f ms (min %.1f, max %.1f), loss %.4f"
      % (nd, ts[len(ts) // 2], ts[0], ts[-1], float(loss)))
