#!/usr/bin/env python
"""Soak of the native casmvs / ucs training path: N eager steps on fresh random samples of a small tile, watching for non-finite losses,
parameters and BatchNorm buffers, and for memory growth (the pack cache, the zero-fill arenas and the scratch buffers must not leak).
    python tools/soak_train3d.py [steps] [casmvs|ucs]"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd import rpc_synth

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
model = sys.argv[2] if len(sys.argv) > 2 else "casmvs"
H, W, nd = 128, 256, [16, 8, 8]
torch.manual_seed(0)
if model == "casmvs":
    from satmvs_amd.networks.casmvs import CascadeMVSNet
    net = CascadeMVSNet("rpc", min_interval=2.5, ndepths=nd)
else:
    from satmvs_amd.networks.ucs import UCSNet
    net = UCSNet("rpc", stage_configs=nd)
net = net.to(dev).train()
opt = torch.optim.RMSprop(net.parameters(), lr=1e-4, alpha=0.9)
rpc = rpc_synth.make_view_rpcs(3, H, W, seed=0)[None]
pm = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev), "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev),
      "stage3": torch.from_numpy(rpc).to(dev)}
dv = torch.tensor([[0.0, 400.0]], device=dev)
gt = {s: torch.full((1, H // k, W // k), 200.0, device=dev) for s, k in (("stage1", 4), ("stage2", 2), ("stage3", 1))}
mem = []
for it in range(steps):
    imgs = torch.randn(1, 3, 3, H, W, device=dev)
    opt.zero_grad(set_to_none=True)
    out = net(imgs, pm, dv)
    loss = sum(w * F.smooth_l1_loss(out[s]["depth"], gt[s]) for s, w in (("stage1", 0.5), ("stage2", 1.0), ("stage3", 2.0)))
    loss.backward()
    opt.step()
    if it % 50 == 0 or it == steps - 1:
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(loss)) and all(bool(torch.isfinite(p).all()) for p in net.parameters()) and \
            all(bool(torch.isfinite(b.float()).all()) for b in net.buffers())
        mem.append(torch.cuda.memory_allocated() / 2 ** 20)
        print("step %4d  loss %.4f  finite %s  allocated %.1f MB  reserved %.1f MB" % (it, float(loss), ok, mem[-1], torch.cuda.memory_reserved() / 2 ** 20))
        if not ok:
            print("MISMATCH: non-finite values at step %d" % it)
            break
print("%s: %d steps, allocated memory first / last checkpoint %.1f / %.1f MB%s" % (model, steps, mem[0], mem[-1], "" if mem[-1] <= mem[1 if len(mem) > 1 else 0] * 1.05 + 1 else "  MISMATCH: memory grows"))
