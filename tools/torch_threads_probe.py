import sys, time, os, torch
sys.path.insert(0, '.')
from oracle import torch_composite as tc
from satmvs_amd import rpc_synth
print("cpu_count", os.cpu_count(), "torch default threads", torch.get_num_threads())
V, C, D, H, W = 3, 32, 64, 384, 768
g = torch.Generator(device="cpu").manual_seed(0)
feats = [torch.randn((1, C, H, W), generator=g) for _ in range(V)]
rpc = torch.from_numpy(rpc_synth.make_view_rpcs(V, H, W, seed=0)[None])
depth = torch.linspace(0.0, 400.0, D).view(1, D, 1, 1).expand(1, D, H, W).contiguous()
for n in (8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    tc.variance_planes(feats, rpc, depth, 0, 1)
    t0 = time.perf_counter(); tc.variance_planes(feats, rpc, depth, 1, 3); dt = (time.perf_counter() - t0) / 2
    print("threads %3d: %.3f s/plane = %.2f Mvox/s" % (n, dt, H * W / dt / 1e6), flush=True)
