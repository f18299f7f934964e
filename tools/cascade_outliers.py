"""Per-forward wall times of the inference cascade (what bench.py's extra.cfg3 reports the median / min / max of): where do the
occasional 2-3x forwards come from?  python tools/cascade_outliers.py [n]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from satmvs_amd import rpc_synth
from satmvs_amd.networks.casred import Infer_CascadeREDNet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
H, W = 384, 768
torch.manual_seed(0)
net = Infer_CascadeREDNet("rpc", ndepths=[48, 32, 8]).to(dev).eval()
imgs = torch.randn(1, 3, 3, H, W, device=dev)
rpc = rpc_synth.make_view_rpcs(3, H, W, seed=0)[None]
pm = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev), "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev), "stage3": torch.from_numpy(rpc).to(dev)}
dv = torch.tensor([[0.0, 400.0]], device=dev)
for mode in ("plain", "gc disabled", "events"):
    if mode == "gc disabled":
        gc.disable()
    with torch.no_grad():
        for _ in range(3):
            net(imgs, pm, dv)
        torch.cuda.synchronize()
        ts, hs = [], []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            a.record()
            net(imgs, pm, dv)
            b.record()
            h = time.perf_counter() - t0
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            hs.append((h * 1e3, a.elapsed_time(b)))
    print(mode, "wall ms:", " ".join("%.1f" % t for t in ts))
    big = [i for i, t in enumerate(ts) if t > 1.5 * np.median(ts)]
    print("   median %.2f  outliers at %s: host-issue ms / device ms = %s" % (np.median(ts), big, [(round(hs[i][0], 1), round(hs[i][1], 1)) for i in big]))
    gc.enable()
