import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from satmvs_amd.modules.module import FeatureNet, CostRegNet, slice_RED_Regularization
dev = torch.device("cuda:0")
torch.manual_seed(1)
def rand_bn(net):
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.7, 1.3); m.bias.data.normal_(0, 0.1)
worst = 0.0
for arch in ("unet", "fpn"):
    net = FeatureNet(8, 3, 4, arch).to(dev).eval(); rand_bn(net)
    for (n, h, w) in ((1, 8, 8), (3, 72, 136), (2, 132, 200), (1, 260, 68)):
        x = torch.randn(n, 3, h, w, device=dev)
        with torch.no_grad():
            a = net(x)
            os.environ["SMVS_FEATNET_TORCH"] = "1"; b = net(x); del os.environ["SMVS_FEATNET_TORCH"]
        e = max(float((a[k] - b[k]).abs().max()) for k in a); worst = max(worst, e)
        print("featnet", arch, (n, h, w), "max diff %.2e" % e)
for c in (8, 16, 32):
    net = CostRegNet(c, 8).to(dev).eval(); rand_bn(net)
    for (d, h, w) in ((8, 8, 8), (16, 24, 72), (8, 40, 136), (24, 16, 200)):
        x = torch.randn(1, c, d, h, w, device=dev)
        with torch.no_grad():
            a = net(x)
            os.environ["SMVS_COSTREG_TORCH"] = "1"; b = net(x); del os.environ["SMVS_COSTREG_TORCH"]
        e = float((a - b).abs().max()) / max(1.0, float(b.abs().max())); worst = max(worst, e)
        print("costreg C%d" % c, (d, h, w), "max rel diff %.2e" % e)
for c in (8, 16, 32):
    net = slice_RED_Regularization(c, 8).to(dev).eval()
    for (b_, h, w) in ((1, 8, 8), (2, 24, 72), (1, 40, 136), (1, 16, 200), (1, 264, 520)):
        x = torch.randn(b_, c, h, w, device=dev)
        st = net.initial_states(b_, h, w, dev)
        with torch.no_grad():
            a = net(x, *st); a2 = net(x * 0.7, *a[1:])
            os.environ["SMVS_RED_TORCH"] = "1"
            st = net.initial_states(b_, h, w, dev); r = net(x, *st); r2 = net(x * 0.7, *r[1:]); del os.environ["SMVS_RED_TORCH"]
        e = max(float((p - q).abs().max()) for p, q in zip(a2, r2)); worst = max(worst, e)
        print("red C%d" % c, (b_, h, w), "max diff %.2e" % e)
print("WORST", worst)
