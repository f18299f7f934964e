"""Four native CostRegNet forwards per stage shape (for kernel traces): python tools/run_costreg_once.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd.modules.module import CostRegNet
dev = torch.device("cuda:0"); torch.manual_seed(0)
for C, D, H, W in ((32, 48, 96, 192), (16, 32, 192, 384), (8, 8, 384, 768)):
    net = CostRegNet(C, 8).to(dev).eval(); x = torch.randn(1, C, D, H, W, device=dev)
    with torch.no_grad():
        for _ in range(4): y = net(x)
    torch.cuda.synchronize()
