import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd.modules.module import CostRegNet
dev = torch.device("cuda:0"); torch.manual_seed(0)
net = CostRegNet(32, 8).to(dev).eval(); x = torch.randn(1, 32, 48, 96, 192, device=dev)
with torch.no_grad():
    for _ in range(4): y = net(x)
torch.cuda.synchronize()
