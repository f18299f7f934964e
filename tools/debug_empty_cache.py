"""Use-after-free probe of the native training operators: a RED training step at 768x384 with torch.cuda.empty_cache() (hipFree of every
cached block) fired after every element-wise operator, no synchronisation in between.  A kernel still reading a tensor that Python has
already released would fault here."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satmvs_amd.modules import module as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
calls = [0]
for cls in (M._GroupNorm1Fn, M._GruMulCatFn, M._GruBlendFn):
    fwd, bwd = cls.forward, cls.backward
    def wrap(f):
        def g(*a):
            out = f(*a)
            calls[0] += 1
            torch.cuda.empty_cache()
            return out
        return staticmethod(g)
    cls.forward, cls.backward = wrap(fwd), wrap(bwd)
for C, H, W in ((8, 384, 768), (32, 96, 192)):
    reg = M.slice_RED_Regularization(C, 8).to(dev).train()
    st = reg.initial_states(1, H, W, dev)
    for plane in range(3):
        cost = torch.randn(1, C, H, W, device=dev, requires_grad=True)
        o = reg(cost, *st)
        st = [s for s in o[1:]] if len(o) > 1 else st
        o[0].mean().backward(retain_graph=False) if plane == 2 else None
        st = [s.detach() for s in st]
    torch.cuda.synchronize()
    print("ok %dx%d after %d native calls with empty_cache()" % (H, W, calls[0]), flush=True)
