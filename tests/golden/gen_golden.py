#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/*.npz by IMPORTING the Python reference.

Runs only in the build container (needs /root/reference, read-only).  The fixtures are data:
seeded inputs and the reference's outputs.  No reference source travels to the GPU box.

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

Inputs are synthesised by satmvs_amd/rpc_synth.py (our own host code); every expected output
comes from the reference's functions, file:line noted per fixture.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("SATMVS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

# The reference hard-codes .cuda(); on this CPU-only container make it a no-op.
torch.Tensor.cuda = lambda self, *a, **k: self
torch.set_num_threads(4)

import warnings  # noqa: E402

warnings.filterwarnings("ignore")

from modules import warping as ref_warping  # noqa: E402
from modules import module as ref_module  # noqa: E402
from modules import depth_range as ref_depth_range  # noqa: E402
from networks import casred as ref_casred  # noqa: E402
from networks import casmvs as ref_casmvs  # noqa: E402
from networks import ucs as ref_ucs  # noqa: E402
from tools.RPCCore import RPCModelParameter  # noqa: E402
from tools.iccv_solver import solve_iccv  # noqa: E402

from satmvs_amd import rpc_synth  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KB  %s" % (name, os.path.getsize(path) / 1024.0, sorted(arrays)))


def np_state(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def ref_rpcs(num_views, H, W, seed, batch=1):
    """(B,V,170) via OUR synthesiser for the direct part and the REFERENCE's ICCV inverse fit."""
    out = np.zeros((batch, num_views, 170))
    for b in range(batch):
        for v in range(num_views):
            d = rpc_synth.make_direct_rpc(H, W, seed=seed * 101 + 17 * b + v,
                                          tilt=rpc_synth._DEFAULT_TILTS[(v + b) % 7])
            m = RPCModelParameter(d.copy())
            m.Calculate_Inverse_RPC()                      # tools/RPCCore.py:188
            out[b, v] = np.array(m.get_data())
    return out


def height_volume(B, D, H, W, seed, lo=0.0, hi=400.0, jitter=6.0):
    """Stage-2/3-like per-pixel hypotheses: plane-wise linspace + smooth per-pixel offset."""
    rng = np.random.default_rng(seed)
    base = np.linspace(lo, hi, D, dtype=np.float64).reshape(1, D, 1, 1)
    yy, xx = np.meshgrid(np.linspace(0, 3, H), np.linspace(0, 5, W), indexing="ij")
    bump = jitter * (np.sin(yy + rng.uniform(0, 3)) * np.cos(xx + rng.uniform(0, 3)))
    return (base + bump[None, None] + rng.normal(0, 0.3, (B, D, H, W))).astype(np.float32)


# ------------------------------------------------------------------------------------------
def gen_iccv():
    """tools/iccv_solver.py:42-64 known answers + the inverse fit of tools/RPCCore.py:188-240."""
    A1 = np.array([[94.61, -22.11, -11.45, -6.96], [-22.11, 70.51, -6.95, -8.42],
                   [-11.45, -6.95, 96.09, -20.21], [-6.96, -8.42, -20.21, 66.63]])
    L1 = np.array([-43.52, 178.81, -120.11, -30.07])
    A2 = np.array([[5, -2, -1, -2], [-2, 5, -1, -2], [-1, -1, 3, -1], [-2, -2, -1, 5]], float)
    L2 = np.array([-11, 10, -2, 3], float)
    x1, t1 = solve_iccv(A1, L1)
    x2, t2 = solve_iccv(A2, L2)
    direct = np.stack([rpc_synth.make_direct_rpc(128, 256, seed=s, tilt=t)
                       for s, t in ((3, 0.0), (4, 0.05), (5, -0.08))])
    full = []
    for d in direct:
        m = RPCModelParameter(d.copy())
        m.Calculate_Inverse_RPC()
        full.append(np.array(m.get_data()))
    save("iccv", A1=A1, L1=L1, x1=x1, it1=np.int64(t1), A2=A2, L2=L2, x2=x2, it2=np.int64(t2),
         direct=direct, full=np.stack(full))


def gen_project():
    """RPC_Photo2Obj / RPC_Obj2Photo (modules/warping.py:255,218) and the numpy twins
    (tools/RPCCore.py:424-489) on random points."""
    rng = np.random.default_rng(7)
    H, W, n = 128, 256, 600
    rpc = ref_rpcs(3, H, W, seed=2)[0]                     # (3,170)
    samp = rng.uniform(-10, W + 10, n)
    line = rng.uniform(-10, H + 10, n)
    hei = rng.uniform(-20, 420, n)
    out = {"rpc": rpc, "samp": samp, "line": line, "hei": hei}
    for v in range(3):
        r = torch.from_numpy(rpc[v:v + 1])
        coef = torch.ones((1, n, 20), dtype=torch.double)
        lat, lon = ref_warping.RPC_Photo2Obj(torch.from_numpy(samp)[None], torch.from_numpy(line)[None],
                                             torch.from_numpy(hei)[None], r, coef)
        s2, l2 = ref_warping.RPC_Obj2Photo(lat, lon, torch.from_numpy(hei)[None], r, coef)
        m = RPCModelParameter(rpc[v].copy())
        nlat, nlon = m.RPC_PHOTO2OBJ(samp, line, hei)
        ns, nl = m.RPC_OBJ2PHOTO(nlat, nlon, hei)
        out.update({"lat%d" % v: lat[0].numpy(), "lon%d" % v: lon[0].numpy(),
                    "samp_back%d" % v: s2[0].numpy(), "line_back%d" % v: l2[0].numpy(),
                    "np_lat%d" % v: nlat, "np_lon%d" % v: nlon, "np_samp%d" % v: ns, "np_line%d" % v: nl})
    save("rpc_project", **out)


def gen_grid_sample():
    """F.grid_sample(bilinear, zeros) with the default align_corners (modules/warping.py:358)."""
    torch.manual_seed(11)
    inp = torch.randn(2, 3, 19, 23)
    grid = torch.rand(2, 16, 40, 2) * 2.6 - 1.3
    grid[0, 0, :4, 0] = torch.tensor([-1.0, 1.0, float("nan"), 1e30])
    out = torch.nn.functional.grid_sample(inp, grid, mode="bilinear", padding_mode="zeros")
    save("grid_sample", inp=inp.numpy(), grid=grid.numpy(), out=out.numpy())


def gen_rpc_warp():
    """rpc_warping (modules/warping.py:310-365): 4-D and 2-D depth_values, B=2, plus the f64
    intermediates of RPC_Photo2Obj/RPC_Obj2Photo on the same lattice."""
    B, C, D, H, W = 2, 8, 6, 32, 64
    torch.manual_seed(5)
    rpc = ref_rpcs(2, H, W, seed=9, batch=B)               # (B,2,170): view0=ref, view1=src
    src_fea = torch.randn(B, C, H, W)
    dv4 = height_volume(B, D, H, W, seed=1)
    dv2 = np.linspace(0, 400, D, dtype=np.float32)[None].repeat(B, 0) + np.array([[0.0], [7.5]], np.float32)
    ref_r, src_r = torch.from_numpy(rpc[:, 0]), torch.from_numpy(rpc[:, 1])
    coef = torch.ones((B, H * W * D, 20), dtype=torch.double)
    w4 = ref_warping.rpc_warping(src_fea, src_r, ref_r, torch.from_numpy(dv4), coef)
    w2 = ref_warping.rpc_warping(src_fea, src_r, ref_r, torch.from_numpy(dv2), coef)
    # intermediates exactly as warping.py:323-341 builds them
    y, x = torch.meshgrid([torch.arange(0, H, dtype=torch.double), torch.arange(0, W, dtype=torch.double)])
    y = y.contiguous().view(1, 1, H, W).repeat(B, D, 1, 1).view(B, -1)
    x = x.contiguous().view(1, 1, H, W).repeat(B, D, 1, 1).view(B, -1)
    h = torch.from_numpy(dv4).view(B, -1).double()
    lat, lon = ref_warping.RPC_Photo2Obj(x, y, h, ref_r, coef)
    samp, line = ref_warping.RPC_Obj2Photo(lat, lon, h, src_r, coef)
    save("rpc_warp", rpc=rpc, src_fea=src_fea.numpy(), depth4=dv4, depth2=dv2,
         warped4=w4.numpy(), warped2=w2.numpy(),
         lat=lat.view(B, D, H, W).numpy(), lon=lon.view(B, D, H, W).numpy(),
         samp=samp.view(B, D, H, W).numpy(), line=line.view(B, D, H, W).numpy())


def _qc_dict(rpc_b):
    """(B,170) -> the dict of dataset/data_io.py:123-150, built with the reference-layout tensor."""
    keys = ["line_off", "samp_off", "lat_off", "lon_off", "height_off",
            "line_scale", "samp_scale", "lat_scale", "lon_scale", "height_scale"]
    d = {k: torch.from_numpy(np.ascontiguousarray(rpc_b[:, i])) for i, k in enumerate(keys)}
    names = ["line_num", "line_den", "samp_num", "samp_den", "lat_num", "lat_den", "lon_num", "lon_den"]
    for j, nm in enumerate(names):
        t = np.stack([rpc_synth.coeffs_to_qc_tensor(r[10 + 20 * j: 30 + 20 * j]) for r in rpc_b])
        d[nm + "_tensor"] = torch.from_numpy(t)
    return d


def gen_qc():
    """rpc_warping_enisum (modules/warping.py:139-178) with the QC tensors of
    dataset/data_io.py:95-120; also pins coeffs_to_qc_tensor against the reference to_tensor."""
    B, C, D, H, W = 1, 4, 3, 16, 24
    torch.manual_seed(6)
    rpc = ref_rpcs(2, H, W, seed=21, batch=B)
    src_fea = torch.randn(B, C, H, W)
    dv4 = height_volume(B, D, H, W, seed=2)
    # reference to_tensor, executed from its source with the GDAL import stripped
    src = open(os.path.join(REF, "dataset", "data_io.py")).read()
    start = src.index("def to_tensor")
    end = src.index("def load_rpc_as_qc_tensor")
    ns = {"np": np}
    exec(src[start:end], ns)
    c20 = np.random.default_rng(3).normal(size=20)
    t_ref = ns["to_tensor"](c20)
    assert np.array_equal(t_ref, rpc_synth.coeffs_to_qc_tensor(c20)), "QC tensor layout differs from reference"
    out = ref_warping.rpc_warping_enisum(src_fea, _qc_dict(rpc[:, 1]), _qc_dict(rpc[:, 0]), torch.from_numpy(dv4))
    save("rpc_warp_qc", rpc=rpc, src_fea=src_fea.numpy(), depth4=dv4, warped=out.numpy(),
         qc_c20=c20, qc_tensor=t_ref)


def _pinhole_mats(V, H, W, seed, batch=1):
    """K.E 4x4 'projection matrices' as dataset/virdataset.py:96-105 hands them over."""
    rng = np.random.default_rng(seed)
    out = np.zeros((batch, V, 4, 4))
    for b in range(batch):
        for v in range(V):
            f = 1.1 * W
            K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
            ang = rng.normal(0, 0.02, 3) * (v > 0)
            Rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
            Ry = np.array([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]])
            Rz = np.array([[np.cos(ang[2]), -np.sin(ang[2]), 0], [np.sin(ang[2]), np.cos(ang[2]), 0], [0, 0, 1]])
            R = Rz @ Ry @ Rx
            t = np.array([30.0 * v * (-1) ** v, 4.0 * v, 1.0 * v]) + rng.normal(0, 1.0, 3) * (v > 0)
            E = np.eye(4)
            E[:3, :3], E[:3, 3] = R, t
            P = np.eye(4)
            P[:3, :4] = K @ E[:3, :4]
            out[b, v] = P
    return out


def gen_homo():
    """homo_warping (modules/warping.py:6-44), 2-D and 4-D depth."""
    B, C, D, H, W = 2, 6, 5, 24, 40
    torch.manual_seed(8)
    proj = _pinhole_mats(2, H, W, seed=4, batch=B)
    src_fea = torch.randn(B, C, H, W)
    d2 = np.linspace(400, 700, D, dtype=np.float32)[None].repeat(B, 0)
    d4 = (d2[:, :, None, None] + height_volume(B, D, H, W, seed=3, lo=0, hi=0, jitter=15.0)).astype(np.float32)
    sp, rp = torch.from_numpy(proj[:, 1]), torch.from_numpy(proj[:, 0])
    w2 = ref_warping.homo_warping(src_fea, sp, rp, torch.from_numpy(d2))
    w4 = ref_warping.homo_warping(src_fea, sp, rp, torch.from_numpy(d4))
    composed = torch.matmul(sp, torch.inverse(rp)).numpy()
    save("homo_warp", proj=proj, src_fea=src_fea.numpy(), depth2=d2, depth4=d4,
         warped2=w2.numpy(), warped4=w4.numpy(), composed=composed)


class _Capture(torch.nn.Module):
    """Stand-in regulariser that records its input (the variance volume) and returns a peaky cost."""

    def __init__(self, lam=8.0):
        super().__init__()
        self.lam = lam
        self.seen = None

    def forward(self, vol):
        self.seen = vol.detach().clone()
        return -self.lam * vol.mean(1)                     # (B,D,H,W)


def gen_costvol():
    """Body of compute_depth_when_train (networks/casred.py:22-62): variance volume, softmax,
    depth_regression, max-prob -- RPC and pinhole, 3 views."""
    B, C, D, H, W, V = 1, 8, 8, 32, 64, 3
    torch.manual_seed(12)
    feats = [torch.randn(B, C, H, W) for _ in range(V)]
    rpc = ref_rpcs(V, H, W, seed=31, batch=B)
    dv = height_volume(B, D, H, W, seed=5)
    cap = _Capture()
    out = ref_casred.compute_depth_when_train(feats, torch.from_numpy(rpc), torch.from_numpy(dv), D, cap, "rpc", False)
    var_rpc = cap.seen.numpy()
    reg_rpc = (-cap.lam * cap.seen.mean(1)).numpy()
    proj = _pinhole_mats(V, H, W, seed=6, batch=B)
    dvp = np.linspace(400, 700, D, dtype=np.float32).reshape(1, D, 1, 1).repeat(H, 2).repeat(W, 3)
    out_p = ref_casred.compute_depth_when_train(feats, torch.from_numpy(proj), torch.from_numpy(dvp), D, cap, "pinhole", False)
    save("costvol", feats=np.stack([f.numpy() for f in feats]), rpc=rpc, depth=dv, variance_rpc=var_rpc,
         reg_rpc=reg_rpc, depth_rpc=out["depth"].numpy(), conf_rpc=out["photometric_confidence"].numpy(),
         proj=proj, depth_pin=dvp, variance_pin=cap.seen.numpy(),
         depth_pin_out=out_p["depth"].numpy(), conf_pin_out=out_p["photometric_confidence"].numpy())


def gen_photo():
    """A PHOTO-CONSISTENT problem with a peaky regulariser: three views of one synthetic textured surface,
    rendered through their RPCs, pushed through the reference's FeatureNet (seeded), then the reference's
    compute_depth_when_train (networks/casred.py:10-62) with a stand-in regulariser that returns -lam * mean variance.
    The softmax is sharply peaked at the plane where the warped features agree, so the regressed height follows the
    surface and reacts to sub-pixel warp errors (with random features it is a blur of all planes)."""
    B, D, H, W, V = 1, 16, 64, 128, 3
    rpc = ref_rpcs(V, H, W, seed=71, batch=B)
    rng = np.random.default_rng(72)
    lat0, lon0, ls, os_ = rpc[0, 0, 2], rpc[0, 0, 3], rpc[0, 0, 7], rpc[0, 0, 8]

    def surface(lat, lon):
        u, v = (lat - lat0) / ls, (lon - lon0) / os_
        return 200.0 + 22.0 * np.sin(2.1 * u + 0.4) * np.cos(1.7 * v - 0.3) + 9.0 * np.sin(4.3 * v + 1.0)

    waves = [(rng.uniform(20, 220) * rng.choice([-1, 1]), rng.uniform(20, 220), rng.uniform(0, 6.28), rng.uniform(0.3, 1.0)) for _ in range(3 * 28)]

    def texture(ch, lat, lon):
        u, v = (lat - lat0) / ls, (lon - lon0) / os_
        return sum(a * np.sin(fu * u + fv * v + ph) for fu, fv, ph, a in waves[28 * ch:28 * ch + 28]) / 4.0

    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    imgs = np.zeros((B, V, 3, H, W), np.float32)
    truth = None
    for v in range(V):
        h = np.full((H, W), 200.0)
        for _ in range(12):                                   # ray / surface intersection by fixed-point iteration
            lat, lon = rpc_synth.photo2obj(rpc[0, v], xx.ravel(), yy.ravel(), h.ravel())
            h = surface(lat, lon).reshape(H, W)
        lat, lon = rpc_synth.photo2obj(rpc[0, v], xx.ravel(), yy.ravel(), h.ravel())
        for ch in range(3):
            imgs[0, v, ch] = texture(ch, lat, lon).reshape(H, W)
        if v == 0:
            truth = h.astype(np.float32)
    torch.manual_seed(73)
    fnet = ref_module.FeatureNet(base_channels=8, stride=4, num_stage=3, arch_mode="unet").eval()
    with torch.no_grad():
        feats = [fnet(torch.from_numpy(imgs[:, v]))["stage3"] for v in range(V)]       # (B,8,H,W) each
        dv = height_volume(B, D, H, W, seed=74, lo=160.0, hi=240.0, jitter=1.5)
        cap = _Capture(lam=float(os.environ.get("PHOTO_LAM", "2e6")))
        out = ref_casred.compute_depth_when_train(feats, torch.from_numpy(rpc), torch.from_numpy(dv), D, cap, "rpc", False)
    depth = out["depth"].numpy()
    vm = cap.seen.mean(1)
    print("photo: variance mean over channels: min over planes %.4g, mean %.4g" % (float(vm.min(1)[0].mean()), float(vm.mean())))
    err = np.abs(depth[0] - truth)[8:-8, 8:-8]
    print("photo: median |height - truth| %.2f m, 90%% %.2f m, mean confidence %.3f" % (
        np.median(err), np.percentile(err, 90), float(out["photometric_confidence"].mean())))
    assert np.median(err) < 6.0 and float(out["photometric_confidence"].mean()) > 0.4, "not peaky / not locked on the surface"
    save("photo", feats=np.stack([f.numpy() for f in feats]), rpc=rpc, depth_values=dv, lam=np.float32(cap.lam),
         depth=depth, conf=out["photometric_confidence"].numpy(), truth=truth)


def gen_grad():
    """Gradients through the reference's differentiable path: d loss / d features of compute_depth_when_train
    (networks/casred.py:10-62: rpc_warping's grid_sample backward, the in-place variance accumulation, softmax,
    depth_regression) with a differentiable stand-in regulariser (-lam * mean variance), loss = sum(w * depth)."""
    B, C, D, H, W, V = 1, 8, 6, 16, 32, 3
    torch.manual_seed(81)
    feats = [torch.randn(B, C, H, W, requires_grad=True) for _ in range(V)]
    rpc = ref_rpcs(V, H, W, seed=82, batch=B)
    dv = height_volume(B, D, H, W, seed=83)
    wmap = torch.randn(B, H, W)
    lam = 4.0
    out = ref_casred.compute_depth_when_train(feats, torch.from_numpy(rpc), torch.from_numpy(dv), D,
                                              lambda vol: -lam * vol.mean(1), "rpc", False)
    loss = (out["depth"] * wmap).sum()
    loss.backward()
    save("grad", feats=np.stack([f.detach().numpy() for f in feats]), rpc=rpc, depth_values=dv, wmap=wmap.numpy(),
         lam=np.float32(lam), depth=out["depth"].detach().numpy(), loss=np.float64(loss.item()),
         grads=np.stack([f.grad.numpy() for f in feats]))


def gen_io():
    """dataset/data_io.py:17-92 (load_pfm, load_rpc_as_array): the module needs GDAL to import, so the reader functions
    are exec'd from their source lines with nothing of them stored.  A PFM and an .rpc text written by OUR writers are read
    by the REFERENCE's readers; the files and what the reference read are the fixture."""
    import re as _re  # noqa: F401
    src = open(os.path.join(REF, "dataset", "data_io.py")).read()
    a, b = src.index("def load_pfm(fname):"), src.index("def save_pfm(file, image, scale=1):")
    c, d = src.index("def load_rpc_as_array(filepath):"), src.index("def to_tensor(data):")
    ns = {"np": np, "re": __import__("re"), "os": os, "sys": sys}
    exec(src[a:b] + src[c:d], ns)
    from satmvs_amd import data_io
    rng = np.random.default_rng(91)
    img = (rng.standard_normal((9, 14)) * 50 + 200).astype(np.float32)
    rpc = ref_rpcs(1, 32, 64, seed=92)[0, 0]
    io_dir = os.path.join(HERE, "io")
    os.makedirs(io_dir, exist_ok=True)
    data_io.save_pfm(os.path.join(io_dir, "height.pfm"), img)
    data_io.save_rpc(os.path.join(io_dir, "view.rpc"), rpc)
    ref_img = ns["load_pfm"](os.path.join(io_dir, "height.pfm"))
    ref_rpc, hmax, hmin = ns["load_rpc_as_array"](os.path.join(io_dir, "view.rpc"))
    assert np.array_equal(ref_img, img) and np.array_equal(ref_rpc, rpc)
    save("io", pfm=np.ascontiguousarray(ref_img), rpc=ref_rpc, h_max=np.float64(hmax), h_min=np.float64(hmin))


def gen_pred():
    """compute_depth_when_pred (networks/casred.py:161-238) with seeded slice_RED_Regularization
    weights, and compute_depth_when_train with RED_Regularization on the same weights."""
    B, C, D, H, W, V = 1, 8, 6, 32, 64, 3
    torch.manual_seed(13)
    feats = [torch.randn(B, C, H, W) * 0.5 for _ in range(V)]
    rpc = ref_rpcs(V, H, W, seed=41, batch=B)
    dv = height_volume(B, D, H, W, seed=7)
    reg = ref_module.slice_RED_Regularization(C, 8).eval()
    reg_train = ref_module.RED_Regularization(C, 8).eval()
    reg_train.load_state_dict(reg.state_dict())
    with torch.no_grad():
        o_pred = ref_casred.compute_depth_when_pred(feats, torch.from_numpy(rpc), torch.from_numpy(dv), D, reg, "rpc", False)
        o_train = ref_casred.compute_depth_when_train(feats, torch.from_numpy(rpc), torch.from_numpy(dv), D, reg_train, "rpc", False)
        # one slice step for the regulariser alone
        x = torch.randn(B, C, H, W)
        st = [torch.zeros(B, 8, H, W), torch.zeros(B, 16, H // 2, W // 2),
              torch.zeros(B, 32, H // 4, W // 4), torch.zeros(B, 64, H // 8, W // 8)]
        r1 = reg(x, *st)
        r2 = reg(x * 0.5, *r1[1:])
    arrays = {"w." + k: v for k, v in np_state(reg).items()}
    save("red_pred", feats=np.stack([f.numpy() for f in feats]), rpc=rpc, depth=dv,
         pred_depth=o_pred["depth"].numpy(), pred_conf=o_pred["photometric_confidence"].numpy(),
         train_depth=o_train["depth"].numpy(), train_conf=o_train["photometric_confidence"].numpy(),
         slice_x=x.numpy(), slice_out1=r1[0].numpy(), slice_out2=r2[0].numpy(),
         slice_s1=r2[1].numpy(), slice_s2=r2[2].numpy(), slice_s3=r2[3].numpy(), slice_s4=r2[4].numpy(),
         **arrays)


def gen_regress():
    """softmax + depth_regression + max (casred.py:58-62) and the streaming accumulators
    (casred.py:218-236) on random regulariser outputs."""
    B, D, H, W = 2, 7, 12, 20
    torch.manual_seed(14)
    reg = torch.randn(B, D, H, W) * 3
    dv = torch.from_numpy(height_volume(B, D, H, W, seed=8))
    p = torch.softmax(reg, dim=1)
    depth = ref_module.depth_regression(p, depth_values=dv)
    conf = p.max(1)[0]
    exp_sum = torch.zeros(B, 1, H, W, dtype=torch.double)
    dimg = torch.zeros_like(exp_sum)
    mx = torch.zeros_like(exp_sum)
    for d in range(D):                                     # casred.py:218-231, transcribed as a driver loop
        prob = reg[:, d:d + 1].double().exp()
        flag = (mx < prob).double()
        mx = flag * prob + (1 - flag) * mx
        dimg = dv[:, d:d + 1].double() * prob + dimg
        exp_sum = exp_sum + prob
    fes = exp_sum + 1e-10
    # casmvs / ucs flavour (window-4 confidence, casmvs.py:69-74; ucs standard deviation, ucs.py:73-74): the reference's
    # own DepthNet.forward / compute_depth run on a small pinhole problem whose "regulariser" ignores the volume and
    # returns the seeded `reg`, so everything after the regulariser is the reference's code, not a transcription.
    C = 4
    feats = [torch.randn(B, C, H, W) for _ in range(2)]
    proj = torch.eye(4, dtype=torch.double).repeat(B, 2, 1, 1)
    fake_reg = lambda vol: reg.unsqueeze(1)                                   # noqa: E731
    with torch.no_grad():
        cas = ref_casmvs.DepthNet()(feats, proj, dv, D, fake_reg, "pinhole")
        ucs = ref_ucs.compute_depth(feats, proj, dv, fake_reg, 1.5, "pinhole", False)
    assert torch.equal(cas["depth"], depth) and torch.equal(ucs["depth"], depth)
    assert torch.equal(cas["photometric_confidence"], ucs["photometric_confidence"])
    save("regress", reg=reg.numpy(), depth_values=dv.numpy(), sm_depth=depth.numpy(), sm_conf=conf.numpy(),
         st_exp_sum=exp_sum.numpy(), st_depth_img=dimg.numpy(), st_max=mx.numpy(),
         st_depth=(dimg / fes).squeeze(1).float().numpy(), st_conf=(mx / fes).squeeze(1).float().numpy(),
         w4_conf=cas["photometric_confidence"].numpy(), ucs_variance=ucs["variance"].numpy(), ucs_lamb=np.float32(1.5))


def gen_depth_range():
    """modules/depth_range.py:4-42 + the trilinear resize of networks/casred.py:138-145."""
    B, H, W = 1, 32, 64
    torch.manual_seed(15)
    dv = torch.tensor([[10.0, 410.0]])
    s1 = ref_depth_range.get_depth_range_samples(dv, 8, 10.0, "cpu", torch.float32, [B, H, W])
    cur = torch.rand(B, H, W) * 300 + 50
    s2 = ref_depth_range.get_depth_range_samples(cur, 6, 5.0, "cpu", torch.float32, [B, H, W])
    r1 = torch.nn.functional.interpolate(s1.unsqueeze(1), [8, H // 4, W // 4], mode="trilinear", align_corners=False).squeeze(1)
    r2 = torch.nn.functional.interpolate(s2.unsqueeze(1), [6, H // 2, W // 2], mode="trilinear", align_corners=False).squeeze(1)
    # the two generated stages end to end, as networks/casred.py:134-145 runs them: previous height map -> bilinear resize to the
    # image -> samples -> trilinear resize to the stage grid (stage 2: 1/4 -> image -> 1/2; stage 3: 1/2 -> image -> 1/1)
    prev_a = torch.rand(B, H // 4, W // 4) * 300 + 50
    cur_a = torch.nn.functional.interpolate(prev_a.unsqueeze(1), [H, W], mode="bilinear", align_corners=False).squeeze(1)
    s_a = ref_depth_range.get_depth_range_samples(cur_a, 6, 2 * 2.5, "cpu", torch.float32, [B, H, W])
    r_a = torch.nn.functional.interpolate(s_a.unsqueeze(1), [6, H // 2, W // 2], mode="trilinear", align_corners=False).squeeze(1)
    prev_b = torch.rand(B, H // 2, W // 2) * 300 + 50
    cur_b = torch.nn.functional.interpolate(prev_b.unsqueeze(1), [H, W], mode="bilinear", align_corners=False).squeeze(1)
    s_b = ref_depth_range.get_depth_range_samples(cur_b, 8, 1 * 2.5, "cpu", torch.float32, [B, H, W])
    r_b = torch.nn.functional.interpolate(s_b.unsqueeze(1), [8, H, W], mode="trilinear", align_corners=False).squeeze(1)
    save("depth_range", dv=dv.numpy(), s1=s1.numpy(), cur=cur.numpy(), s2=s2.numpy(), r1=r1.numpy(), r2=r2.numpy(),
         prev_a=prev_a.numpy(), r_a=r_a.numpy(), prev_b=prev_b.numpy(), r_b=r_b.numpy())


def gen_ucs_samples():
    """UCS-Net stage 2 / 3 hypotheses as networks/ucs.py:49-58 builds them: bilinear resize of the previous depth and
    variance outputs to the stage grid, then modules/depth_range.py:45-86 (uncertainty_aware_samples).  The range is chosen so
    that both clamps fire on part of the pixels."""
    B, H, W = 2, 24, 40
    torch.manual_seed(23)
    prev = torch.rand(B, H // 2, W // 2) * 300 + 50
    var = torch.rand(B, H // 2, W // 2) * 20 + 0.5
    dmin, dmax = torch.tensor([80.0, 40.0]), torch.tensor([300.0, 360.0])
    out = {}
    for name, nd in (("s8", 8), ("s12", 12)):
        cur = torch.nn.functional.interpolate(prev.unsqueeze(1), [H, W], mode="bilinear", align_corners=False)
        ev = torch.nn.functional.interpolate(var.unsqueeze(1), [H, W], mode="bilinear", align_corners=False)
        out[name] = ref_depth_range.uncertainty_aware_samples(cur, dmin, dmax, ev, nd, "cpu", torch.float32, [B, H, W]).numpy()
    save("ucs_samples", prev=prev.numpy(), var=var.numpy(), dmin=dmin.numpy(), dmax=dmax.numpy(), **out)


def gen_cascade():
    """Full forwards at 3-view 64x128 (ndepths 16/8/8): CascadeREDNet, Infer_CascadeREDNet
    (networks/casred.py:114,285), CascadeMVSNet (networks/casmvs.py:79), UCSNet (networks/ucs.py:79).

    Weights are NOT stored (13 MB of random floats): each net is built right after
    torch.manual_seed(seed) with torch's default initialisers, and the fixture carries the seed
    plus a per-parameter (sum, sum-of-squares) checksum so the test can prove that its own module
    tree, built under the same seed, holds the identical parameters before comparing outputs.
    """
    B, V, H, W = 1, 3, 64, 128
    nd = [16, 8, 8]
    torch.manual_seed(16)
    imgs = torch.randn(B, V, 3, H, W)
    rpc_full = ref_rpcs(V, H, W, seed=51, batch=B)
    proj = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc_full, 4)),
            "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc_full, 2)),
            "stage3": torch.from_numpy(rpc_full)}
    dv = torch.tensor([[20.0, 380.0]])
    arrays = {"imgs": imgs.numpy(), "rpc": rpc_full, "dv": dv.numpy(), "ndepths": np.array(nd)}

    def record(tag, net, out):
        for s in ("stage1", "stage2", "stage3"):
            for k, v in out[s].items():
                arrays["%s.%s.%s" % (tag, s, k)] = v.numpy()
        names, sums = [], []
        for k, v in net.state_dict().items():
            if "num_batches_tracked" in k:
                continue
            names.append(k)
            sums.append([float(v.double().sum()), float((v.double() ** 2).sum())])
        arrays[tag + ".param_names"] = np.array(names)
        arrays[tag + ".param_sums"] = np.array(sums)

    def run(tag, seed, ctor):
        torch.manual_seed(seed)
        net = ctor().eval()
        arrays[tag + ".seed"] = np.int64(seed)
        with torch.no_grad():
            out = net(imgs, proj, dv)
        record(tag, net, out)
        return net

    red = run("red", 17, lambda: ref_casred.CascadeREDNet("rpc", min_interval=2.5, ndepths=nd))
    inf = ref_casred.Infer_CascadeREDNet("rpc", min_interval=2.5, ndepths=nd).eval()
    inf.load_state_dict(red.state_dict())
    with torch.no_grad():
        o = inf(imgs, proj, dv)
    for s in ("stage1", "stage2", "stage3"):
        for k, v in o[s].items():
            arrays["redinf.%s.%s" % (s, k)] = v.numpy()
    run("casmvs", 18, lambda: ref_casmvs.CascadeMVSNet("rpc", min_interval=2.5, ndepths=nd))
    run("ucs", 19, lambda: ref_ucs.UCSNet("rpc", stage_configs=nd))
    save("cascade", **arrays)


def gen_cascade_pinhole():
    """BASELINE cfg5 family: full forwards with geo_model="pinhole" (homography warp, modules/warping.py:6-44)
    at 3-view 64x128, ndepths 16/8/8: UCSNet (networks/ucs.py:79), CascadeREDNet and Infer_CascadeREDNet
    (networks/casred.py:114,285; same weights).  Stage matrices = intrinsics rows scaled by 1/4, 1/2, 1 as
    dataset/virdataset.py does.  Weights by seed + checksum as in gen_cascade."""
    B, V, H, W = 1, 3, 64, 128
    nd = [16, 8, 8]
    torch.manual_seed(31)
    imgs = torch.randn(B, V, 3, H, W)
    full = _pinhole_mats(V, H, W, seed=9, batch=B)

    def scaled(s):
        m = full.copy()
        m[:, :, :2, :] /= s
        return torch.from_numpy(m)

    proj = {"stage1": scaled(4), "stage2": scaled(2), "stage3": scaled(1)}
    dv = torch.tensor([[420.0, 680.0]])
    arrays = {"imgs": imgs.numpy(), "proj": full, "dv": dv.numpy(), "ndepths": np.array(nd)}

    def record(tag, out):
        for s in ("stage1", "stage2", "stage3"):
            for k, v in out[s].items():
                arrays["%s.%s.%s" % (tag, s, k)] = v.numpy()

    def sums(tag, net):
        arrays[tag + ".param_sums"] = np.array([[float(v.double().sum()), float((v.double() ** 2).sum())]
                                                for k, v in net.state_dict().items() if "num_batches_tracked" not in k])

    torch.manual_seed(33)
    red = ref_casred.CascadeREDNet("pinhole", min_interval=2.5, ndepths=nd).eval()
    arrays["red.seed"] = np.int64(33)
    sums("red", red)
    inf = ref_casred.Infer_CascadeREDNet("pinhole", min_interval=2.5, ndepths=nd).eval()
    inf.load_state_dict(red.state_dict())
    torch.manual_seed(34)
    ucs = ref_ucs.UCSNet("pinhole", stage_configs=nd).eval()
    arrays["ucs.seed"] = np.int64(34)
    sums("ucs", ucs)
    with torch.no_grad():
        record("red", red(imgs, proj, dv))
        record("redinf", inf(imgs, proj, dv))
        record("ucs", ucs(imgs, proj, dv))
    save("cascade_pinhole", **arrays)


def gen_costreg():
    """CostRegNet.forward (modules/module.py:546-577) in eval mode with non-trivial BatchNorm running
    statistics; weights exported (1.2 MB)."""
    torch.manual_seed(23)
    net = ref_module.CostRegNet(8, 8).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.7, 1.3)
            m.bias.data.normal_(0, 0.1)
    x = torch.randn(1, 8, 8, 16, 24)
    with torch.no_grad():
        y = net(x)
    arrays = {"w." + k: v for k, v in np_state(net).items() if "num_batches_tracked" not in k}
    save("costreg", x=x.numpy(), y=y.numpy(), **arrays)


def gen_featnet():
    """FeatureNet.forward (modules/module.py:442-543; base 8, 3 stages) in eval mode with non-trivial BatchNorm
    running statistics, two views of 40x56 (ragged against the 64-wide tiles), both arch modes ("unet" used by
    casred / ucs, "fpn" by casmvs); weights exported."""
    for arch, seed in (("unet", 29), ("fpn", 30)):
        torch.manual_seed(seed)
        net = ref_module.FeatureNet(base_channels=8, stride=4, num_stage=3, arch_mode=arch).eval()
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.7, 1.3)
                m.bias.data.normal_(0, 0.1)
        x = torch.randn(2, 3, 40, 56)
        with torch.no_grad():
            y = net(x)
        arrays = {"w." + k: v for k, v in np_state(net).items() if "num_batches_tracked" not in k}
        save("featnet" if arch == "unet" else "featnet_fpn", x=x.numpy(), s1=y["stage1"].numpy(), s2=y["stage2"].numpy(),
             s3=y["stage3"].numpy(), **arrays)


def gen_train():
    """One training step of the reference, train.py:267-302 without the optimiser: CascadeREDNet in train() mode
    (BatchNorm on batch statistics, whole-volume RED regulariser, differentiable warp: networks/casred.py:22-62, :114-158)
    -> cas_mvsnet_loss (networks/loss.py:5-25, dlossw 0.5/1/2 as in train.py's default) -> loss.backward().
    Stored: the loss, the per-stage heights, d loss / d parameter in full for FeatureNet and for conv_gru1 of every
    stage's regulariser, and a (sum, sum of squares) checksum of EVERY parameter gradient.  Inputs, seed and weights are
    those of gen_cascade (the net is rebuilt from seed 17; the fixture carries the parameter checksums)."""
    from networks.loss import cas_mvsnet_loss
    B, V, H, W = 1, 3, 64, 128
    nd = [16, 8, 8]
    torch.manual_seed(16)
    imgs = torch.randn(B, V, 3, H, W)
    rpc_full = ref_rpcs(V, H, W, seed=51, batch=B)
    proj = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc_full, 4)),
            "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc_full, 2)),
            "stage3": torch.from_numpy(rpc_full)}
    dv = torch.tensor([[20.0, 380.0]])
    gt, mask = {}, {}
    for i, s in enumerate((4, 2, 1)):
        h, w = H // s, W // s
        yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
        gt["stage%d" % (i + 1)] = (200.0 + 40.0 * torch.sin(3.0 * xx + 0.3) * torch.cos(2.0 * yy - 0.2)).unsqueeze(0)
        m = torch.ones(1, h, w)
        m[:, : h // 8, : w // 6] = 0.0
        mask["stage%d" % (i + 1)] = m
    torch.manual_seed(17)
    net = ref_casred.CascadeREDNet("rpc", min_interval=2.5, ndepths=nd).train()
    out = net(imgs, proj, dv)
    loss, depth_loss = cas_mvsnet_loss(out, gt, mask, dlossw=[0.5, 1.0, 2.0])
    loss.backward()
    arrays = {"loss": loss.detach().numpy(), "depth_loss": depth_loss.detach().numpy(), "seed": np.int64(17), "ndepths": np.array(nd),
              "dlossw": np.array([0.5, 1.0, 2.0])}
    for s in ("stage1", "stage2", "stage3"):
        arrays["gt." + s], arrays["mask." + s] = gt[s].numpy(), mask[s].numpy()
        arrays["depth." + s] = out[s]["depth"].detach().numpy()
    names, sums = [], []
    for k, p_ in net.named_parameters():
        assert p_.grad is not None, k
        g_ = p_.grad
        names.append(k)
        sums.append([float(g_.double().sum()), float((g_.double() ** 2).sum())])
        if k.startswith("feature.") or ".conv_gru1." in k:
            arrays["grad." + k] = g_.numpy()
    arrays["grad_names"], arrays["grad_sums"] = np.array(names), np.array(sums)
    save("train_step", **arrays)


def gen_train3d():
    """One training step of the reference's two cascades with the 3-D regulariser, train.py:267-302 without the optimiser:
    CascadeMVSNet (networks/casmvs.py:79-140) and UCSNet (networks/ucs.py:79-150) in train() mode (BatchNorm2d / BatchNorm3d on batch
    statistics, CostRegNet module.py:546-577 under autograd, differentiable warp) -> cas_mvsnet_loss (networks/loss.py:5-25, dlossw
    0.5/1/2) -> loss.backward().  Inputs, seeds and weights are those of gen_cascade (casmvs: seed 18, ucs: seed 19).  Stored per net:
    the loss, the per-stage heights, d loss / d parameter in full for stage 1's regulariser (cost_regularization.0.*, tensors of at
    most 20 000 entries) and for
    FeatureNet's first and last layers, a (sum, sum of squares) checksum of EVERY parameter gradient, and the BatchNorm running
    statistics after the step as checksums."""
    from networks.loss import cas_mvsnet_loss
    B, V, H, W = 1, 3, 64, 128
    nd = [16, 8, 8]
    torch.manual_seed(16)
    imgs = torch.randn(B, V, 3, H, W)
    rpc_full = ref_rpcs(V, H, W, seed=51, batch=B)
    proj = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc_full, 4)),
            "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc_full, 2)),
            "stage3": torch.from_numpy(rpc_full)}
    dv = torch.tensor([[20.0, 380.0]])
    gt, mask = {}, {}
    for i, s in enumerate((4, 2, 1)):
        h, w = H // s, W // s
        yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
        gt["stage%d" % (i + 1)] = (200.0 + 40.0 * torch.sin(3.0 * xx + 0.3) * torch.cos(2.0 * yy - 0.2)).unsqueeze(0)
        m = torch.ones(1, h, w)
        m[:, : h // 8, : w // 6] = 0.0
        mask["stage%d" % (i + 1)] = m
    arrays = {"ndepths": np.array(nd), "dlossw": np.array([0.5, 1.0, 2.0])}
    for s in ("stage1", "stage2", "stage3"):
        arrays["gt." + s], arrays["mask." + s] = gt[s].numpy(), mask[s].numpy()
    for tag, seed, ctor in (("casmvs", 18, lambda: ref_casmvs.CascadeMVSNet("rpc", min_interval=2.5, ndepths=nd)),
                            ("ucs", 19, lambda: ref_ucs.UCSNet("rpc", stage_configs=nd))):
        torch.manual_seed(seed)
        net = ctor().train()
        out = net(imgs, proj, dv)
        loss, depth_loss = cas_mvsnet_loss(out, gt, mask, dlossw=[0.5, 1.0, 2.0])
        loss.backward()
        arrays[tag + ".seed"] = np.int64(seed)
        arrays[tag + ".loss"], arrays[tag + ".depth_loss"] = loss.detach().numpy(), depth_loss.detach().numpy()
        for s in ("stage1", "stage2", "stage3"):
            arrays["%s.depth.%s" % (tag, s)] = out[s]["depth"].detach().numpy()
        names, sums = [], []
        full = [k for k, _ in net.named_parameters()]
        first_last = {full[0], full[1], full[2]} | set(k for k in full if ".out1." in k or ".out3." in k)
        for k, p_ in net.named_parameters():
            assert p_.grad is not None, k
            g_ = p_.grad
            names.append(k)
            sums.append([float(g_.double().sum()), float((g_.double() ** 2).sum())])
            if (k.startswith("cost_regularization.0.") and g_.numel() <= 20000) or k in first_last:      # (conv4..conv7: checksums only)
                arrays["%s.grad.%s" % (tag, k)] = g_.numpy()
        arrays[tag + ".grad_names"], arrays[tag + ".grad_sums"] = np.array(names), np.array(sums)
        bnames, bsums = [], []
        for k, b_ in net.named_buffers():
            if "running_" in k:
                bnames.append(k)
                bsums.append([float(b_.double().sum()), float((b_.double() ** 2).sum())])
        arrays[tag + ".buffer_names"], arrays[tag + ".buffer_sums"] = np.array(bnames), np.array(bsums)
    save("train_step3d", **arrays)


def filter_scene(H=64, W=96, V=3, seed=3):
    """V consistent height maps of one smooth surface (one per view, through OUR synthesiser: inputs only), an 8 m
    blunder patch in the last view and a confidence map with a low-confidence corner."""
    rpc = rpc_synth.make_view_rpcs(V, H, W, seed=seed)
    lat0, lon0, ls, os_ = rpc[0][2], rpc[0][3], rpc[0][7], rpc[0][8]

    def surface(lat, lon):
        u, v = (lat - lat0) / ls, (lon - lon0) / os_
        return 200.0 + 30.0 * np.sin(2.3 * u + 0.2) * np.cos(1.9 * v - 0.4)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    depths = []
    for v in range(V):
        h = np.full((H, W), 200.0)
        for _ in range(12):
            lat, lon = rpc_synth.photo2obj(rpc[v], xx.ravel(), yy.ravel(), h.ravel())
            h = surface(lat, lon).reshape(H, W)
        depths.append(h.astype(np.float32))
    depths = np.stack(depths)
    depths[V - 1, 10:20, 30:50] += 8.0
    prob = np.random.default_rng(seed).uniform(0.2, 1.0, (H, W)).astype(np.float32)
    prob[:8, :12] = 0.05
    return depths, rpc, prob


def gen_filter():
    """tools/rpc_filter.py:11-112 run AS IS: reproject_with_depth, check_geometric_consistency, filter_depth, on top of the
    reference's own cupy projector tools/rpc_tensor.py:109-165.  Two modules the image lacks are stood in for at import
    time, nothing of the reference is edited or stored: `cupy` by numpy (the eight array functions rpc_tensor.py calls),
    and `cv2` by a module whose remap() is the oracle's restatement of OpenCV's fixed-point bilinear remap -- so every
    step of the filter is pinned by this fixture EXCEPT cv2.remap itself."""
    import types
    from oracle import oracle as orc
    cupy = types.ModuleType("cupy")
    for name in ("array", "asarray", "einsum", "ones_like", "stack", "tensordot"):
        setattr(cupy, name, getattr(np, name))
    cupy.asnumpy = np.asarray
    cupy.float = float
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT = 1, 0

    def remap(src, map1, map2, interpolation=None, borderMode=None, borderValue=0):
        assert interpolation == cv2.INTER_LINEAR and borderMode == cv2.BORDER_CONSTANT
        assert map1.dtype == np.float32 and map2.dtype == np.float32
        return orc.remap_linear_const(src, map1, map2, border=float(borderValue))
    cv2.remap = remap
    sys.modules["cupy"], sys.modules["cv2"] = cupy, cv2
    np.float, np.bool = float, bool                        # numpy < 1.24 aliases the reference still uses (rpc_filter.py:21,84)
    from tools import rpc_filter as ref_filter
    depths, rpc, prob = filter_scene()
    out = {"depths": depths, "rpc": rpc, "prob": prob, "p_ratio": np.float64(1.0), "d_ratio": np.float64(2.5),
           "geo_consist_num": np.int64(2), "confidence_ratio": np.float64(0.3)}
    for v in (1, 2):
        dep, xb, yb, xs, ys = ref_filter.reproject_with_depth(depths[0].copy(), rpc[0], depths[v].copy(), rpc[v])
        m, dm, xs2, ys2 = ref_filter.check_geometric_consistency(depths[0].copy(), rpc[0], depths[v].copy(), rpc[v], 1.0, 2.5)
        assert np.array_equal(xs, xs2) and np.array_equal(ys, ys2)
        out.update({"v%d.sampled" % v: dep, "v%d.x_back" % v: xb, "v%d.y_back" % v: yb, "v%d.x_src" % v: xs, "v%d.y_src" % v: ys,
                    "v%d.mask" % v: m, "v%d.depth_masked" % v: dm})
    final, avg = ref_filter.filter_depth(depths.copy(), rpc, 1.0, 2.5, 2, prob=prob, confidence_ratio=0.3)
    final0, avg0 = ref_filter.filter_depth(depths.copy(), rpc, 1.0, 2.5, 1)
    out.update({"final_mask": final, "averaged": avg, "final_mask_noprob": final0, "averaged_noprob": avg0})
    save("filter", **out)


def pinhole_filter_scene(H=64, W=96, V=3, seed=4):
    """V consistent pinhole depth maps of one smooth surface seen from above (cameras a little apart, rotated about their axes),
    a blunder patch in the last view: inputs only, built here."""
    rng = np.random.default_rng(seed)

    def surface(X, Y):
        return 8.0 * np.sin(0.011 * X + 0.3) * np.cos(0.013 * Y - 0.2)
    Ks, Es, depths = [], [], []
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    for v in range(V):
        f = 1.15 * W
        K = np.array([[f, 0.0, W / 2.0 + 0.5 * v], [0.0, f, H / 2.0 - 0.25 * v], [0.0, 0.0, 1.0]])
        a = 0.03 * v * (-1) ** v
        R = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]]) @ np.diag([1.0, -1.0, -1.0])
        Cc = np.array([18.0 * v * (-1) ** v, 7.0 * v, 400.0 + 0.5 * v])      # (the reference compares the two views' depths as they are: same flying height)
        E = np.eye(4)
        E[:3, :3], E[:3, 3] = R, -R @ Cc
        P = np.concatenate((K @ E[:3], np.array([[0.0, 0.0, 0.0, 1.0]])), axis=0)
        Pi = np.linalg.inv(P)
        d = np.full((H, W), 400.0)
        for _ in range(20):                                    # depth along the ray such that the point lies on the surface
            Xw = Pi @ np.vstack(((d * uu).ravel(), (d * vv).ravel(), d.ravel(), np.ones(H * W)))
            Z = surface(Xw[0], Xw[1])
            d = (R @ np.vstack((Xw[0], Xw[1], Z)) + E[:3, 3:4])[2].reshape(H, W)
        Ks.append(K); Es.append(E); depths.append(d.astype(np.float32))
    depths = np.stack(depths)
    depths[V - 1, 12:22, 28:52] *= 1.03
    return depths, np.stack(Ks), np.stack(Es), rng


def gen_filter_pinhole():
    """tools/pinhole_filter.py:7-67 run AS IS (reproject_with_depth, check_geometric_consistency) on a three-view pinhole scene.
    `cv2` is stood in for at import time by a module whose remap() is the oracle's restatement of OpenCV's fixed-point bilinear
    remap (default border: constant 0) -- so every step is pinned by this fixture EXCEPT cv2.remap itself; nothing of the
    reference is edited or stored."""
    import importlib
    import types
    from oracle import oracle as orc
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT = 1, 0

    def remap(src, map1, map2, interpolation=None, borderMode=None, borderValue=0):
        assert interpolation == cv2.INTER_LINEAR and borderMode in (None, cv2.BORDER_CONSTANT)
        assert map1.dtype == np.float32 and map2.dtype == np.float32
        return orc.remap_linear_const(src, map1, map2, border=float(borderValue))
    cv2.remap = remap
    sys.modules["cv2"] = cv2
    sys.modules.pop("tools.pinhole_filter", None)
    ref = importlib.import_module("tools.pinhole_filter")
    depths, Ks, Es, _ = pinhole_filter_scene()
    out = {"depths": depths, "K": Ks, "E": Es, "p_thre": np.float64(1.0), "relative_d_thre": np.float64(0.01)}
    for v in (1, 2):
        dep, xb, yb, xs, ys = ref.reproject_with_depth(depths[0].copy(), Ks[0], Es[0], depths[v].copy(), Ks[v], Es[v])
        m, dm, xs2, ys2 = ref.check_geometric_consistency(depths[0].copy(), Ks[0], Es[0], depths[v].copy(), Ks[v], Es[v], 1.0, 0.01)
        assert np.array_equal(xs, xs2) and np.array_equal(ys, ys2)
        out.update({"v%d.sampled" % v: dep, "v%d.x_back" % v: xb, "v%d.y_back" % v: yb, "v%d.x_src" % v: xs, "v%d.y_src" % v: ys,
                    "v%d.mask" % v: m, "v%d.depth_masked" % v: dm})
        print("   pinhole filter v%d: mask mean %.3f, blunder patch kept %.3f" % (v, m.mean(), m[12:22, 28:52].mean()))
    save("filter_pinhole", **out)


def _import_ref_dataset():
    """The reference's MVSDataset class, imported unmodified behind the import-time stand-ins described in gen_dataset."""
    import types
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST = 0

    def resize(src, dsize, interpolation=None):
        assert interpolation == cv2.INTER_NEAREST
        ow, oh = dsize
        h, w = src.shape[:2]
        ys = np.minimum(np.floor(np.arange(oh) * (h / oh)).astype(np.int64), h - 1)
        xs = np.minimum(np.floor(np.arange(ow) * (w / ow)).astype(np.int64), w - 1)
        return np.ascontiguousarray(src[ys][:, xs])
    cv2.resize = resize
    cv2.setNumThreads = lambda n: None
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda f: None)
    sys.modules["cv2"] = cv2
    osgeo = types.ModuleType("osgeo")
    osgeo.gdal = types.ModuleType("osgeo.gdal")
    sys.modules.setdefault("osgeo", osgeo)
    sys.modules.setdefault("osgeo.gdal", osgeo.gdal)
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"], sys.modules["matplotlib.pyplot"] = mpl, mpl.pyplot
    from dataset.satmvsdataset import MVSDataset as RefDataset
    return RefDataset


def gen_dataset():
    """dataset/satmvsdataset.py:36-160 (MVSDataset.get_sample / get_pred_sample) + dataset/gen_list.py + read_img / center_image,
    run AS IS on a small scene folder written by OUR writers (tests/golden/scene/: 3 views x 2 tiles of 32x64 PNG -- one tile
    single-band --, .rpc, .pfm).  Modules the image lacks are stood in for at import time, nothing of the reference is edited or
    stored: osgeo.gdal / matplotlib.pyplot (imported, never called on this path) by empty modules, cv2 by a module whose
    resize(..., INTER_NEAREST) is OpenCV's nearest rule src = floor(dst * scale) in numpy -- so everything the assembler does is
    pinned by this fixture EXCEPT cv2.resize itself."""
    from PIL import Image
    from satmvs_amd import data_io
    RefDataset = _import_ref_dataset()

    scene = os.path.join(HERE, "scene")
    V, H, W = 3, 32, 64
    rng = np.random.default_rng(101)
    for t, tile in enumerate(("tile_a", "tile_b")):
        rpc = ref_rpcs(V, H, W, seed=102 + t)[0]
        for v in range(V):
            for kind in ("image", "rpc", "height"):
                os.makedirs(os.path.join(scene, kind, str(v)), exist_ok=True)
            base = rng.integers(0, 256, (H // 4, W // 4, 3)).astype(np.float32)
            img = np.clip(np.kron(base, np.ones((4, 4, 1), np.float32)) + rng.normal(0, 9, (H, W, 3)), 0, 255).astype(np.uint8)
            if tile == "tile_b":
                Image.fromarray(img[:, :, 0], "L").save(os.path.join(scene, "image", str(v), tile + ".png"))     # single-band tile
            else:
                Image.fromarray(img, "RGB").save(os.path.join(scene, "image", str(v), tile + ".png"))
            data_io.save_rpc(os.path.join(scene, "rpc", str(v), tile + ".rpc"), rpc[v])
            hm = (rpc[v][4] + rpc[v][9] * rng.uniform(-1.3, 1.3, (H, W))).astype(np.float32)                    # some heights outside the range: mask
            data_io.save_pfm(os.path.join(scene, "height", str(v), tile + ".pfm"), hm)
    out = {}
    for mode, ref_view in (("test", 2), ("test", 0), ("pred", 2)):
        ds = RefDataset(scene, mode, V, ref_view=ref_view)
        out["%s%d.len" % (mode, ref_view)] = np.int64(len(ds))
        for i in range(len(ds)):
            smp = ds[i]
            key = "%s%d.%s.%s" % (mode, ref_view, smp["out_view"], smp["out_name"])
            out[key + ".imgs"] = smp["imgs"]
            out[key + ".depth_values"] = smp["depth_values"]
            for st in ("stage1", "stage2", "stage3"):
                out[key + ".cam." + st] = smp["cam_para"][st]
                if mode != "pred":
                    out[key + ".depth." + st] = smp["depth"][st]
                    out[key + ".mask." + st] = smp["mask"][st]
    save("dataset", **out)


_QC_SCALARS = ["line_off", "samp_off", "lat_off", "lon_off", "height_off", "line_scale", "samp_scale", "lat_scale", "lon_scale", "height_scale"]
_QC_TENSORS = ["line_num", "line_den", "samp_num", "samp_den", "lat_num", "lat_den", "lon_num", "lon_den"]


def gen_dataset_qc():
    """dataset/satmvsdataset.py:166-296 (MVSDataset(use_qc=True): get_sample_qc / get_pred_sample_qc over
    dataset/data_io.py:95-150 load_rpc_as_qc_tensor / to_tensor) run AS IS on the scene folder gen_dataset wrote: per stage and view the
    ten scalars and the eight (4,4,4) tensors of the QC dictionaries, images, height range, height maps and masks."""
    RefDataset = _import_ref_dataset()
    scene = os.path.join(HERE, "scene")
    out = {}
    for mode, ref_view in (("test", 2), ("test", 0), ("pred", 2)):
        ds = RefDataset(scene, mode, 3, ref_view=ref_view, use_qc=True)
        out["%s%d.len" % (mode, ref_view)] = np.int64(len(ds))
        for i in range(len(ds)):
            smp = ds[i]
            key = "%s%d.%s.%s" % (mode, ref_view, smp["out_view"], smp["out_name"])
            out[key + ".imgs"] = smp["imgs"]
            out[key + ".depth_values"] = smp["depth_values"]
            for st in ("stage1", "stage2", "stage3"):
                cams = smp["cam_para"][st]
                assert isinstance(cams, list) and len(cams) == 3
                out[key + ".cam." + st + ".scalars"] = np.array([[c[k] for k in _QC_SCALARS] for c in cams], np.float64)
                out[key + ".cam." + st + ".tensors"] = np.stack([np.stack([c[k + "_tensor"] for k in _QC_TENSORS]) for c in cams])
                if mode != "pred":
                    out[key + ".depth." + st] = smp["depth"][st]
                    out[key + ".mask." + st] = smp["mask"][st]
    save("dataset_qc", **out)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for fn in (gen_iccv, gen_project, gen_grid_sample, gen_rpc_warp, gen_qc, gen_homo, gen_costvol, gen_pred,
               gen_regress, gen_depth_range, gen_ucs_samples, gen_cascade, gen_cascade_pinhole, gen_costreg, gen_featnet, gen_photo, gen_grad, gen_io, gen_filter, gen_filter_pinhole, gen_train, gen_train3d, gen_dataset, gen_dataset_qc):
        if only and fn.__name__[4:] not in only:
            continue
        fn()
