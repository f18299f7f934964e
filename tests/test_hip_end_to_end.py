"""GPU end-to-end parity against the reference's own forward passes (golden vectors captured by
importing the reference on CPU): the cascade networks, the plane-at-a-time pred path and the
depth-sharded pred path on one GPU.

Tolerance on the regressed height map: 1e-3 m (north_star).  The golden nets carry random seeded
weights, MIOpen convolutions on the GPU vs torch CPU convolutions differ at float32 round-off
(~1e-6 relative on the regulariser output), which the softmax-weighted mean turns into <=1e-3 m on
heights of a few hundred metres.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    torch.backends.cudnn.benchmark = False
    return torch.device("cuda:0")


def _net(tag, nd):
    from satmvs_amd.networks import casmvs, casred, ucs
    if tag == "red":
        return casred.CascadeREDNet("rpc", min_interval=2.5, ndepths=nd)
    if tag == "redinf":
        return casred.Infer_CascadeREDNet("rpc", min_interval=2.5, ndepths=nd)
    if tag == "casmvs":
        return casmvs.CascadeMVSNet("rpc", min_interval=2.5, ndepths=nd)
    return ucs.UCSNet("rpc", stage_configs=nd)


def _inputs(g, dev):
    from satmvs_amd import rpc_synth
    imgs = torch.from_numpy(g["imgs"]).to(dev)
    rpc = g["rpc"]
    proj = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev),
            "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev),
            "stage3": torch.from_numpy(rpc).to(dev)}
    return imgs, proj, torch.from_numpy(g["dv"]).to(dev)


@pytest.mark.parametrize("tag", ["red", "redinf", "casmvs", "ucs"])
def test_cascade_forward_matches_reference(dev, golden, tag, arith):
    g = golden("cascade")
    nd = [int(v) for v in g["ndepths"]]
    seed_tag = "red" if tag == "redinf" else tag
    torch.manual_seed(int(g[seed_tag + ".seed"]))
    net = _net(tag, nd)                                    # same seed + same construction order = same weights
    sd = {k: v for k, v in net.state_dict().items() if "num_batches_tracked" not in k}
    sums = np.array([[float(v.double().sum()), float((v.double() ** 2).sum())] for v in sd.values()])
    np.testing.assert_allclose(sums, g[seed_tag + ".param_sums"], rtol=1e-12, atol=1e-12)
    net = net.to(dev).eval()
    imgs, proj, dv = _inputs(g, dev)
    with torch.no_grad():
        out = net(imgs, proj, dv)
    for s in ("stage1", "stage2", "stage3"):
        want = g["%s.%s.depth" % (tag, s)]
        got = out[s]["depth"].cpu().numpy()
        assert got.shape == want.shape
        err = np.abs(got - want).max()
        assert err <= H_TOL, "%s %s: max height error %.3g m" % (tag, s, err)
        wc = g["%s.%s.photometric_confidence" % (tag, s)]
        np.testing.assert_allclose(out[s]["photometric_confidence"].cpu().numpy(), wc, rtol=1e-3, atol=1e-5)
        if tag == "ucs":                                       # lamb * std-dev of the height distribution (ucs.py:73-74)
            np.testing.assert_allclose(out[s]["variance"].cpu().numpy(), g["ucs.%s.variance" % s], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("tag", ["red", "redinf", "ucs"])
def test_cascade_forward_pinhole_matches_reference(dev, golden, tag, arith):
    """BASELINE cfg5 family: geo_model="pinhole" end to end (homography cost volume; for the RED networks the
    native plane pipeline with geo_kind 1) against the reference's outputs, heights within 1e-3."""
    from satmvs_amd.networks import casred, ucs
    g = golden("cascade_pinhole")
    nd = [int(v) for v in g["ndepths"]]
    seed_tag = "red" if tag == "redinf" else tag
    torch.manual_seed(int(g[seed_tag + ".seed"]))
    if tag == "red":
        net = casred.CascadeREDNet("pinhole", min_interval=2.5, ndepths=nd)
    elif tag == "redinf":
        net = casred.Infer_CascadeREDNet("pinhole", min_interval=2.5, ndepths=nd)
    else:
        net = ucs.UCSNet("pinhole", stage_configs=nd)
    sd = {k: v for k, v in net.state_dict().items() if "num_batches_tracked" not in k}
    sums = np.array([[float(v.double().sum()), float((v.double() ** 2).sum())] for v in sd.values()])
    np.testing.assert_allclose(sums, g[seed_tag + ".param_sums"], rtol=1e-12, atol=1e-12)
    net = net.to(dev).eval()
    imgs = torch.from_numpy(g["imgs"]).to(dev)
    full = g["proj"]

    def scaled(s):
        m = full.copy()
        m[:, :, :2, :] /= s
        return torch.from_numpy(m).to(dev)

    proj = {"stage1": scaled(4), "stage2": scaled(2), "stage3": scaled(1)}
    with torch.no_grad():
        out = net(imgs, proj, torch.from_numpy(g["dv"]).to(dev))
    for s in ("stage1", "stage2", "stage3"):
        want = g["%s.%s.depth" % (tag, s)]
        got = out[s]["depth"].cpu().numpy()
        assert got.shape == want.shape
        err = np.abs(got - want).max()
        assert err <= H_TOL, "%s %s: max depth error %.3g" % (tag, s, err)
        np.testing.assert_allclose(out[s]["photometric_confidence"].cpu().numpy(),
                                   g["%s.%s.photometric_confidence" % (tag, s)], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("tag", ["red", "redinf", "ucs"])
def test_cascade_forward_use_qc_equals_coefficient_path(dev, golden, tag):
    """use_qc=True (per-view dicts of quaternary-cubic tensors, dataset/data_io.py:123-150) through the same
    networks: the reference states the two parameterisations are equivalent (bit-identical on CPU);
    here the QC tensors are folded back to the 20 coefficients on entry, so the outputs must agree to float64
    round-off of that fold (heights within 1e-4)."""
    from satmvs_amd import rpc_synth
    g = golden("cascade")
    nd = [int(v) for v in g["ndepths"]]
    seed_tag = "red" if tag == "redinf" else tag
    keys = ["line_off", "samp_off", "lat_off", "lon_off", "height_off", "line_scale", "samp_scale", "lat_scale",
            "lon_scale", "height_scale"]
    names = ["line_num", "line_den", "samp_num", "samp_den", "lat_num", "lat_den", "lon_num", "lon_den"]

    def qc_views(rpc):                                     # (B,V,170) -> list over views of dicts
        out = []
        for v in range(rpc.shape[1]):
            r = rpc[:, v]
            d = {k: torch.from_numpy(np.ascontiguousarray(r[:, i])).to(dev) for i, k in enumerate(keys)}
            for j, nm in enumerate(names):
                d[nm + "_tensor"] = torch.from_numpy(
                    np.stack([rpc_synth.coeffs_to_qc_tensor(x[10 + 20 * j:30 + 20 * j]) for x in r])).to(dev)
            out.append(d)
        return out

    outs = []
    for use_qc in (False, True):
        torch.manual_seed(int(g[seed_tag + ".seed"]))
        from satmvs_amd.networks import casred, ucs
        if tag == "red":
            net = casred.CascadeREDNet("rpc", min_interval=2.5, ndepths=nd, use_qc=use_qc)
        elif tag == "redinf":
            net = casred.Infer_CascadeREDNet("rpc", min_interval=2.5, ndepths=nd, use_qc=use_qc)
        else:
            net = ucs.UCSNet("rpc", stage_configs=nd, use_qc=use_qc)
        net = net.to(dev).eval()
        imgs, proj, dv = _inputs(g, dev)
        if use_qc:
            rpc = g["rpc"]
            proj = {"stage1": qc_views(rpc_synth.rescale_rpc(rpc, 4)), "stage2": qc_views(rpc_synth.rescale_rpc(rpc, 2)),
                    "stage3": qc_views(rpc)}
        with torch.no_grad():
            outs.append(net(imgs, proj, dv))
    for s in ("stage1", "stage2", "stage3"):
        a, b = outs[0][s]["depth"], outs[1][s]["depth"]
        assert float((a - b).abs().max()) <= 1e-4, "%s %s" % (tag, s)


def _red_pred_setup(g, dev, cls):
    reg = cls(8, 8).eval()
    reg.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")})
    feats = [torch.from_numpy(f).to(dev) for f in g["feats"]]
    return reg.to(dev), feats, torch.from_numpy(g["rpc"]).to(dev), torch.from_numpy(g["depth"]).to(dev)


def test_red_native_step_matches_reference(dev, golden, oracle):
    """smvs_red_step_fwd (HIP convolutions, GroupNorm, GRU gating) vs the reference's slice_RED outputs
    and vs the oracle's RED step, two consecutive planes (state carried).  float32 convolution sums in a
    different order: tolerance 2e-5 absolute on values of magnitude ~3."""
    from satmvs_amd.modules.module import slice_RED_Regularization
    g = golden("red_pred")
    reg = slice_RED_Regularization(8, 8).eval()
    reg.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")})
    reg = reg.to(dev)
    x = torch.from_numpy(g["slice_x"]).to(dev)
    B, _, H, W = x.shape
    st = reg.initial_states(B, H, W, dev)
    with torch.no_grad():
        assert reg._use_native(x)
        r1 = reg(x, *st)
        out1 = r1[0].clone()
        r2 = reg(x * 0.5, *r1[1:])
    np.testing.assert_allclose(out1.cpu().numpy(), g["slice_out1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(r2[0].cpu().numpy(), g["slice_out2"], rtol=0, atol=2e-5)
    for i, k in enumerate(["slice_s1", "slice_s2", "slice_s3", "slice_s4"]):
        np.testing.assert_allclose(r2[1 + i].cpu().numpy(), g[k], rtol=0, atol=2e-5)
    wt = {k[2:]: g[k] for k in g.files if k.startswith("w.")}
    zeros = [np.zeros((B, 8, H, W), np.float32), np.zeros((B, 16, H // 2, W // 2), np.float32),
             np.zeros((B, 32, H // 4, W // 4), np.float32), np.zeros((B, 64, H // 8, W // 8), np.float32)]
    o1, s1 = oracle.red_step(wt, g["slice_x"], zeros)
    np.testing.assert_allclose(out1.cpu().numpy(), o1, rtol=0, atol=2e-5)
    # the differentiable PyTorch composite of the same module agrees too (training path)
    st2 = reg.initial_states(B, H, W, dev)
    xg = x.clone().requires_grad_(True)
    t1 = reg(xg, *st2)
    np.testing.assert_allclose(t1[0].detach().cpu().numpy(), out1.cpu().numpy(), rtol=0, atol=2e-5)


def test_costreg_native_matches_reference(dev, golden, oracle):
    """smvs_costreg_fwd (3-D convolutions, folded BatchNorm, skips) vs the reference's CostRegNet output
    (eval mode, non-trivial running statistics) and vs the oracle."""
    from satmvs_amd.modules.module import CostRegNet
    g = golden("costreg")
    net = CostRegNet(8, 8).eval()
    net.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}, strict=False)
    net = net.to(dev)
    x = torch.from_numpy(g["x"]).to(dev)
    with torch.no_grad():
        assert net._use_native(x)
        y = net(x)
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=0, atol=1e-5)
    wt = {k[2:]: g[k] for k in g.files if k.startswith("w.")}
    np.testing.assert_allclose(y.cpu().numpy(), oracle.costregnet(wt, g["x"]), rtol=0, atol=1e-5)
    os.environ["SMVS_COSTREG_TORCH"] = "1"                 # the PyTorch composite of the same module agrees
    try:
        with torch.no_grad():
            y2 = net(x)
    finally:
        del os.environ["SMVS_COSTREG_TORCH"]
    np.testing.assert_allclose(y2.cpu().numpy(), y.cpu().numpy(), rtol=0, atol=1e-5)


@pytest.mark.parametrize("arch", ["unet", "fpn"])
def test_featnet_native_matches_reference(dev, golden, oracle, arch):
    """smvs_featnet_fwd (all views in one call; 3x3/5x5/1x1 convolutions, folded BatchNorm, fused skip
    concatenation / upsample-add laterals) vs the reference's FeatureNet outputs (eval mode, non-trivial running
    statistics), vs the oracle, and vs the PyTorch composite of the same module.  Tolerance 2e-6 on values of
    magnitude <= 0.4."""
    from satmvs_amd.modules.module import FeatureNet
    g = golden("featnet" if arch == "unet" else "featnet_fpn")
    net = FeatureNet(base_channels=8, stride=4, num_stage=3, arch_mode=arch).eval()
    net.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}, strict=False)
    net = net.to(dev)
    x = torch.from_numpy(g["x"]).to(dev)
    with torch.no_grad():
        assert net._use_native(x)
        y = net(x)
        views = net.forward_views(x[None])                # (B=1, V=2, 3, H, W): the networks' entry point
    wt = {k[2:]: g[k] for k in g.files if k.startswith("w.")}
    o = oracle.featurenet(wt, g["x"], arch)
    for i, k in enumerate(["stage1", "stage2", "stage3"]):
        np.testing.assert_allclose(y[k].cpu().numpy(), g["s%d" % (i + 1)], rtol=0, atol=2e-6)
        np.testing.assert_allclose(y[k].cpu().numpy(), o[i], rtol=0, atol=2e-6)
        for v in range(2):
            assert torch.equal(views[v][k][0], y[k][v])
    os.environ["SMVS_FEATNET_TORCH"] = "1"
    try:
        with torch.no_grad():
            y2 = net(x)
    finally:
        del os.environ["SMVS_FEATNET_TORCH"]
    for k in y:
        np.testing.assert_allclose(y2[k].cpu().numpy(), y[k].cpu().numpy(), rtol=0, atol=2e-6)


def test_pred_path_matches_reference(dev, golden, arith):
    """compute_depth_when_pred (plane loop, recurrent state, streaming float64 regression)."""
    from satmvs_amd.modules.module import slice_RED_Regularization
    from satmvs_amd.networks.casred import compute_depth_when_pred
    g = golden("red_pred")
    reg, feats, rpc, dv = _red_pred_setup(g, dev, slice_RED_Regularization)
    with torch.no_grad():
        out = compute_depth_when_pred(feats, rpc, dv, dv.shape[1], reg, "rpc", False)
    assert np.abs(out["depth"].cpu().numpy() - g["pred_depth"]).max() <= H_TOL
    np.testing.assert_allclose(out["photometric_confidence"].cpu().numpy(), g["pred_conf"], rtol=1e-4, atol=1e-6)


def test_train_path_matches_reference(dev, golden, arith):
    from satmvs_amd.modules.module import RED_Regularization
    from satmvs_amd.networks.casred import compute_depth_when_train
    g = golden("red_pred")
    reg, feats, rpc, dv = _red_pred_setup(g, dev, RED_Regularization)
    with torch.no_grad():
        out = compute_depth_when_train(feats, rpc, dv, dv.shape[1], reg, "rpc", False)
    assert np.abs(out["depth"].cpu().numpy() - g["train_depth"]).max() <= H_TOL
    np.testing.assert_allclose(out["photometric_confidence"].cpu().numpy(), g["train_conf"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,D", [(2, 6), (1, 21), (2, 17)])
def test_red_volume_pipeline_equals_per_plane_steps(dev, B, D):
    """smvs_red_volume_planes (stream-pipelined plane loop, variance volume never materialised, B = 2 to cover
    the batch stride of the (B,D,H,W) output and the (plane, batch) sample order of the chunked front; 21 / 17 planes =
    chunks of 8 + a short tail, ring of two chunks reused) against RED_Regularization.forward on the materialised
    volume (one smvs_red_step_fwd per plane on one stream): same kernels, same per-plane order -> same bits."""
    from satmvs_amd import rpc_synth
    from satmvs_amd.modules.module import RED_Regularization
    from satmvs_amd.modules.warping import variance_cost_volume
    torch.manual_seed(5)
    V, C, H, W = 3, 8, 32, 40
    reg = RED_Regularization(C, 8).to(dev).eval()
    feats = [torch.randn(B, C, H, W, device=dev) for _ in range(V)]
    rpc = np.stack([rpc_synth.make_view_rpcs(V, H, W, seed=11 + b) for b in range(B)])
    proj = torch.from_numpy(rpc).to(dev)
    dv = (torch.linspace(50, 350, D, device=dev).view(1, D, 1, 1) + 3 * torch.rand(B, D, H, W, device=dev)).contiguous()
    with torch.no_grad():
        assert reg._use_native(feats[0])
        a = reg.native_volume(feats, proj, dv, "rpc", False)
        b = reg(variance_cost_volume(feats, proj, dv, "rpc", False))
    assert a.shape == b.shape == (B, D, H, W)
    assert torch.equal(a, b)


def test_single_stream_mode_graph_capture_and_shutdown(dev, golden):
    """smvs_red_set_streams(0): the plane pipeline stays on the caller's stream -- same bits as the default three-stream
    pipeline, and legal under stream capture (the whole pred loop of a stage recorded into one hipGraph and replayed).
    smvs_shutdown() releases the pooled helper streams / events; the next call re-creates them."""
    from satmvs_amd import _lib
    from satmvs_amd.modules.module import slice_RED_Regularization
    from satmvs_amd.networks.casred import compute_depth_when_pred
    g = golden("red_pred")
    reg, feats, rpc, dv = _red_pred_setup(g, dev, slice_RED_Regularization)
    lib = _lib.load()
    with torch.no_grad():
        want = compute_depth_when_pred(feats, rpc, dv, dv.shape[1], reg, "rpc", False)
        assert lib.smvs_red_set_streams(0) == 2
        try:
            got = compute_depth_when_pred(feats, rpc, dv, dv.shape[1], reg, "rpc", False)
            assert torch.equal(got["depth"], want["depth"]) and torch.equal(got["photometric_confidence"], want["photometric_confidence"])
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    cap = compute_depth_when_pred(feats, rpc, dv, dv.shape[1], reg, "rpc", False)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(cap["depth"], want["depth"])
        finally:
            assert lib.smvs_red_set_streams(2) == 0
        assert lib.smvs_shutdown() == 0
        again = compute_depth_when_pred(feats, rpc, dv, dv.shape[1], reg, "rpc", False)
    assert torch.equal(again["depth"], want["depth"])


def test_sharded_pred_equals_unsharded_on_one_gpu(dev, golden):
    """satmvs_amd.shard with no process group (world 1) is the plain pred path, bit for bit."""
    from satmvs_amd import shard
    from satmvs_amd.modules.module import slice_RED_Regularization
    from satmvs_amd.networks.casred import compute_depth_when_pred
    g = golden("red_pred")
    reg, feats, rpc, dv = _red_pred_setup(g, dev, slice_RED_Regularization)
    with torch.no_grad():
        a = compute_depth_when_pred(feats, rpc, dv, dv.shape[1], reg, "rpc", False)
        b = shard.sharded_compute_depth_when_pred(feats, rpc, dv, dv.shape[1], reg, "rpc", False)
    assert torch.equal(a["depth"], b["depth"]) and torch.equal(a["photometric_confidence"], b["photometric_confidence"])


@pytest.mark.parametrize("shape,act,sliced", [
    ((2, 8, 37, 50), "sigmoid", True),          # gate half of a 16-channel tensor, batch 2, odd plane size (scalar path)
    ((1, 16, 48, 96), "tanh", False),           # vector path
    ((1, 8, 192, 384), "sigmoid", True),        # several segments per row
    ((3, 64, 12, 24), None, False),             # coarsest level: many channels, tiny planes
    ((1, 8, 384, 768), "sigmoid", True),        # the real tile: level 1 of cascade stage 3 (gate tensor, sliced)
    ((1, 8, 384, 768), "tanh", False),
])
def test_groupnorm1_native_matches_torch(dev, shape, act, sliced):
    """smvs_groupnorm1_fwd / _bwd (the training path's GroupNorm(1, C) + gate activation, csrc/groupnorm.hip) against
    torch's group_norm + sigmoid / tanh under autograd: outputs 1e-6, input / weight / bias gradients 2e-5 of their
    scale (float64 statistics here, float32 there)."""
    from satmvs_amd.modules.module import GroupNorm1
    B, C, H, W = shape
    torch.manual_seed(3)
    full = (torch.randn((B, 2 * C if sliced else C, H, W), device=dev) * 1.7 + 0.3)
    gn = GroupNorm1(1, C, 1e-5, True).to(dev)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
    gout = torch.randn((B, C, H, W), device=dev)
    res = []
    for native in (True, False):
        src = full.clone().requires_grad_(True)
        x = src[:, C:] if sliced else src
        gn.zero_grad()
        if native:
            y = gn(x, act)
        else:
            y = torch.nn.functional.group_norm(x, 1, gn.weight, gn.bias, 1e-5)
            y = torch.sigmoid(y) if act == "sigmoid" else torch.tanh(y) if act == "tanh" else y
        y.backward(gout)
        res.append((y.detach(), src.grad.clone(), gn.weight.grad.clone(), gn.bias.grad.clone()))
    (y0, dx0, dw0, db0), (y1, dx1, dw1, db1) = res
    assert float((y0 - y1).abs().max()) <= 2e-6
    for a, b in ((dx0, dx1), (dw0, dw1), (db0, db1)):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7
    if sliced:
        assert float(dx0[:, :C].abs().max()) == 0.0             # the other half of the gate tensor gets no gradient from this norm


@pytest.mark.parametrize("B,C,H,W", [(1, 8, 33, 50), (2, 16, 24, 48), (1, 8, 384, 768)])        # scalar path, batch 2, the real tile
def test_groupnorm1_fused_epilogues_equal_the_separate_launches(dev, B, C, H, W):
    """smvs_groupnorm1_pair_fwd_mul / smvs_groupnorm1_fwd_blend (the ConvGRU cell's r*h and u-blend folded into the norms' apply pass)
    give the bits of smvs_groupnorm1_pair_fwd + a product / smvs_groupnorm1_fwd + smvs_gru_blend_fwd."""
    from satmvs_amd import _lib
    torch.manual_seed(5)
    HW = H * W
    st = _lib.current_stream(dev)
    gates = torch.randn(B, 2 * C, H, W, device=dev) * 1.3 + 0.2
    h = torch.randn(B, C, H, W, device=dev)
    p = [torch.rand(C, device=dev) + 0.5 for _ in range(6)]
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    ws = lambda n: torch.empty(max(64, n), dtype=torch.float64, device=dev)
    nblk = (C * HW + 4095) // 4096
    ru0, ru1, rh, s0, s1 = e(B, 2 * C, H, W), e(B, 2 * C, H, W), e(B, C, H, W), e(2 * B, 2), e(2 * B, 2)
    _lib.call("smvs_groupnorm1_pair_fwd", _lib.ptr(gates), _lib.ptr(p[0]), _lib.ptr(p[1]), _lib.ptr(p[2]), _lib.ptr(p[3]), 1e-5, 1, _lib.ptr(ru0),
              _lib.ptr(s0), _lib.ptr(ws(4 * B * nblk)), B, C, HW, st)
    _lib.call("smvs_groupnorm1_pair_fwd_mul", _lib.ptr(gates), _lib.ptr(p[0]), _lib.ptr(p[1]), _lib.ptr(p[2]), _lib.ptr(p[3]), 1e-5, 1, _lib.ptr(ru1),
              _lib.ptr(s1), _lib.ptr(ws(4 * B * nblk)), _lib.ptr(h), _lib.ptr(rh), B, C, HW, st)
    assert torch.equal(ru0, ru1) and torch.equal(s0, s1)
    assert torch.equal(rh, ru0[:, :C] * h)
    craw = torch.randn(B, C, H, W, device=dev)
    c0, c1, o0, o1, t0, t1 = e(B, C, H, W), e(B, C, H, W), e(B, C, H, W), e(B, C, H, W), e(B, 2), e(B, 2)
    u = ru0[:, C:]
    _lib.call("smvs_groupnorm1_fwd", _lib.ptr(craw), C * HW, _lib.ptr(p[4]), _lib.ptr(p[5]), 1e-5, 2, _lib.ptr(c0), _lib.ptr(t0),
              _lib.ptr(ws(2 * B * nblk)), B, C, HW, st)
    _lib.call("smvs_gru_blend_fwd", _lib.ptr(u.contiguous()), _lib.ptr(h), _lib.ptr(c0), _lib.ptr(o0), h.numel(), st)
    _lib.call("smvs_groupnorm1_fwd_blend", _lib.ptr(craw), C * HW, _lib.ptr(p[4]), _lib.ptr(p[5]), 1e-5, 2, _lib.ptr(c1), _lib.ptr(t1),
              _lib.ptr(ws(2 * B * nblk)), _lib.ptr(u), 2 * C * HW, _lib.ptr(h), _lib.ptr(o1), B, C, HW, st)
    assert torch.equal(c0, c1) and torch.equal(t0, t1) and torch.equal(o0, o1)


@pytest.mark.parametrize("B,cin,ch,H,W", [(1, 8, 8, 33, 50), (2, 16, 16, 24, 48), (1, 64, 64, 12, 24), (2, 32, 8, 7, 9),
                                           (1, 8, 8, 384, 768)])      # the real tile: level 1 of cascade stage 3
def test_convgru_cell_native_elementwise_matches_torch(dev, B, cin, ch, H, W):
    """ConvGRUCell2 on the GPU (native GroupNorm + activation, cat(x, r*h) and the u-blend as one launch each way) against the
    reference's operator sequence (module.py:22-58) written with torch operators on the same parameters: output 2e-6,
    gradients w.r.t. x, h and every parameter 5e-5 of their scale."""
    import torch.nn.functional as F
    from satmvs_amd.modules.module import ConvGRUCell2
    torch.manual_seed(11)
    cell = ConvGRUCell2(cin, ch, 3).to(dev)
    x0, h0 = torch.randn(B, cin, H, W, device=dev), torch.randn(B, ch, H, W, device=dev)
    gout = torch.randn(B, ch, H, W, device=dev)

    def reference(x, h):
        f = cell.gate_conv(torch.cat((x, h), 1))
        r, u = torch.split(f, ch, 1)
        r = torch.sigmoid(F.group_norm(r, 1, cell.reset_gate_norm.weight, cell.reset_gate_norm.bias, 1e-5))
        u = torch.sigmoid(F.group_norm(u, 1, cell.update_gate_norm.weight, cell.update_gate_norm.bias, 1e-5))
        o = cell.output_conv(torch.cat((x, r * h), 1))
        y = torch.tanh(F.group_norm(o, 1, cell.output_norm.weight, cell.output_norm.bias, 1e-5))
        return u * h + (1 - u) * y

    res = []
    for fn in (lambda a, b: cell(a, b)[0], reference):
        x, h = x0.clone().requires_grad_(True), h0.clone().requires_grad_(True)
        cell.zero_grad()
        out = fn(x, h)
        out.backward(gout)
        res.append([out.detach(), x.grad.clone(), h.grad.clone()] + [p.grad.clone() for p in cell.parameters()])
    assert float((res[0][0] - res[1][0]).abs().max()) <= 2e-6
    for a, b in zip(res[0][1:], res[1][1:]):
        assert float((a - b).abs().max()) <= 5e-5 * float(b.abs().max()) + 1e-7


def test_training_forward_switches_miopen_find_off(dev, golden):
    """The reference's train.py:21 sets cudnn.benchmark = True; the RED training forward switches it off with a warning
    (satmvs_amd.modules.module.guard_miopen_find: MIOpen's search faults in this network's training forward on this image, with or
    without this library in the process -- profiles/r04_miopen_find_repro.txt).  Inference forwards leave the flag alone, and the
    first one after a guarded training forward restores it."""
    import warnings
    from satmvs_amd.modules import module
    from satmvs_amd.networks import casred
    g = golden("cascade")
    nd = [int(v) for v in g["ndepths"]]
    imgs, proj, dv = _inputs(g, dev)
    saved = torch.backends.cudnn.benchmark
    try:
        torch.manual_seed(0)
        net = casred.CascadeREDNet("rpc", min_interval=2.5, ndepths=nd).to(dev)
        torch.backends.cudnn.benchmark = True
        with torch.no_grad():
            net.eval()(imgs, proj, dv)
        assert torch.backends.cudnn.benchmark is True
        module.SW.find_warned = False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out = net.train()(imgs, proj, dv)
        assert torch.backends.cudnn.benchmark is False
        assert any("cudnn.benchmark switched off" in str(x.message) for x in w)
        assert torch.isfinite(out["stage3"]["depth"]).all()
        with torch.no_grad():                                   # the next inference forward hands the caller's setting back
            net.eval()(imgs, proj, dv)
        assert torch.backends.cudnn.benchmark is True
    finally:
        module.SW.find_switched_off = False
        torch.backends.cudnn.benchmark = saved


def test_training_step_runs_and_gradients_flow(dev, golden):
    """One optimisation-free training step through the native volume: loss.backward() reaches the
    feature extractor through smvs_costvol_bwd (train.py:279-285 analogue)."""
    from satmvs_amd.networks import casred
    g = golden("cascade")
    nd = [int(v) for v in g["ndepths"]]
    torch.manual_seed(0)
    net = casred.CascadeREDNet("rpc", min_interval=2.5, ndepths=nd).to(dev).train()
    imgs, proj, dv = _inputs(g, dev)
    out = net(imgs, proj, dv)
    loss = sum(out[s]["depth"].mean() for s in ("stage1", "stage2", "stage3"))
    loss.backward()
    gnorm = sum(float(p.grad.abs().sum()) for p in net.feature.parameters() if p.grad is not None)
    assert np.isfinite(gnorm) and gnorm > 0


def test_training_step_matches_reference(dev, golden, arith):
    """One training step against the reference's own (train.py:267-302 without the optimiser; fixture
    tests/golden/train_step.npz = CascadeREDNet.train() -> cas_mvsnet_loss -> backward, run on the CPU by
    gen_golden.py::gen_train): same seed => same weights, native cost-volume forward AND backward under the PyTorch
    composites of FeatureNet / RED.  Loss within 1e-5 relative; every stored gradient (all of FeatureNet, conv_gru1 of
    each stage's regulariser) within 2e-4 of its largest entry; the (sum, sum of squares) checksums of ALL 183
    parameter gradients within 1e-3 relative -- float32 convolutions in another order, atomics in the scatter."""
    import torch.nn.functional as F
    from satmvs_amd.networks import casred
    g, gc = golden("train_step"), golden("cascade")
    nd = [int(v) for v in g["ndepths"]]
    torch.manual_seed(int(g["seed"]))
    net = casred.CascadeREDNet("rpc", min_interval=2.5, ndepths=nd)
    sd = {k: v for k, v in net.state_dict().items() if "num_batches_tracked" not in k}
    sums = np.array([[float(v.double().sum()), float((v.double() ** 2).sum())] for v in sd.values()])
    np.testing.assert_allclose(sums, gc["red.param_sums"], rtol=1e-12, atol=1e-12)
    net = net.to(dev).train()
    imgs, proj, dv = _inputs(gc, dev)
    out = net(imgs, proj, dv)
    # cas_mvsnet_loss, networks/loss.py:5-25: smooth-L1 over the masked pixels of every stage, weighted
    loss = torch.zeros((), device=dev)
    for i, s in enumerate(("stage1", "stage2", "stage3")):
        m = torch.from_numpy(g["mask." + s]).to(dev) > 0.5
        loss = loss + float(g["dlossw"][i]) * F.smooth_l1_loss(out[s]["depth"][m], torch.from_numpy(g["gt." + s]).to(dev)[m],
                                                               reduction="mean")
        assert np.abs(out[s]["depth"].detach().cpu().numpy() - g["depth." + s]).max() <= H_TOL, s
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-5)
    grads = {k: p.grad for k, p in net.named_parameters()}
    assert list(grads) == [str(n) for n in g["grad_names"]]
    worst = 0.0
    for k in g.files:
        if not k.startswith("grad."):
            continue
        want, got = g[k], grads[k[5:]].detach().cpu().numpy()
        scale = float(np.abs(want).max())
        err = float(np.abs(got - want).max()) / max(scale, 1e-12)
        worst = max(worst, err)
        assert err <= 2e-4, "%s: %.3g of its largest entry" % (k, err)
    # every parameter: gradient norm within 1e-3 (the bias of the last layer has a mathematically ZERO gradient -- softmax
    # over planes is shift-invariant -- so pure round-off there is measured against the largest norm instead)
    nmax = float(np.sqrt(g["grad_sums"][:, 1].max()))
    for (name, (s1, s2)) in zip(g["grad_names"], g["grad_sums"]):
        n = float(grads[str(name)].double().norm())
        assert abs(n - np.sqrt(s2)) <= 1e-3 * np.sqrt(s2) + 1e-6 * nmax, name


def test_graphed_training_step_matches_eager(dev, golden):
    """satmvs_amd.train_graph.GraphedTrainStep: the whole step (forward, loss, backward, RMSprop) captured in one HIP graph.
    Same seed, same sample: the first replay's loss equals the eager step's to 1e-6 and its gradients to 2e-4 of their
    scale (atomics), the warm-up inside the capture leaves parameters / BatchNorm buffers / optimizer state untouched, a
    second replay trains on (different loss), and a new sample is picked up through the static buffers."""
    import torch.nn.functional as F
    from satmvs_amd.networks import casred
    from satmvs_amd.train_graph import GraphedTrainStep
    g = golden("cascade")
    nd = [int(v) for v in g["ndepths"]]
    imgs, proj, dv = _inputs(g, dev)
    gts = {s: torch.full_like(torch.from_numpy(g["red." + s + ".depth"]).to(dev), 30.0) for s in ("stage1", "stage2", "stage3")}

    def loss_fn(out, gt):
        return sum(w * F.smooth_l1_loss(out[s]["depth"], gt[s], reduction="mean") for s, w in (("stage1", 0.5), ("stage2", 1.0), ("stage3", 2.0)))

    def make():
        torch.manual_seed(0)
        net = casred.CascadeREDNet("rpc", min_interval=2.5, ndepths=nd).to(dev).train()
        return net, torch.optim.RMSprop(net.parameters(), lr=1e-3, alpha=0.9, capturable=True)

    net_e, opt_e = make()
    opt_e.zero_grad(set_to_none=True)
    loss_e = loss_fn(net_e(imgs, proj, dv), gts)
    loss_e.backward()
    grads_e = {k: p.grad.clone() for k, p in net_e.named_parameters()}
    opt_e.step()
    opt_e.zero_grad(set_to_none=True)
    loss_e2 = float(loss_fn(net_e(imgs, proj, dv), gts).detach())      # the eager run's SECOND step: the optimizer state has to match too

    net_g, opt_g = make()
    before = {k: v.clone() for k, v in net_g.state_dict().items()}
    step = GraphedTrainStep(net_g, opt_g, loss_fn)
    step._capture((imgs, proj, dv, gts))
    for k, v in net_g.state_dict().items():
        assert torch.equal(v, before[k]), "the capture's warm-up changed %s" % k
    step._sig = None
    loss_g, out_g = step(imgs, proj, dv, gts)
    loss_g, loss_e = float(loss_g), float(loss_e.detach())      # the returned tensors are the graph's: read before the next call
    assert abs(loss_g - loss_e) <= 1e-6 * abs(loss_e)
    for k, p in net_g.named_parameters():
        scale = float(grads_e[k].abs().max())
        assert float((p.grad - grads_e[k]).abs().max()) <= 2e-4 * scale + 1e-9, k
    assert set(out_g) >= {"stage1", "stage2", "stage3"}
    loss2, _ = step(imgs, proj, dv, gts)                         # second replay: the parameters moved
    loss2 = float(loss2)
    assert loss2 != loss_g and np.isfinite(loss2)
    # RMSprop's first update is lr * g / (sqrt(0.1) |g|): parameters whose gradient is round-off noise move by +-lr/0.32 in either
    # run, with no first-order effect on the loss
    assert abs(loss2 - loss_e2) <= 1e-3 * abs(loss_e2), (loss2, loss_e2)
    gts2 = {s: t + 5.0 for s, t in gts.items()}                  # another sample, same shapes: no re-capture, other loss
    graph = step._graph
    loss3, _ = step(imgs, proj, dv, gts2)
    assert step._graph is graph and abs(float(loss3) - loss2) > 1.0
    # a scheduler stepping the rate (the reference's MultiStepLR, train.py:287): the change is noticed and the step re-captured
    n = step.captures
    before = {k: p.detach().clone() for k, p in net_g.named_parameters()}
    for grp in opt_g.param_groups:
        grp["lr"] = 0.0
    step(imgs, proj, dv, gts2)
    assert step.captures == n + 1 and step._graph is not graph
    for k, p in net_g.named_parameters():
        assert torch.equal(p, before[k]), "a step at lr = 0 moved %s: the old rate was replayed" % k
    step(imgs, proj, dv, gts2)
    assert step.captures == n + 1                               # unchanged hyper-parameters: plain replay
    # a replay steps the parameters on the device without touching autograd's version counters: the kernel-layout copies of the
    # inference path must not outlive them (modules.module.bump_param_epoch) -- eval through the native kernels == eval through the
    # PyTorch composite on the CURRENT parameters, before and after another (lr > 0) replay
    for grp in opt_g.param_groups:
        grp["lr"] = 5e-2
    step(imgs, proj, dv, gts2)                                   # (re-captured for the new rate: everything below is plain replay)
    n = step.captures
    heights = []
    for rep in range(2):
        net_g.eval()
        with torch.no_grad():
            nat = net_g(imgs, proj, dv)["stage3"]["depth"].clone()
            os.environ["SMVS_RED_TORCH"] = "1"
            try:
                comp = net_g(imgs, proj, dv)["stage3"]["depth"].clone()
            finally:
                del os.environ["SMVS_RED_TORCH"]
        assert float((nat - comp).abs().max()) <= 2e-3, (rep, float((nat - comp).abs().max()))
        heights.append(nat)
        net_g.train()
        step(imgs, proj, dv, gts2)
    assert float((heights[0] - heights[1]).abs().max()) > 2e-2   # the step in between did move the network
    assert step.captures == n
    # a per-iteration schedule of a python-float rate re-captures on every call: after three in a row the step says so (once) and
    # points at a tensor lr; the re-captures themselves keep training correctly (one warm-up step each, state restored)
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for i in range(4):
            for grp in opt_g.param_groups:
                grp["lr"] = 1e-3 * (1.0 + 0.1 * i)
            loss_i, _ = step(imgs, proj, dv, gts2)
            assert np.isfinite(float(loss_i))
    assert step.captures == n + 4
    assert sum("re-captured" in str(x.message) for x in w) == 1


def test_native_modules_match_composites_at_ragged_shapes(dev):
    """FeatureNet / CostRegNet / RED native kernels vs the PyTorch composites of the same modules at sizes that are
    ragged against every tile (64-wide lanes, 32-wide MFMA tiles, 4-row workgroups), with batch > 1: tile-edge
    and halo handling.  float32 round-off only (2e-5 on values of magnitude <= ~5)."""
    from satmvs_amd.modules.module import CostRegNet, FeatureNet, slice_RED_Regularization

    def rand_bn(net):
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.7, 1.3); m.bias.data.normal_(0, 0.1)

    def composite(env, fn):
        os.environ[env] = "1"
        try:
            return fn()
        finally:
            del os.environ[env]

    torch.manual_seed(3)
    with torch.no_grad():
        for arch in ("unet", "fpn"):
            net = FeatureNet(8, 3, 4, arch).to(dev).eval()
            rand_bn(net)
            for shape in ((3, 3, 72, 136), (1, 3, 260, 68)):
                x = torch.randn(*shape, device=dev)
                a, b = net(x), composite("SMVS_FEATNET_TORCH", lambda: net(x))
                for k in a:
                    assert float((a[k] - b[k]).abs().max()) <= 2e-5, (arch, shape, k)
        for c in (8, 32):
            net = CostRegNet(c, 8).to(dev).eval()
            rand_bn(net)
            for shape in ((2, c, 8, 24, 72), (1, c, 24, 16, 200)):
                x = torch.randn(*shape, device=dev)
                a, b = net(x), composite("SMVS_COSTREG_TORCH", lambda: net(x))
                assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), (c, shape)
        for c in (8, 32):
            net = slice_RED_Regularization(c, 8).to(dev).eval()
            for (bsz, h, w) in ((2, 24, 72), (1, 40, 136), (1, 264, 520)):
                x = torch.randn(bsz, c, h, w, device=dev)

                def two_planes():
                    st = net.initial_states(bsz, h, w, dev)
                    r = net(x, *st)
                    return net(x * 0.7, *r[1:])

                a, b = two_planes(), composite("SMVS_RED_TORCH", two_planes)
                for p_, q_ in zip(a, b):
                    assert float((p_ - q_).abs().max()) <= 2e-5, (c, bsz, h, w)


def test_native_paths_run_on_dataparallel_replicas(dev, golden, arith):
    """torch.nn.parallel.replicate (what nn.DataParallel does for every forward with >= 2 devices; here both
    replicas on cuda:0) leaves modules whose named_parameters() is empty: the native RED / CostRegNet / FeatureNet
    paths must still find their weights and reproduce the parent's outputs bit for bit."""
    from satmvs_amd.networks import casmvs, casred
    g = golden("cascade")
    nd = [int(v) for v in g["ndepths"]]
    imgs, proj, dv = _inputs(g, dev)
    for make in (lambda: casred.Infer_CascadeREDNet("rpc", min_interval=2.5, ndepths=nd),
                 lambda: casmvs.CascadeMVSNet("rpc", min_interval=2.5, ndepths=nd)):
        torch.manual_seed(5)
        net = make().to(dev).eval()
        with torch.no_grad():
            want = net(imgs, proj, dv)
            reps = torch.nn.parallel.replicate(net, [0, 0], detach=True)
            assert len(dict(reps[1].named_parameters())) == 0
            got = reps[1](imgs, proj, dv)
        for s in ("stage1", "stage2", "stage3"):
            assert torch.equal(got[s]["depth"], want[s]["depth"]), s
            assert torch.equal(got[s]["photometric_confidence"], want[s]["photometric_confidence"]), s
        # a model that carries its OWN arithmetic (arith=, round 5) keeps it on a replica that runs on another thread -- what
        # nn.DataParallel does -- whatever the process default (the `arith` fixture) is
        import threading
        other = "fused" if arith == "exact" else "exact"
        net.arith = other
        box = {}
        with torch.no_grad():
            want_o = net(imgs, proj, dv)
            reps = torch.nn.parallel.replicate(net, [0, 0], detach=True)

            def run():
                torch.cuda.set_device(dev)
                with torch.no_grad():
                    box["out"] = reps[1](imgs, proj, dv)
                torch.cuda.synchronize()
            t = threading.Thread(target=run)
            t.start(); t.join()
        assert reps[1].arith == other
        for s in ("stage1", "stage2", "stage3"):
            assert torch.equal(box["out"][s]["depth"], want_o[s]["depth"]), s
        assert not torch.equal(want_o["stage3"]["depth"], want["stage3"]["depth"])
        net.arith = None


def test_scene_folder_to_height_map(dev):
    """A WHU-TLC-shaped scene folder (tests/golden/scene/: PNG views, .rpc, .pfm) through satmvs_amd.dataset.MVSDataset into the
    inference cascade: what predict.py does with the reference's loader (predict.py:60-110), end to end from this repository
    alone.  Random weights: the check is plumbing (shapes, dtypes, keys, finite heights inside the tile's height range)."""
    from satmvs_amd.dataset import MVSDataset
    from satmvs_amd.networks import casred
    scene = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene")
    ds = MVSDataset(scene, "pred", 3)
    smp = ds[0]
    imgs = torch.from_numpy(smp["imgs"])[None].to(dev)
    proj = {k: torch.from_numpy(v)[None].to(dev) for k, v in smp["cam_para"].items()}
    dv = torch.from_numpy(smp["depth_values"])[None].to(dev)
    torch.manual_seed(3)
    net = casred.Infer_CascadeREDNet("rpc", min_interval=2.5, ndepths=[16, 8, 8]).to(dev).eval()
    with torch.no_grad():
        out = net(imgs, proj, dv)
    h = out["stage3"]["depth"]
    assert h.shape == (1, 32, 64) and torch.isfinite(h).all()
    lo, hi = float(dv[0, 0]), float(dv[0, 1])
    assert lo - 50.0 <= float(h.min()) and float(h.max()) <= hi + 50.0


@pytest.mark.parametrize("B,cin,cout,H,W", [(1, 24, 16, 96, 192), (2, 16, 8, 33, 70), (1, 5, 3, 8, 24), (1, 128, 64, 12, 24), (1, 32, 32, 48, 96), (1, 16, 16, 192, 384)])
def test_conv3x3_native_weight_gradient_matches_torch(dev, B, cin, cout, H, W):
    """smvs_conv3x3_wgrad (csrc/conv_wgrad.hip) behind satmvs_amd.modules.module._conv3x3: weight and bias gradient of the ConvGRU cells'
    3x3 convolutions against torch autograd of the same nn.Conv2d -- channel counts that are odd / not multiples of 8, widths that are
    not multiples of 64, batch 2, the coarse 12x24 level and a full stage-2 plane.  Another summation order (wave-private sums over
    rows, lane reduction, float atomics over row chunks): 2e-4 of the gradient's scale; forward and input gradient are torch's own."""
    from satmvs_amd.modules import module as M
    torch.manual_seed(B * 1000 + cin * 10 + cout)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    x = torch.randn(B, cin, H, W, device=dev, requires_grad=True)
    gy = torch.randn(B, cout, H, W, device=dev)
    y0 = conv(x)
    y0.backward(gy)
    want = (conv.weight.grad.clone(), conv.bias.grad.clone(), x.grad.clone())
    conv.zero_grad(); x.grad = None
    y1 = M._conv3x3(conv, x)
    assert y1.grad_fn is not None and "Conv3x3Wgrad" in type(y1.grad_fn).__name__
    # the forward is torch's own convolution: the same bits -- unless MIOpen itself answers two identical calls with different solvers
    # (seen on one box for the 128 -> 64 channel 12x24 level: a first call and a later call of the SAME nn.Conv2d differed in the last bits)
    if not torch.equal(y1, y0):
        with torch.no_grad():
            again = conv(x)
        assert torch.equal(y1, again) or float((y1 - y0).abs().max()) <= 1e-5 * float(y0.abs().max())
    y1.backward(gy)
    for got, ref, name in ((conv.weight.grad, want[0], "weight"), (conv.bias.grad, want[1], "bias")):
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 2e-4 * scale, (name, float((got - ref).abs().max()), scale)
    assert torch.allclose(x.grad, want[2], rtol=1e-4, atol=1e-4 * float(want[2].abs().max()))


@pytest.mark.parametrize("B,ca,cb,cout,H,W", [(1, 32, 8, 16, 96, 192), (1, 32, 8, 8, 96, 192), (2, 16, 16, 32, 24, 48), (1, 64, 64, 128, 12, 24),
                                               (1, 32, 32, 32, 48, 96), (1, 8, 8, 16, 384, 768), (1, 24, 0, 8, 33, 70), (3, 6, 5, 3, 9, 21)])
def test_conv3x3_cat_native_matches_torch(dev, B, ca, cb, cout, H, W):
    """smvs_conv3x3_fwd / smvs_conv3x3_pack / smvs_conv3x3_wgrad_cat behind modules.module._conv3x3_cat: the ConvGRU cells' convolution over
    cat(x, h) without the concatenated tensor -- forward, both input gradients, weight and bias gradient against torch autograd of
    conv2d(cat(x, h)): direct kernels (16 / 8 outputs), MFMA kernels (32 / 128 outputs; 64-channel adjoint), the 4-rows-per-lane variant at
    the real tile, a single operand, batch 2 / 3, odd sizes (torch fallback where the first operand's channel count is odd).  2e-5 of
    each tensor's scale (float32 sums in another order)."""
    from satmvs_amd.modules import module as M
    torch.manual_seed(B * 1000 + ca * 10 + cout)
    conv = torch.nn.Conv2d(ca + cb, cout, 3, padding=1).to(dev)
    xa = torch.randn(B, ca, H, W, device=dev, requires_grad=True)
    xb = torch.randn(B, cb, H, W, device=dev, requires_grad=True) if cb else None
    x_all = [xa] + ([xb] if cb else [])
    y0 = conv(torch.cat(x_all, 1))
    gy = torch.randn_like(y0)
    y0.backward(gy)
    want = [conv.weight.grad.clone(), conv.bias.grad.clone()] + [t.grad.clone() for t in x_all]
    conv.zero_grad()
    for t in x_all:
        t.grad = None
    y1 = M._conv3x3_cat(conv, xa, xb)
    native = ca % 2 == 0 or cb == 0
    assert ("Conv3x3Native" in type(y1.grad_fn).__name__) == native
    assert float((y1 - y0).abs().max()) <= 2e-5 * float(y0.abs().max())
    y1.backward(gy)
    got = [conv.weight.grad, conv.bias.grad] + [t.grad for t in x_all]
    for g, r, name in zip(got, want, ("weight", "bias", "xa", "xb")):
        scale = float(r.abs().max())
        assert g.shape == r.shape
        assert float((g - r).abs().max()) <= (2e-4 if name in ("weight", "bias") else 2e-5) * scale, (name, float((g - r).abs().max()), scale)
    # a changed weight (optimizer step) is repacked
    with torch.no_grad():
        conv.weight.mul_(0.5)
    y2 = M._conv3x3_cat(conv, xa, xb)
    assert float((y2 - conv(torch.cat(x_all, 1))).abs().max()) <= 2e-5 * float(y0.abs().max())


@pytest.mark.parametrize("kind,B,cin,cout,H,W,bias", [
    ("conv_s2", 1, 32, 16, 96, 192, False), ("conv_s2", 2, 16, 32, 24, 48, False), ("conv_s2", 1, 32, 64, 24, 48, False), ("conv_s2", 1, 8, 16, 384, 768, False),
    ("convT_s2", 1, 16, 8, 48, 96, False), ("convT_s2", 2, 64, 32, 12, 24, False), ("convT_s2", 1, 32, 16, 96, 192, False),
    ("convT_s1", 1, 8, 1, 96, 192, True), ("convT_s1", 2, 8, 1, 16, 72, True), ("conv_s1", 1, 24, 8, 40, 64, True)])
def test_regulariser_layer_native_matches_torch(dev, kind, B, cin, cout, H, W, bias):
    """modules.module._conv3x3_cat(conv, x, None, relu) for the regulariser's encoder / decoder layers: smvs_conv3x3_fwd kinds 1 (stride-2
    correlation) and 2 (stride-2 transposed convolution) and the flipped-tap correlation of the transposed output layer, each with its
    adjoint as the input gradient, the fused ReLU and its mask in the backward, and the weight / bias gradient -- against torch autograd of
    relu(layer(x)).  2e-5 of each tensor's scale."""
    from satmvs_amd.modules import module as M
    torch.manual_seed(B * 1000 + cin * 10 + cout)
    if kind == "conv_s2":
        conv = torch.nn.Conv2d(cin, cout, 3, stride=2, padding=1, bias=bias).to(dev)
    elif kind == "conv_s1":
        conv = torch.nn.Conv2d(cin, cout, 3, stride=1, padding=1, bias=bias).to(dev)
    elif kind == "convT_s2":
        conv = torch.nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=bias).to(dev)
    else:
        conv = torch.nn.ConvTranspose2d(cin, cout, 3, stride=1, padding=1, output_padding=0, bias=bias).to(dev)
    relu = kind != "convT_s1"
    x = torch.randn(B, cin, H, W, device=dev, requires_grad=True)
    y0 = conv(x)
    y0 = torch.relu(y0) if relu else y0
    gy = torch.randn_like(y0)
    y0.backward(gy)
    want = [conv.weight.grad.clone(), x.grad.clone()] + ([conv.bias.grad.clone()] if bias else [])
    conv.zero_grad(); x.grad = None
    y1 = M._conv3x3_cat(conv, x, None, relu)
    assert "Conv3x3Native" in type(y1.grad_fn).__name__
    assert y1.shape == y0.shape and float((y1 - y0).abs().max()) <= 2e-5 * float(y0.abs().max())
    y1.backward(gy)
    got = [conv.weight.grad, x.grad] + ([conv.bias.grad] if bias else [])
    for g, r, name in zip(got, want, ("weight", "x", "bias")):
        scale = float(r.abs().max())
        assert g.shape == r.shape
        assert float((g - r).abs().max()) <= (2e-4 if name != "x" else 2e-5) * scale, (kind, name, float((g - r).abs().max()), scale)


@pytest.mark.parametrize("kind,B,cin,cout,H,W,bias", [
    ("conv_s2", 1, 8, 16, 96, 192, False), ("conv_s2", 2, 16, 32, 24, 48, False), ("conv_s2", 1, 5, 3, 10, 36, True), ("conv_s2", 1, 32, 64, 24, 48, False),
    ("convT_s2", 1, 16, 8, 48, 96, False), ("convT_s2", 2, 64, 32, 12, 24, False), ("convT_s2", 1, 3, 5, 7, 33, True),
    ("convT_s1", 1, 8, 1, 96, 192, True), ("convT_s1", 2, 7, 2, 17, 70, True)])
def test_conv3x3_native_weight_gradient_strided_and_transposed(dev, kind, B, cin, cout, H, W, bias):
    """smvs_conv3x3_wgrad_strided behind _conv3x3 for the regulariser's other 3x3 layers: the stride-2 encoder convolutions, the stride-2
    transposed decoder convolutions (output_padding 1) and the stride-1 transposed output layer with its bias -- against torch autograd of
    the same layer (weight / bias gradient 2e-4 of the gradient's scale, forward bit-identical, input gradient torch's own)."""
    from satmvs_amd.modules import module as M
    torch.manual_seed(B * 1000 + cin * 10 + cout)
    if kind == "conv_s2":
        conv = torch.nn.Conv2d(cin, cout, 3, stride=2, padding=1, bias=bias).to(dev)
    elif kind == "convT_s2":
        conv = torch.nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=bias).to(dev)
    else:
        conv = torch.nn.ConvTranspose2d(cin, cout, 3, stride=1, padding=1, output_padding=0, bias=bias).to(dev)
    x = torch.randn(B, cin, H, W, device=dev, requires_grad=True)
    y0 = conv(x)
    gy = torch.randn_like(y0)
    y0.backward(gy)
    want = (conv.weight.grad.clone(), conv.bias.grad.clone() if bias else None, x.grad.clone())
    conv.zero_grad(); x.grad = None
    y1 = M._conv3x3(conv, x)
    assert y1.grad_fn is not None and "Conv3x3Wgrad" in type(y1.grad_fn).__name__
    assert torch.equal(y1, y0)
    y1.backward(gy)
    pairs = [(conv.weight.grad, want[0], "weight")] + ([(conv.bias.grad, want[1], "bias")] if bias else [])
    for got, ref, name in pairs:
        scale = float(ref.abs().max())
        assert got.shape == ref.shape
        assert float((got - ref).abs().max()) <= 2e-4 * scale, (kind, name, float((got - ref).abs().max()), scale)
    assert torch.allclose(x.grad, want[2], rtol=1e-4, atol=1e-4 * float(want[2].abs().max()))


@pytest.mark.parametrize("B", [1, 2])
def test_red_training_loop_streams_equal_single_stream(dev, B):
    """RED_Regularization's training loop with the ConvGRU levels on side streams and the planes software-pipelined
    (modules.module._planes_software_pipelined) against the same loop on one stream: the same kernels in the same per-tensor order, so
    the regularised volume is bit-identical and the gradients agree to the float atomics of the weight-gradient kernel (1e-5 of their
    scale).  Batch 1 runs the one-node ConvGRU cells, batch 2 the piecewise path."""
    from satmvs_amd.modules import module as M
    torch.manual_seed(21)
    red = M.RED_Regularization(16).to(dev).train()
    vol0 = torch.randn(B, 16, 10, 48, 96, device=dev)
    weight = torch.linspace(0, 1, B * 10 * 48 * 96, device=dev).view(B, 10, 48, 96)
    res = []
    saved = M.SW.train_streams
    try:
        for streams in (True, False):
            M.SW.train_streams = streams
            vol = vol0.clone().requires_grad_(True)
            red.zero_grad()
            out = red(vol)
            (out * weight).sum().backward()
            torch.cuda.synchronize()
            res.append((out.detach().clone(), vol.grad.clone(), [p.grad.clone() for p in red.parameters()]))
    finally:
        M.SW.train_streams = saved
    (o1, g1, p1), (o0, g0, p0) = res
    assert torch.equal(o1, o0)
    assert float((g1 - g0).abs().max()) <= 1e-5 * float(g0.abs().max())
    for a, b in zip(p1, p0):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9
