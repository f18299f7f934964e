"""GPU parity of the regulariser pipelines at BASELINE cfg3's REAL shapes (768x384 image, cascade 48/32/8):

    stage 1   96 x 192 x 48 planes, C = 32        stage 2   192 x 384 x 32, C = 16        stage 3   384 x 768 x 8, C = 8

The golden cascades are 64x128 and the pipeline test of tests/test_hip_end_to_end.py is 32x40: neither reaches the
launch shapes the kernels pick from the geometry at these sizes (chunks of 4 planes above 131 072 pixels, the
fine-level MFMA regime with one workgroup per tile column, the channel-split direct convolutions below 512
workgroups, every stage-2/3 grid).  Checked here, per stage:

  * smvs_red_volume_planes (the stream-pipelined plane loop) against oracle.c::red_step on the first planes of the
    sweep with the recurrent state carried (2e-5 class float32 round-off, tolerance scaled with the magnitude), and
    against the PyTorch composite of the same module (SMVS_RED_TORCH=1) on EVERY plane;
  * smvs_red_pred_planes (compute_depth_when_pred: variance plane -> RED step -> streaming float64 regression) against
    the composite: heights within 1e-3 m (north_star);
  * smvs_costreg_fwd and smvs_featnet_fwd against oracle.c at the same sizes.

The oracle side is pinned on the CPU by tests/test_oracle_golden.py::test_{red_step,costregnet,featurenet}_oracle_vs_reference.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H_TOL = 1e-3

# name, channels, H, W, planes, per-pixel hypotheses?
STAGES = [("stage1", 32, 96, 192, 48, False), ("stage2", 16, 192, 384, 32, True), ("stage3", 8, 384, 768, 8, True)]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    torch.backends.cudnn.benchmark = False
    return torch.device("cuda:0")


def _composite(env, fn):
    os.environ[env] = "1"
    try:
        return fn()
    finally:
        del os.environ[env]


def _stage_problem(C, H, W, D, per_pixel, dev, seed):
    """Seeded 3-view features (unit variance, like FeatureNet's outputs after its last 1x1), TLC-shaped RPCs of the
    stage's size, hypotheses: stage 1 = (B,D) planes over 0..400 m, stages 2-3 = per-pixel planes around a smooth
    surface (what the cascade hands them)."""
    from satmvs_amd import rpc_synth
    g = torch.Generator(device="cpu").manual_seed(seed)
    feats = [torch.randn((1, C, H, W), generator=g).to(dev) for _ in range(3)]
    rpc = torch.from_numpy(rpc_synth.make_view_rpcs(3, H, W, seed=seed)[None]).to(dev)
    if not per_pixel:
        dv = torch.linspace(0.0, 400.0, D).view(1, D).to(dev)
    else:
        yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
        surf = 200.0 + 60.0 * torch.sin(3.0 * xx) * torch.cos(2.0 * yy)
        half = 2.5 * D / 2
        dv = (surf[None, None] + torch.linspace(-half, half, D).view(1, D, 1, 1)).contiguous().to(dev)
    return feats, rpc, dv


def _weights(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


@pytest.mark.parametrize("name,C,H,W,D,per_pixel", STAGES)
def test_red_volume_pipeline_full_size(dev, oracle, arith, name, C, H, W, D, per_pixel):
    from satmvs_amd.modules.module import RED_Regularization
    from satmvs_amd.modules.warping import variance_cost_volume
    torch.manual_seed(11)
    reg = RED_Regularization(C, 8).to(dev).eval()
    feats, rpc, dv = _stage_problem(C, H, W, D, per_pixel, dev, seed=21)
    with torch.no_grad():
        assert reg._use_native(feats[0])
        got = reg.native_volume(feats, rpc, dv, "rpc", False)                       # smvs_red_volume_planes
        var = variance_cost_volume(feats, rpc, dv, "rpc", False)                     # bit-identical to the oracle's in the exact mode (test_hip_parity); both modes: `arith`
        ref = _composite("SMVS_RED_TORCH", lambda: reg(var))                         # PyTorch / MIOpen composite, every plane
    assert got.shape == ref.shape == (1, D, H, W)
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err <= 5e-5 * scale, "%s: pipeline vs composite %.3g (scale %.3g)" % (name, err, scale)
    # oracle.c on the first planes, state carried: the chunked front, the recurrent chain and the decoder of plane 1
    # consume what plane 0 left behind
    wt = _weights(reg)
    st = [np.zeros((1, 8, H, W), np.float32), np.zeros((1, 16, H // 2, W // 2), np.float32),
          np.zeros((1, 32, H // 4, W // 4), np.float32), np.zeros((1, 64, H // 8, W // 8), np.float32)]
    v = var[:, :, :3].cpu().numpy()
    for d in range(3 if name != "stage3" else 2):
        o, st = oracle.red_step(wt, v[:, :, d], st)
        e = float(np.abs(got[:, d].cpu().numpy() - o[:, 0]).max())
        assert e <= 5e-5 * scale, "%s plane %d: pipeline vs oracle %.3g (scale %.3g)" % (name, d, e, scale)


@pytest.mark.parametrize("name,C,H,W,D,per_pixel", STAGES)
def test_red_pred_pipeline_full_size(dev, arith, name, C, H, W, D, per_pixel):
    from satmvs_amd.modules.module import slice_RED_Regularization
    from satmvs_amd.networks.casred import compute_depth_when_pred
    torch.manual_seed(12)
    reg = slice_RED_Regularization(C, 8).to(dev).eval()
    feats, rpc, dv = _stage_problem(C, H, W, D, per_pixel, dev, seed=22)
    with torch.no_grad():
        a = compute_depth_when_pred(feats, rpc, dv, D, reg, "rpc", False)            # smvs_red_pred_planes
        b = _composite("SMVS_RED_TORCH", lambda: compute_depth_when_pred(feats, rpc, dv, D, reg, "rpc", False))
    err = float((a["depth"] - b["depth"]).abs().max())
    assert err <= H_TOL, "%s: height error %.3g m" % (name, err)
    np.testing.assert_allclose(a["photometric_confidence"].cpu().numpy(), b["photometric_confidence"].cpu().numpy(),
                               rtol=1e-3, atol=1e-5)


def _rand_bn(net):
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.7, 1.3); m.bias.data.normal_(0, 0.1)


@pytest.mark.parametrize("name,C,H,W,D,per_pixel", STAGES)
def test_costreg_full_size(dev, oracle, name, C, H, W, D, per_pixel):
    from satmvs_amd.modules.module import CostRegNet
    torch.manual_seed(13)
    net = CostRegNet(C, 8).to(dev).eval()
    _rand_bn(net)
    x = torch.randn(1, C, D, H, W, device=dev).abs_()                                  # a variance volume is non-negative
    with torch.no_grad():
        assert net._use_native(x)
        y = net(x)
    want = oracle.costregnet(_weights(net), x.cpu().numpy())
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(y.cpu().numpy() - want).max())
    assert err <= 2e-5 * scale, "%s: CostRegNet vs oracle %.3g (scale %.3g)" % (name, err, scale)


@pytest.mark.parametrize("arch", ["unet", "fpn"])
def test_featnet_full_size(dev, oracle, arch):
    from satmvs_amd.modules.module import FeatureNet
    torch.manual_seed(14)
    net = FeatureNet(base_channels=8, stride=4, num_stage=3, arch_mode=arch).to(dev).eval()
    _rand_bn(net)
    x = torch.randn(3, 3, 384, 768, device=dev)                                        # the three views of a 768x384 tile
    with torch.no_grad():
        assert net._use_native(x)
        y = net(x)
    o = oracle.featurenet(_weights(net), x.cpu().numpy(), arch)
    for i, k in enumerate(["stage1", "stage2", "stage3"]):
        scale = max(1.0, float(np.abs(o[i]).max()))
        err = float(np.abs(y[k].cpu().numpy() - o[i]).max())
        assert err <= 1e-5 * scale, "%s %s: FeatureNet vs oracle %.3g (scale %.3g)" % (arch, k, err, scale)
