"""Whole cascades at the size bench.py times (BASELINE cfg3 / cfg5: 3-view 768x384, planes 48/32/8): the native pipeline --
FeatureNet kernels -> per stage [hypotheses generated in the kernel -> fused warp + variance build -> RED / CostRegNet kernels
-> regression] -- against the SAME network with every regulariser, feature extractor and plane loop forced onto the stock
PyTorch composites (SMVS_RED_TORCH / SMVS_COSTREG_TORCH / SMVS_FEATNET_TORCH = 1: MIOpen convolutions, materialised
hypotheses) on the same seeded weights and inputs.  Heights within 1e-3 m per stage (north_star), in both arithmetic modes of
the variance build.

Covers what the per-operator full-size tests (tests/test_full_size_regularisers.py: synthetic hypotheses, one stage at a
time) and the 64x128 reference goldens (tests/test_hip_end_to_end.py) leave open: the cascade glue at the real launch shapes
-- stage-to-stage hand-over of the height map, GeneratedHeights at scale 2 and 1, the 1/4-, 1/2- and full-resolution RPCs /
projection matrices (/root/reference/networks/casred.py:285-333, networks/casmvs.py:61-130, networks/ucs.py:79-157).
The composites themselves are pinned against the reference by the goldens."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H_TOL = 1e-3
H, W = 384, 768
COMPOSITE_SWITCHES = ("SMVS_RED_TORCH", "SMVS_COSTREG_TORCH", "SMVS_FEATNET_TORCH")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    torch.backends.cudnn.benchmark = False
    return torch.device("cuda:0")


def build_net(tag, geo):
    from satmvs_amd.networks import casmvs, casred, ucs
    if tag == "redinf":
        return casred.Infer_CascadeREDNet(geo, min_interval=2.5, ndepths=[48, 32, 8])
    if tag == "red":
        return casred.CascadeREDNet(geo, min_interval=2.5, ndepths=[48, 32, 8])
    if tag == "casmvs":
        return casmvs.CascadeMVSNet(geo, min_interval=2.5, ndepths=[48, 32, 8])
    return ucs.UCSNet(geo, stage_configs=[48, 32, 8])


def cascade_inputs(geo, dev, seed=5):
    """Seeded 3-view tile: smooth random images (so that FeatureNet's outputs have structure), TLC-shaped RPCs or the synthetic
    pinhole rig at the three cascade scales, the (B,2) height range."""
    from satmvs_amd import rpc_synth
    g = torch.Generator(device="cpu").manual_seed(seed)
    imgs = torch.nn.functional.avg_pool2d(torch.randn((3, 3, H + 4, W + 4), generator=g), 5, stride=1).contiguous()[None].to(dev)
    if geo == "rpc":
        rpc = rpc_synth.make_view_rpcs(3, H, W, seed=seed)[None]
        pm = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev), "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev),
              "stage3": torch.from_numpy(rpc).to(dev)}
        dv = torch.tensor([[0.0, 400.0]], device=dev)
    else:
        full = np.zeros((1, 3, 4, 4))                              # K @ E per view, as tests/test_hip_parity.py::_inputs
        for v in range(3):
            f = 1.1 * W
            K = np.array([[f, 0, W / 2.0, 0], [0, f, H / 2.0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1]])
            E = np.eye(4)
            E[:3, 3] = [25.0 * v * (-1) ** v, 3.0 * v, 0.5 * v]
            full[0, v] = K @ E
        pm = {}
        for name, s in (("stage1", 4), ("stage2", 2), ("stage3", 1)):
            m = full.copy()
            m[:, :, :2, :] /= s
            pm[name] = torch.from_numpy(m).to(dev)
        dv = torch.tensor([[400.0, 700.0]], device=dev)
    return imgs, pm, dv


def randomise_batchnorm(net, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            m.running_mean.copy_(0.2 * torch.randn(m.running_mean.shape, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
            m.weight.data.copy_(0.7 + 0.6 * torch.rand(m.weight.shape, generator=g))
            m.bias.data.copy_(0.1 * torch.randn(m.bias.shape, generator=g))


def _composite_forward(net, imgs, pm, dv):
    for k in COMPOSITE_SWITCHES:
        os.environ[k] = "1"
    try:
        out = net(imgs, pm, dv)
        torch.cuda.synchronize()
        return out
    finally:
        for k in COMPOSITE_SWITCHES:
            del os.environ[k]


def _stage_errors(a, b, key="depth"):
    out = {}
    for s in ("stage1", "stage2", "stage3"):
        assert a[s][key].shape == b[s]["depth"].shape
        assert torch.isfinite(a[s][key]).all()
        out[s] = float((a[s][key].double() - b[s]["depth"].double()).abs().max())
    return out


def native_vs_composite(net, imgs, pm, dv):
    """Free-running comparison: {stage: max |height difference| in m} of the native forward against the all-composite forward."""
    with torch.no_grad():
        a = net(imgs, pm, dv)
        torch.cuda.synchronize()
        b = _composite_forward(net, imgs, pm, dv)
    return _stage_errors(a, b), a, b


def red_stages_against_float64(net, imgs, pm, dv, geo, detail=None, var_mode="exact"):
    """Stage by stage for the RED cascades, every stage on the SAME inputs (the composite run's features and incoming height map):
         native    the stage's native pipeline (hypotheses generated in the kernels -> variance planes -> RED kernels -> regression)
         composite the same stage on torch / MIOpen float32 operators
         float64   the stage's regulariser and regression evaluated in float64 (torch operators on a .double() copy of the
                   module) on the float32 variance volume of the exact build (bit-identical to the reference's); var_mode="current":
                   on the variance volume of the arithmetic in force, i.e. what the two float32 pipelines consumed
    Returns {stage: (max |native - float64|, max |composite - float64|, max |native - composite|)} in metres."""
    import copy
    from satmvs_amd import _lib
    from satmvs_amd.modules.depth_range import GeneratedHeights, stage_hypotheses
    from satmvs_amd.modules.warping import variance_cost_volume
    out = {}
    with torch.no_grad():
        os.environ["SMVS_FEATNET_TORCH"] = "1"
        try:
            feats_all = net.feature.forward_views(imgs)
        finally:
            del os.environ["SMVS_FEATNET_TORCH"]
        prev = None
        for k in range(3):
            key = "stage%d" % (k + 1)
            feats = [f[key] for f in feats_all]
            scale = int(net.stage_infos[key]["scale"])
            reg = net.cost_regularization[k]
            nd = net.ndepths[k]
            dvg = stage_hypotheses(prev, dv, nd, net.depth_interals_ratio[k] * net.min_interval, (H, W), (H // scale, W // scale),
                                   imgs.dtype, imgs.device, imgs.shape[0])
            nat = type(net).compute(feats, pm[key], depth_values=dvg, num_depth=nd, cost_regularization=reg, geo_model=geo, use_qc=False)["depth"]
            os.environ["SMVS_RED_TORCH"] = "1"
            try:
                comp = type(net).compute(feats, pm[key], depth_values=dvg, num_depth=nd, cost_regularization=reg, geo_model=geo, use_qc=False)["depth"]
                dvt = dvg.materialize() if isinstance(dvg, GeneratedHeights) else dvg
                with _lib.arith_scope(None if var_mode == "current" else var_mode):      # "current": the variance the stage itself consumed
                    var = variance_cost_volume(feats, pm[key], dvt, geo)
                reg64 = copy.deepcopy(reg).double()
                st = reg64.initial_states(1, H // scale, W // scale, imgs.device, torch.float64)
                logits = []
                for d in range(nd):
                    r, *st = reg64.step(var[:, :, d].double(), *st)
                    logits.append(r)
                hv = dvt.double() if dvt.dim() == 4 else dvt.double().view(1, nd, 1, 1)
                if type(net).__name__ == "Infer_CascadeREDNet":
                    # the pred loop's own regression (casred.py:218-236): exp(double(logit)) WITHOUT a maximum subtracted, sums + 1e-10 --
                    # not the same function as a softmax where every plane's exp underflows (strongly negative logits give height 0)
                    p64 = torch.exp(torch.stack(logits, 1).squeeze(2))
                    es = p64.sum(1)
                    h64 = (p64 * hv).sum(1) / (es + 1e-10)
                    p64 = p64 / (es + 1e-10).unsqueeze(1)
                else:
                    p64 = torch.softmax(torch.stack(logits, 1).squeeze(2), 1)
                    h64 = (p64 * hv).sum(1)
            finally:
                del os.environ["SMVS_RED_TORCH"]
            out[key] = (float((nat.double() - h64).abs().max()), float((comp.double() - h64).abs().max()), float((nat - comp).abs().max()))
            if detail is not None:
                detail[key] = {"native": nat, "composite": comp, "float64": h64, "p64": p64.max(1)[0], "hyp": dvt}
            prev = comp
            del var, logits, p64, reg64
    return out


@pytest.mark.parametrize("tag,geo", [("redinf", "rpc"), ("red", "rpc"), ("casmvs", "rpc"), ("ucs", "rpc"), ("redinf", "pinhole"), ("ucs", "pinhole")])
def test_cascade_native_matches_composites_at_full_size(dev, tag, geo, arith):
    """CascadeMVSNet / UCSNet: every stage within 1e-3 m of the all-composite forward.

    The RED cascades with RANDOM weights are ill-conditioned at this size: 32-48 recurrent ConvGRU steps over 7e4-3e5 pixels turn
    the float32 round-off of ANY convolution implementation (1e-5 relative on the logits: tests/test_full_size_regularisers.py)
    into centimetres of height, because the softmax of a random regulariser is nearly flat over a 40-400 m span (measured:
    swapping only FeatureNet for its composite, 1e-6 relative on the features, moves stage 2 by 4e-2 m; trained weights lock
    onto the photo-consistent plane instead, and the 64x128 reference goldens hold 1e-3 m).  A float32-vs-float32 comparison
    therefore cannot hold 1e-3 m there; each is compared with a FLOAT64 evaluation of the same stage instead, as a sanity bound (see
    the comment at the assertion); the contract itself is held by tests/test_full_size_red_conditioned.py (round 5).  Free-running,
    the two float32 cascades stay within a loose sanity bound."""
    torch.manual_seed(31)
    net = build_net(tag, geo).to(dev).eval()
    randomise_batchnorm(net, 32)
    imgs, pm, dv = cascade_inputs(geo, dev)
    err, a, b = native_vs_composite(net, imgs, pm, dv)
    if tag in ("red", "redinf"):
        assert max(err.values()) <= 0.25, err
        for s in err:
            assert float((a[s]["depth"] - b[s]["depth"]).abs().mean()) <= 0.02, s
        # (all three on the variance volume of the arithmetic in force.  This random-weight case amplifies round-off ~1e5-fold at stage 2,
        #  chaotically: the factor between two float32 implementations' distances from float64 moves between 1.1 and 1.7 from run to run
        #  (atomics in the GroupNorm statistics), so the bound here is a sanity bound -- 2.5x; the CONTRACT, 1e-3 m at every pixel with no
        #  allowance, is held on the well-conditioned cascade of tests/test_full_size_red_conditioned.py, where native and composite are
        #  equally close to float64: 2.7e-4 ... 5.3e-4 m)
        cur = red_stages_against_float64(net, imgs, pm, dv, geo, var_mode="current")
        # second check (round-5 advisor): the same stages against float64 on the EXACT variance volume, whatever arithmetic is in force -- a
        # looser sanity bound (4x), and the measured ratios go on record so that a drift of the native pipeline shows
        # (SMVS_CASCADE_LOG=<file>; profiles/r06_cascade_float64.txt)
        exa = cur if arith == "exact" else red_stages_against_float64(net, imgs, pm, dv, geo, var_mode="exact")
        log = os.environ.get("SMVS_CASCADE_LOG")
        if log:
            with open(log, "a") as f:
                f.write("%s %s (%s arithmetic) native / composite distance from float64, metres [ratio]: own volume %s | exact volume %s\n" % (
                    tag, geo, arith,
                    {s: "%.3g / %.3g [%.2f]" % (v[0], v[1], v[0] / max(v[1], 1e-30)) for s, v in cur.items()},
                    {s: "%.3g / %.3g [%.2f]" % (v[0], v[1], v[0] / max(v[1], 1e-30)) for s, v in exa.items()}))
        for s, (e_nat, e_comp, e_nc) in cur.items():
            assert e_nat <= max(H_TOL, 2.5 * e_comp), "%s %s %s (%s arithmetic): native %.3g m from float64, composite %.3g m, apart %.3g m" % (
                tag, geo, s, arith, e_nat, e_comp, e_nc)
        for s, (e_nat, e_comp, e_nc) in exa.items():
            assert e_nat <= max(H_TOL, 4.0 * e_comp), "%s %s %s (%s arithmetic, exact-volume float64): native %.3g m, composite %.3g m" % (
                tag, geo, s, arith, e_nat, e_comp)
    else:
        for s, e in err.items():
            assert e <= H_TOL, "%s %s %s (%s arithmetic): native vs composite height difference %.3g m" % (tag, geo, s, arith, e)
            np.testing.assert_allclose(a[s]["photometric_confidence"].cpu().numpy(), b[s]["photometric_confidence"].cpu().numpy(), rtol=2e-3, atol=2e-5)
    assert a["stage3"]["depth"].shape == (1, H, W)
    assert float(a["stage3"]["depth"].std()) > 0.05           # not a constant map


def test_training_step_full_size_native_vs_composite(dev):
    """CascadeREDNet.train() at the real tile (3-view 768x384, planes 48/32/8): loss and every parameter gradient of the shipped training
    path -- native layers under autograd, one-node ConvGRU cells, weight gradients deferred to one launch per layer, ConvGRU levels on
    side streams with the plane loop software-pipelined -- against the SAME step with all of that switched off (the module's
    switches (satmvs_amd/modules/switches.py: SW.train_composite_mask / SW.train_streams): torch's convolutions, GroupNorm and element-wise operators, one stream, the
    plain plane loop; that path is pinned against the reference's own step at 64x128, tests/golden/train_step.npz).  Loss to 1e-5;
    a parameter's gradient differs by float32 summation order only (atomics, split-K, another association over the planes), carried
    through up to 48 recurrent planes: 1e-2 of its largest entry at worst (measured 8e-3: the stride-2 encoder
    convolution of stage 2; FeatureNet, downstream of everything, 4e-3), 1e-3 for the median parameter."""
    import torch.nn.functional as F
    from satmvs_amd.modules import module as M
    imgs, pm, dv = cascade_inputs("rpc", dev)
    gts = {s: torch.full((1, H // k, W // k), 210.0, device=dev) for s, k in (("stage1", 4), ("stage2", 2), ("stage3", 1))}

    def run():
        torch.manual_seed(3)
        net = build_net("red", "rpc").to(dev).train()
        out = net(imgs, pm, dv)
        loss = sum(w * F.smooth_l1_loss(out[s]["depth"], gts[s], reduction="mean") for s, w in (("stage1", 0.5), ("stage2", 1.0), ("stage3", 2.0)))
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}

    saved = (M.SW.train_composite_mask, M.SW.train_streams)
    try:
        loss_n, g_n = run()
        M.SW.train_composite_mask, M.SW.train_streams = 63, False
        loss_c, g_c = run()
    finally:
        M.SW.train_composite_mask, M.SW.train_streams = saved
    assert abs(loss_n - loss_c) <= 1e-5 * abs(loss_c), (loss_n, loss_c)
    assert set(g_n) == set(g_c) and len(g_n) > 150
    rel = {}
    biggest = max(float(g.abs().max()) for g in g_c.values())
    for k in g_c:
        scale = float(g_c[k].abs().max())
        if scale < 1e-6 * biggest:            # e.g. the output layer's bias: softmax over the planes is shift-invariant, its gradient is round-off in both runs
            assert float(g_n[k].abs().max()) < 1e-5 * biggest, k
            continue
        rel[k] = float((g_n[k] - g_c[k]).abs().max()) / scale
    top = sorted(rel.items(), key=lambda kv: -kv[1])[:5]
    print("largest relative gradient differences:", ", ".join("%s %.2g" % kv for kv in top))
    assert top[0][1] <= 1e-2, top
    assert sorted(rel.values())[len(rel) // 2] <= 1e-3                # the median parameter
