"""A WELL-CONDITIONED whole-cascade parity case for the RED networks at the size bench.py times (3-view 768x384, planes 48/32/8).

tests/test_full_size_cascade.py compares the native plane pipeline with the torch / MIOpen composites on smooth random images and
random weights; there the softmax of a random regulariser is nearly flat over the whole height span, 32-48 recurrent planes turn ANY
float32 round-off into centimetres (stage 2: 5-6 cm float32-against-float32) and the test can only ask "as close to float64 as
MIOpen is, within 1.5x".  This file builds the case the contract is stated on instead:

  * three PHOTO-CONSISTENT views -- one textured surface rendered through the three RPCs (the recipe of tests/golden/gen_golden.py::
    gen_photo at the real tile, with the texture rendered straight into the three feature pyramids), so that the variance volume has
    one valley per pixel, along the surface;
  * regulariser weights that behave like TRAINED ones, without a checkpoint: the seeded random initialisation (scaled by 1/10, so that
    every layer still carries generic values) plus a deterministic photo-consistency path -- the candidate convolution of the
    full-resolution ConvGRU cell sums the negated variance channels at its centre tap, its update gate is biased towards "take the
    candidate", and the output layer (upconv2d, /root/reference/modules/module.py:612, :693) sums the cell's state -- and the output
    layer of every stage scaled by the smallest power-of-two gain that makes the stage's softmax peaky (mean photometric confidence
    >= 0.5; found once per network by running the native forward).  Random weights ALONE cannot serve: a gain only sharpens the
    multi-modal logits of a random regulariser into an arg max over unrelated planes, and a tie between two of them flips on the last
    bit of any float32 implementation (measured with gain 8: MIOpen's own composite 318-359 m from the float64 evaluation at stage 1);
  * per stage and arithmetic mode: native pipeline, all-composite pipeline and a float64 evaluation of the stage on the same inputs.

Heights within 1e-3 m (north_star) at EVERY pixel of EVERY stage, native against float64 AND native against the composite on the same
stage inputs -- no allowance; free-running through the cascade: every pixel at stage 1, 99.9 % of the pixels at stages 2-3 (see the
test's docstring).  Reference: /root/reference/networks/casred.py:285-333 (cascade), :161-238 (plane loop),
/root/reference/modules/module.py:653-693 (slice_RED_Regularization), :595-649 (RED_Regularization)."""
import os

import numpy as np
import pytest
import torch

from test_full_size_cascade import H, W, H_TOL, build_net, native_vs_composite, randomise_batchnorm, red_stages_against_float64

pytestmark = pytest.mark.gpu
MIN_CONFIDENCE = 0.5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    torch.backends.cudnn.benchmark = False
    return torch.device("cuda:0")


_RENDERED = {}
STAGES = (("stage1", 4, 32), ("stage2", 2, 16), ("stage3", 1, 8))


def photo_consistent_inputs(dev, seed=41):
    """(features: per view {stage: (1,C,h,w)}, {stage: rpc (1,3,170)}, height range (1,2), truth (H,W)).

    Three TLC-shaped RPC views (nadir / forward / backward tilt) of one smooth surface around 200 m that carries, per cascade stage and
    channel, a band-limited random texture defined ON THE GROUND (three octaves of Gaussian-filtered white noise, finest correlation
    length 1.5 pixels of that stage, so that the bilinear taps of the warp interpolate it faithfully).  A view's feature at a pixel is the
    texture at the ground point in which the pixel's ray meets the surface (fixed-point iteration on the inverse RPC): the three views
    agree exactly where the hypothesised height is the surface's and decorrelate away from it -- a variance volume with one valley per
    pixel.  These pyramids stand in for FeatureNet's outputs (a randomly initialised extractor's features are not discriminative: the
    arg min of their mean variance misses the surface by > 80 m at 10 % of the pixels; the extractor's own full-size parity is
    tests/test_full_size_regularisers.py::test_featnet_full_size)."""
    from satmvs_amd import rpc_synth
    if seed not in _RENDERED:
        _RENDERED[seed] = _render(seed)
    feats, rpc, truth = _RENDERED[seed]
    pm = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc[None], 4)).to(dev), "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc[None], 2)).to(dev),
          "stage3": torch.from_numpy(rpc[None].copy()).to(dev)}
    return [{k: torch.from_numpy(a).to(dev) for k, a in f.items()} for f in feats], pm, torch.tensor([[0.0, 400.0]], device=dev), truth


def _gauss_fields(rng, shape, sigmas):
    """Unit-variance sum of Gaussian-filtered white-noise fields (periodic, filtered in the Fourier domain), octave k weighted 1/sqrt(k+1)."""
    fy, fx = np.fft.fftfreq(shape[0])[:, None], np.fft.rfftfreq(shape[1])[None, :]
    out = np.zeros(shape)
    for k, sg in enumerate(sigmas):
        spec = np.fft.rfft2(rng.standard_normal(shape)) * np.exp(-2.0 * (np.pi * sg) ** 2 * (fy * fy + fx * fx))
        t = np.fft.irfft2(spec, s=shape)
        out += t / t.std() / np.sqrt(k + 1.0)
    return out / out.std()


def _render(seed):
    from satmvs_amd import rpc_synth
    rpc = rpc_synth.make_view_rpcs(3, H, W, seed=seed)
    rng = np.random.default_rng(seed + 1)
    lat0, lon0, ls, os_ = rpc[0, 2], rpc[0, 3], rpc[0, 7], rpc[0, 8]

    def surface(lat, lon):
        u, v = (lat - lat0) / ls, (lon - lon0) / os_
        return 200.0 + 22.0 * np.sin(2.1 * u + 0.4) * np.cos(1.7 * v - 0.3) + 9.0 * np.sin(4.3 * v + 1.0)

    gh, gw = int(2 * H * 1.2) + 8, int(2 * W * 1.2) + 8          # ground grid over the normalised lat / lon box: 2 cells per full-resolution pixel
    feats = [dict() for _ in range(3)]
    truth = None
    for st, s, C in STAGES:
        base = [_gauss_fields(rng, (gh, gw), [sg * s * 2.0 for sg in (1.5, 5.0, 16.0)]) for _ in range(4)]
        shifts = [(int(rng.integers(0, gh)), int(rng.integers(0, gw))) for _ in range(C)]        # channel c = field c % 4, rolled
        hs, ws = H // s, W // s
        yy, xx = np.meshgrid(np.arange(hs, dtype=np.float64), np.arange(ws, dtype=np.float64), indexing="ij")
        r_s = rpc_synth.rescale_rpc(rpc, s)
        for v in range(3):
            h = np.full(hs * ws, 200.0)
            for _ in range(12):
                lat, lon = rpc_synth.photo2obj(r_s[v], xx.ravel(), yy.ravel(), h)
                h = surface(lat, lon)
            lat, lon = rpc_synth.photo2obj(r_s[v], xx.ravel(), yy.ravel(), h)
            u, w_ = (lat - lat0) / ls, (lon - lon0) / os_
            fy = np.clip((1.0 - (u + 1.0) * 0.5) * (gh - 1), 0, gh - 1.001)
            fx = np.clip((w_ + 1.0) * 0.5 * (gw - 1), 0, gw - 1.001)
            y0, x0 = np.floor(fy).astype(np.int64), np.floor(fx).astype(np.int64)
            wy, wx = fy - y0, fx - x0
            out = np.zeros((1, C, hs, ws), np.float32)
            for c in range(C):
                t = base[c % 4]
                dy, dx = shifts[c]
                ya, yb, xa, xb = (y0 + dy) % gh, (y0 + 1 + dy) % gh, (x0 + dx) % gw, (x0 + 1 + dx) % gw
                out[0, c] = (t[ya, xa] * (1 - wy) * (1 - wx) + t[ya, xb] * (1 - wy) * wx + t[yb, xa] * wy * (1 - wx) + t[yb, xb] * wy * wx).reshape(hs, ws)
            feats[v][st] = out
            if v == 0 and s == 1:
                truth = h.reshape(hs, ws).astype(np.float32)
    return feats, rpc, truth


def trained_like(net, shrink=0.1, gamma=4.0, update_bias=-3.0, pedestal=1.0, seed=45):
    """See the module docstring: seeded weights * shrink + a photo-consistency path through conv_gru1 -> upconv2d (in place).

    GroupNorm(1, C) rescales every plane's candidate to unit variance, which would lift the fluctuations of a plane that is wrong for
    EVERY pixel to the level of the right plane's peak; half of the candidate channels therefore carry a constant pedestal (+-`pedestal`
    through the convolution's bias: they saturate the tanh and cancel in the output sum), so that the norm's scale is the same for all
    planes and the other half passes the negated mean variance through in the tanh's linear range -- with alternating signs (undone by
    the output layer), so that the plane's mean, which the norm subtracts, does not depend on how many pixels the plane is right for."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for reg in net.cost_regularization:
            for m in reg.modules():
                if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                    m.weight.mul_(shrink)
                    if m.bias is not None:
                        m.bias.mul_(shrink)
                elif isinstance(m, torch.nn.GroupNorm):                      # generic affine parameters instead of (1, 0)
                    m.weight.copy_((0.8 + 0.4 * torch.rand(m.weight.shape, generator=g)).to(m.weight.device))
                    m.bias.copy_((0.1 * torch.randn(m.bias.shape, generator=g)).to(m.bias.device))
            cell = reg.conv_gru1
            hc = cell.output_channel
            C = cell.output_conv.weight.shape[1] - hc                        # channels of the variance plane
            sign = torch.tensor([1.0, -1.0] * (hc // 4), device=cell.output_conv.bias.device)
            cell.output_conv.weight[hc // 2:, :C, 1, 1] += sign.view(-1, 1) / C  # candidate_j = +-sum_c (-var_c) / C + (small random part), j in the upper half
            cell.output_conv.bias[:hc // 2] += pedestal * sign                # ... and +-pedestal in the lower half: the plane's mean stays 0
            cell.output_norm.weight.fill_(gamma)
            cell.output_norm.bias.zero_()
            cell.update_gate_norm.bias.fill_(update_bias)                    # u = sigmoid(~update_bias): the state follows the candidate
            reg.upconv2d.weight[:, 0, 1, 1] += torch.cat([torch.ones(hc // 2, device=sign.device), sign])   # logit = sum_j +-state_j + (small random part)


def make_peaky(net, imgs, pm, dv, min_conf=MIN_CONFIDENCE, max_log2=16):
    """Scales the output layer of every stage's regulariser, stage by stage (a stage's hypotheses follow the previous stage's heights),
    by the smallest power of two that lifts the stage's mean photometric confidence to `min_conf`.  Returns ({stage: gain},
    {stage: mean confidence})."""
    gains, confs = {}, {}
    with torch.no_grad():
        for k in range(3):
            key = "stage%d" % (k + 1)
            out_layer = net.cost_regularization[k].upconv2d
            g = 1.0
            for _ in range(max_log2 + 1):
                out = net(imgs, pm, dv)[key]
                assert torch.isfinite(out["depth"]).all(), "gain %g overflows the plane loop's exp(double(logit))" % g
                conf = float(out["photometric_confidence"].mean())
                if conf >= min_conf:
                    break
                out_layer.weight.mul_(2.0)
                if out_layer.bias is not None:
                    out_layer.bias.mul_(2.0)
                g *= 2.0
            gains[key], confs[key] = g, conf
    return gains, confs


def conditioned_case(tag, dev):
    torch.manual_seed(43)
    net = build_net(tag, "rpc").to(dev).eval()
    randomise_batchnorm(net, 44)
    trained_like(net)
    feats, pm, dv, truth = photo_consistent_inputs(dev)
    net.feature.forward_views = lambda imgs: feats              # the rendered pyramids in FeatureNet's place (native, composite and float64 runs alike)
    imgs = torch.zeros((1, 3, 3, H, W), device=dev)             # only its size is read
    gains, confs = make_peaky(net, imgs, pm, dv)
    return net, imgs, pm, dv, truth, gains, confs


@pytest.mark.parametrize("tag", ["redinf", "red"])
def test_red_cascade_well_conditioned_full_size(dev, tag, arith):
    """Stage by stage on the SAME inputs (the composite run's incoming height map): native, composite and float64 evaluation agree within
    1e-3 m at EVERY pixel of EVERY stage -- measured 2.7e-4 ... 5e-4 m, the native pipeline as close to float64 as the MIOpen composite
    (profiles/r05_cascade_float64.txt).  Free-running (each cascade feeds its own stage-1 / stage-2 heights forward) stage 1 holds
    1e-3 m at every pixel; at stages 2-3 the 5e-4 m the two cascades differ by at stage 1 moves the hypotheses of the pixels that are
    NOT locked on (flat logits along the tile border, where the warped features leave the image) by centimetres, so there the bound is
    on 99.9 % of the pixels and the rest is reported."""
    net, imgs, pm, dv, truth, gains, confs = conditioned_case(tag, dev)
    for s, c in confs.items():
        assert c >= MIN_CONFIDENCE, "stage %s not peaky: mean confidence %.3f at gain %g" % (s, c, gains[s])
    err, a, b = native_vs_composite(net, imgs, pm, dv)
    free = {s: (a[s]["depth"].double() - b[s]["depth"].double()).abs() for s in err}
    frac = {s: float((e > H_TOL).double().mean()) for s, e in free.items()}
    f64 = red_stages_against_float64(net, imgs, pm, dv, "rpc", var_mode="current")
    ref_note = ""
    if arith == "fused":
        # REPORTED, not asserted: the same stages against the float64 evaluation on the REFERENCE's variance volume (the exact build) --
        # what the fused arithmetic's own 1e-5 * max(1, |v|) on the volume turns into behind a gain-16 softmax (2.1e-3 m at 3 of 294 912
        # pixels in round 5).  That is why the cascades and plane pipelines run the exact build unless asked (round 6: _lib.pipeline_arith_scope,
        # include/satmvs.h); the fused mode reaches this test only through the `arith` fixture's scope, and is held to 1e-3 m on its OWN
        # volume below.  bench.py's height_parity_vs_reference.conditioned_768x384 carries both figures on the driver's box.
        det = {}
        red_stages_against_float64(net, imgs, pm, dv, "rpc", detail=det)
        ref = {s: (d["native"].double() - d["float64"]).abs() for s, d in det.items()}
        ref_note = " | native (fused volume) vs float64 on the exact volume: max %s, fraction beyond 1e-3 m %s" % (
            {s: "%.3g" % float(e.max()) for s, e in ref.items()}, {s: "%.2g" % float((e > H_TOL).double().mean()) for s, e in ref.items()})
    herr = np.abs(a["stage3"]["depth"][0].cpu().numpy() - truth)[32:-32, 32:-32]
    msg = ("%s (%s arithmetic) gains %s confidence %s | same stage inputs (native-f64, composite-f64, native-composite) %s | free-running native vs composite: "
           "max %s, fraction of pixels beyond 1e-3 m %s | stage 3 vs the rendered surface: median %.2f m, 90 %% %.2f m") % (
        tag, arith, {s: int(g) for s, g in gains.items()}, {s: "%.3f" % c for s, c in confs.items()},
        {s: tuple("%.3g" % x for x in v) for s, v in f64.items()}, {s: "%.3g" % e for s, e in err.items()}, {s: "%.2g" % f for s, f in frac.items()},
        float(np.median(herr)), float(np.percentile(herr, 90))) + ref_note
    print(msg)
    log = os.environ.get("SMVS_CONDITIONED_LOG")
    if log:
        with open(log, "a") as f:
            f.write(msg + "\n")
    for s in ("stage1", "stage2", "stage3"):
        e_nat, e_comp, e_nc = f64[s]
        assert e_nat <= H_TOL, "%s %s: native %.3g m from the float64 evaluation (composite %.3g m)" % (tag, s, e_nat, e_comp)
        # native vs composite: 1e-3 m (measured <= 8.2e-4).  The composite's own distance from float64 moves with the solver MIOpen picks on
        # a box (seen elsewhere in this suite: two calls of one nn.Conv2d differing in the last bits), so if the two float32 pipelines
        # ever land on opposite sides of float64 the bound is the triangle's: both within 1e-3 m of float64, 1.5e-3 m apart at most
        assert e_nc <= H_TOL or (e_comp <= H_TOL and e_nc <= 1.5 * H_TOL), "%s %s: native %.3g m from the composite on the same stage inputs (composite %.3g m from float64)" % (tag, s, e_nc, e_comp)
        assert frac[s] <= 1e-3, "%s %s: %.2g of the pixels differ by more than 1e-3 m free-running" % (tag, s, frac[s])
    assert err["stage1"] <= H_TOL, "%s stage1: free-running native vs composite %.3g m" % (tag, err["stage1"])
    # the photo-consistency path makes the cascade follow the rendered surface (away from the tile border)
    assert float(np.median(herr)) < 2.0
