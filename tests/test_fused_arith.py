"""GPU parity of the library's DEFAULT arithmetic of the variance build (smvs_set_arith(SMVS_ARITH_FUSED), include/satmvs.h).

The fused instance keeps the float64 geometry, the float32 tap coordinates and the bilinear weights of the exact instance
and takes the variance of the differences to the ref feature (11 instead of 22 packed operations per plane and channel pair
at 3 views).  It is NOT bit-identical to the reference's float32 sequence (networks/casred.py:26-53); the contract is

  * volume:  |fused - reference| <= 1e-5 * max(1, |reference|)            (<= 1e-5 abs on O(1) features)
  * heights: |fused - reference| <= 1e-3 m                                 (north_star) -- the reference-golden end-to-end
             tests (cascades, pred / train path, photo-consistent problem, training step) run in both modes: `arith` fixture
  * against a float64 evaluation of the same taps (oracle.costvol_variance_f64): on photo-consistent features (what the
    networks see) the fused error is 6x smaller than the reference's own -- no  meansq - mean^2  cancellation; on independent
    random features it equals the reference's rounding error at 2-3 views and is 2-4x it at 4-8 views (<= 2e-7 abs).

Same input cases as the bit-level tests of tests/test_hip_parity.py (which run the exact instance), plus the seeded fuzzer.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import test_hip_parity as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
EPS = 2.0 ** -24


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


@pytest.fixture()
def fused():
    from satmvs_amd import _lib
    _lib.set_arith("fused")
    with _lib.arith_scope("fused"):         # (the plane pipelines / cascades follow a scope, not the process default)
        yield
    _lib.set_arith(os.environ.get("SMVS_ARITH", "exact"))


def _within_contract(got, want):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    m = ~np.isnan(want)
    d = np.abs(got[m].astype(np.float64) - want[m].astype(np.float64))
    tol = 1e-5 * np.maximum(1.0, np.abs(want[m].astype(np.float64)))
    worst = float((d / tol).max()) if d.size else 0.0
    assert worst <= 1.0, "largest |delta| / (1e-5 max(1,|v|)) = %.3g" % worst
    return worst


def test_default_mode_is_fused(dev):
    """A fresh process without SMVS_ARITH (the test suite sets it to "exact") starts in the fused mode."""
    env = {k: v for k, v in os.environ.items() if k != "SMVS_ARITH"}
    out = subprocess.run([sys.executable, "-c", "import satmvs_amd; print(satmvs_amd.get_arith())"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("fused"), out.stdout + out.stderr


def test_set_arith_round_trip(dev):
    from satmvs_amd import _lib
    prev = _lib.set_arith("fused")
    assert _lib.get_arith() == "fused"
    assert _lib.set_arith("exact") == "fused" and _lib.get_arith() == "exact"
    assert _lib.load().smvs_set_arith(7) == -1 and _lib.get_arith() == "exact"
    with pytest.raises(ValueError):
        _lib.set_arith("fast")
    _lib.set_arith(prev)


@pytest.mark.parametrize("cfg", [
    dict(B=1, V=3, C=32, D=32, H=128, W=256, jitter=True),     # BASELINE config 1 shape (8 planes per wave)
    dict(B=2, V=5, C=16, D=5, H=33, W=70, jitter=True),        # 4 sources: two units per plane, ragged tile, batch 2
    dict(B=1, V=2, C=10, D=9, H=17, W=130, jitter=False),      # direct kernel (generic channel count), one source
    dict(B=1, V=8, C=8, D=3, H=8, W=64, jitter=True),          # 7 sources (one source per unit)
    dict(B=1, V=3, C=8, D=1, H=4, W=3, jitter=True),           # smaller than one tile, single plane
    dict(B=2, V=3, C=16, D=11, H=37, W=70, jitter=True),       # odd plane count
    dict(B=1, V=2, C=32, D=6, H=20, W=40, jitter=False),       # staged kernel, one source
    dict(B=1, V=4, C=32, D=8, H=40, W=72, jitter=True),        # 3 sources (odd: compiler-scheduled units)
    dict(B=1, V=6, C=16, D=5, H=40, W=72, jitter=True),        # 5 sources
    dict(B=2, V=7, C=32, D=4, H=24, W=66, jitter=False),       # 6 sources: three units per plane (middle-unit tail)
    dict(B=2, V=5, C=16, D=8, H=38, W=70, jitter=True),        # shared-box form: 4 sources, 8 planes, ragged tile, batch 2
    dict(B=1, V=4, C=8, D=16, H=24, W=100, jitter=False),      # shared-box form: 3 sources (one source per unit), (B,D) heights
])
def test_fused_costvol_vs_oracle_and_float64(dev, oracle, fused, cfg):
    from satmvs_amd import _lib
    from satmvs_amd.modules import warping
    feats, rpc, depth = T._inputs(cfg["B"], cfg["V"], cfg["C"], cfg["D"], cfg["H"], cfg["W"], seed=3, jitter=cfg["jitter"])
    want = oracle.costvol_variance(feats, rpc, depth, "rpc")
    f, r, d = [T._t(x, dev) for x in feats], T._t(rpc, dev), T._t(depth, dev)
    got = warping.variance_cost_volume(f, r, d, "rpc").cpu().numpy()
    _within_contract(got, want)
    # Against the float64 evaluation (scale = sum(X^2)/V, X = sum |corner| * weight).  On these INDEPENDENT random features
    # (the worst case: the differences to the ref feature are as large as the features) the fused error equals the
    # reference's own rounding error at 2-3 views (rms ratio 1.0) and is 2-4 times it at 4-8 views (measured: 2.2 / 2.6 / 3.7
    # at 4 / 5 / 7 views -- S*S/V cancels like the reference's mean^2), i.e. <= 2e-7 abs; on photo-consistent features, what
    # the networks see, it is 6 times SMALLER than the reference's (test_photo_consistent_peaky_problem[fused]).
    V = cfg["V"]
    truth, scale = oracle.costvol_variance_f64(feats, rpc, depth, "rpc")
    e_fused = np.abs(got - truth)
    e_ref = np.abs(want - truth)
    worst_fused, worst_ref = float((e_fused / (scale + 1e-30)).max()), float((e_ref / (scale + 1e-30)).max())
    assert worst_ref <= 8 * EPS and worst_fused <= 8 * V * EPS, (worst_fused / EPS, worst_ref / EPS)
    assert np.sqrt((e_fused ** 2).mean()) <= (1.05 if V <= 3 else V) * np.sqrt((e_ref ** 2).mean())
    # the exact instance reproduces the reference bit for bit on the same inputs (the mode travels with the call: the inner scope wins
    # over the `fused` fixture's)
    with _lib.arith_scope("exact"):
        T._close_f32(warping.variance_cost_volume(f, r, d, "rpc"), want)


def test_fused_costvol_golden(dev, golden, fused):
    from satmvs_amd.modules import warping
    g = golden("costvol")
    feats = [T._t(f, dev) for f in g["feats"]]
    _within_contract(warping.variance_cost_volume(feats, T._t(g["rpc"], dev), T._t(g["depth"], dev), "rpc"), g["variance_rpc"])
    _within_contract(warping.variance_cost_volume(feats, T._t(g["proj"], dev), T._t(g["depth_pin"], dev), "pinhole"), g["variance_pin"])


def test_fused_costvol_pinhole_vs_oracle(dev, oracle, fused):
    from satmvs_amd.modules import warping
    feats, proj, depth = T._inputs(1, 3, 16, 12, 48, 96, seed=4, geo="pinhole")
    want = oracle.costvol_variance(feats, proj, depth, "pinhole")
    _within_contract(warping.variance_cost_volume([T._t(f, dev) for f in feats], T._t(proj, dev), T._t(depth, dev), "pinhole"), want)


@pytest.mark.parametrize("C", [8, 16, 10])                      # staged kernels / direct kernel
def test_fused_out_of_image_and_nan(dev, oracle, fused, C):
    """Taps pushed off the source image contribute zero-padded samples, NaN heights give NaN voxels: as the reference."""
    from satmvs_amd.modules import warping
    feats, rpc, depth = T._inputs(1, 3, C, 6, 32, 64, seed=5)
    depth[:, 0] = -4000.0
    depth[:, 1] = 6000.0
    depth[0, 2, 3, 5] = np.nan
    want = oracle.costvol_variance(feats, rpc, depth, "rpc")
    _within_contract(warping.variance_cost_volume([T._t(f, dev) for f in feats], T._t(rpc, dev), T._t(depth, dev), "rpc"), want)


def test_fused_paths_agree_bit_for_bit(dev, fused):
    """Plane windows (1 / 2 / 4 / 8 planes per wave), the oversized-box path and the whole-volume launch run the same
    operation sequence: identical bits, so a sharded build equals the unsharded one in the fused mode too."""
    from satmvs_amd.modules import warping
    feats, rpc, depth = T._inputs(1, 3, 32, 24, 40, 72, seed=6)
    f = [T._t(x, dev) for x in feats]
    r, d = T._t(rpc, dev), T._t(depth, dev)
    full = warping.variance_cost_volume(f, r, d, "rpc")
    for lo, hi in ((0, 1), (3, 11), (23, 24), (8, 24), (0, 2), (4, 8), (0, 16)):
        part = warping.variance_cost_volume(f, r, d, "rpc", d_begin=lo, d_end=hi)
        assert torch.equal(part, full[:, :, lo:hi]), (lo, hi)
    # planes 57 m apart: the tap boxes of an 8-plane group do not fit the staged tile -> direct gathers inside the staged kernel
    wide = np.broadcast_to(np.linspace(0.0, 3000.0, 8, dtype=np.float32).reshape(1, 8, 1, 1), (1, 8, 40, 72)).copy()
    w = T._t(wide, dev)
    whole = warping.variance_cost_volume(f, r, w, "rpc")
    for pl in range(8):
        assert torch.equal(warping.variance_cost_volume(f, r, w, "rpc", d_begin=pl, d_end=pl + 1), whole[:, :, pl:pl + 1]), pl


@pytest.mark.parametrize("cfg", [
    dict(V=3, C=32, D=64, H=384, W=768, planes=(0, 31, 63)),      # config 2 (headline metric shape)
    dict(V=5, C=32, D=8, H=768, W=1536, planes=(2,), span=(0.0, 44.4)),   # config 4 shard at its real spacing (staged 4-source kernel)
    dict(V=3, C=8, D=8, H=384, W=768, planes=(0, 7), span=(190.0, 207.5)),    # config 3, stage 3: C=8
    dict(V=3, C=32, D=64, H=384, W=768, planes=(40,), geo="pinhole"),          # config 5: homography volume
])
def test_fused_full_size(dev, oracle, fused, cfg):
    """BASELINE.json sizes in the default mode: shard consistency and idempotence bit for bit, constant feature maps, whole
    oracle planes within the contract."""
    from satmvs_amd.modules import warping
    V, C, D, H, W = cfg["V"], cfg["C"], cfg["D"], cfg["H"], cfg["W"]
    geo = cfg.get("geo", "rpc")
    feats, rpc, depth = T._inputs(1, V, C, D, H, W, seed=11, geo=geo)
    if "span" in cfg:
        lo, hi = cfg["span"]
        rng = np.random.default_rng(12)
        depth = (np.linspace(lo, hi, D).reshape(1, D, 1, 1) + rng.normal(0, 0.5, (1, D, H, W))).astype(np.float32)
    f = [T._t(x, dev) for x in feats]
    r, d = T._t(rpc, dev), T._t(depth, dev)
    full = warping.variance_cost_volume(f, r, d, geo)
    assert torch.isfinite(full).all()
    assert full.min().item() > -1e-6                         # the difference form does not cancel: non-negative to the last few ulps
    a = warping.variance_cost_volume(f, r, d, geo, d_begin=0, d_end=D // 2)
    b = warping.variance_cost_volume(f, r, d, geo, d_begin=D // 2, d_end=D)
    assert torch.equal(torch.cat([a, b], 2), full)
    assert torch.equal(warping.variance_cost_volume(f, r, d, geo), full)
    consts = torch.arange(1, V * C + 1, dtype=torch.float32, device=dev).view(V, C) / 7.0
    cf = [consts[v].view(1, C, 1, 1).expand(1, C, H, W).contiguous() for v in range(V)]
    mid = torch.full((1, 1, H, W), 200.0 if geo == "rpc" else 550.0, dtype=torch.float32, device=dev)
    z = warping.variance_cost_volume(cf, r, mid, geo)
    want_c = (consts.double() ** 2).mean(0) - consts.double().mean(0) ** 2
    m = min(64 if geo == "rpc" else 96, H // 4)
    err = (z[0, :, 0, m:-m, m:-m].double() - want_c.view(C, 1, 1)).abs().max().item()
    assert err < 1e-5 * float(want_c.max()), err               # (the exact instance needs 1e-3 here: cancellation)
    for pl in cfg["planes"]:
        want = oracle.costvol_variance(feats, rpc, depth, geo, d_begin=pl, d_end=pl + 1)[:, :, pl]
        _within_contract(full[:, :, pl], want)


def test_arithmetic_travels_with_the_call(dev):
    """include/satmvs.h, SMVS_CALL_ARITH_*: inside satmvs_amd._lib.arith_scope a cost-volume call carries its own arithmetic in its
    arguments -- the process default (what smvs_set_arith moves; "exact" in this suite) is neither read nor changed.  Same bits as
    the same mode selected process-wide, for the tensor form, the generated-heights form (smvs_height_gen.arith) and the plane
    pipeline; two threads in different scopes at the same time each get their own (nn.DataParallel replicas call from threads)."""
    import threading
    from satmvs_amd import _lib
    from satmvs_amd.modules import warping
    from satmvs_amd.modules.depth_range import GeneratedHeights
    assert _lib.get_arith() == "exact"
    feats, rpc, depth = T._inputs(1, 3, 16, 8, 48, 96, seed=33)
    f = [T._t(x, dev) for x in feats]
    r, d = T._t(rpc, dev), T._t(depth, dev)
    rng = np.random.default_rng(4)
    gen = GeneratedHeights(T._t((200.0 + rng.normal(0, 5.0, (1, 24, 48))).astype(np.float32), dev), 8, 5.0, (96, 192), (48, 96))
    want = {}
    for mode in ("exact", "fused"):
        _lib.set_arith(mode)
        want[mode] = (warping.variance_cost_volume(f, r, d, "rpc"), warping.variance_cost_volume(f, r, gen, "rpc"))
    _lib.set_arith("exact")
    assert not torch.equal(want["exact"][0], want["fused"][0]) and not torch.equal(want["exact"][1], want["fused"][1])
    for mode in ("fused", "exact"):
        with _lib.arith_scope(mode):
            assert _lib.call_arith() == mode and _lib.get_arith() == "exact"
            assert torch.equal(warping.variance_cost_volume(f, r, d, "rpc"), want[mode][0])
            assert torch.equal(warping.variance_cost_volume(f, r, gen, "rpc"), want[mode][1])
            with _lib.arith_scope(None):                                  # no-op scope
                assert _lib.call_arith() == mode
        assert _lib.call_arith_bits() == 0 and _lib.get_arith() == "exact"
    # with the DEFAULT moved to fused, a call scoped "exact" still runs exact
    _lib.set_arith("fused")
    try:
        with _lib.arith_scope("exact"):
            assert torch.equal(warping.variance_cost_volume(f, r, d, "rpc"), want["exact"][0])
        assert torch.equal(warping.variance_cost_volume(f, r, d, "rpc"), want["fused"][0])
    finally:
        _lib.set_arith("exact")
    # two threads, two scopes, interleaved calls
    got, errs = {}, []
    barrier = threading.Barrier(2)

    def worker(mode):
        try:
            torch.cuda.set_device(dev)
            with _lib.arith_scope(mode):
                outs = []
                for _ in range(6):
                    barrier.wait()
                    outs.append(warping.variance_cost_volume(f, r, d, "rpc"))
                torch.cuda.synchronize()
                got[mode] = outs
        except Exception as e:                                            # noqa: BLE001
            errs.append(e)
            barrier.abort()
    ts = [threading.Thread(target=worker, args=(m,)) for m in ("exact", "fused")]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for mode in ("exact", "fused"):
        assert all(torch.equal(o, want[mode][0]) for o in got[mode]), mode
    # both bits at once are refused
    bad = torch.empty_like(want["exact"][0])
    with pytest.raises(_lib.SatMVSNativeError):
        _lib.call("smvs_rpc_costvol_fwd", _lib.ptr(f[0]), _lib.ptr_array(f[1:]), 2, _lib.ptr(r), _lib.ptr(d), 1 | 0x300, _lib.ptr(bad),
                  1, 16, 8, 48, 96, 0, 8, 8, 0, _lib.current_stream(dev))


def test_models_carry_their_own_arithmetic(dev):
    """Two networks in one process, one built with arith="exact" and one with arith="fused", same weights and inputs: each forward
    equals the forward of the same network inside an arith_scope of that name -- whatever the process default is.  A network
    built without arith= and called outside any scope runs "exact" (round 6: cascades and plane pipelines do not follow the
    process default of the stand-alone builds)."""
    from satmvs_amd import _lib
    from satmvs_amd.networks import casmvs, casred
    import test_full_size_cascade as FS
    H, W = 64, 128
    from satmvs_amd import rpc_synth
    g = torch.Generator(device="cpu").manual_seed(9)
    imgs = torch.nn.functional.avg_pool2d(torch.randn((3, 3, H + 4, W + 4), generator=g), 5, stride=1).contiguous()[None].to(dev)
    rpc = rpc_synth.make_view_rpcs(3, H, W, seed=9)[None]
    pm = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev), "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev),
          "stage3": torch.from_numpy(rpc).to(dev)}
    dv = torch.tensor([[0.0, 400.0]], device=dev)
    for cls, kw in ((casred.Infer_CascadeREDNet, dict(ndepths=[16, 8, 8])), (casmvs.CascadeMVSNet, dict(ndepths=[16, 8, 8]))):
        nets = {}
        for mode in ("exact", "fused"):
            torch.manual_seed(10)
            nets[mode] = cls("rpc", arith=mode, **kw).to(dev).eval()
            FS.randomise_batchnorm(nets[mode], 11)
        torch.manual_seed(10)
        plain = cls("rpc", **kw).to(dev).eval()
        FS.randomise_batchnorm(plain, 11)
        with torch.no_grad():
            want = {}
            for mode in ("exact", "fused"):
                with _lib.arith_scope(mode):
                    want[mode] = plain(imgs, pm, dv)["depth"].clone()
            assert not torch.equal(want["exact"], want["fused"])
            for default in ("exact", "fused"):
                _lib.set_arith(default)
                for mode in ("fused", "exact"):
                    assert torch.equal(nets[mode](imgs, pm, dv)["depth"], want[mode]), (cls.__name__, default, mode)
                assert torch.equal(plain(imgs, pm, dv)["depth"], want["exact"]), (cls.__name__, default, "no scope -> exact")
        _lib.set_arith("exact")


def test_fused_generated_heights_equal_materialised(dev, fused):
    """The *_gen entry points (hypotheses evaluated in the kernel) against the same volume built from the materialised
    hypotheses, in the default mode: identical bits."""
    from satmvs_amd.modules import warping
    from satmvs_amd.modules.depth_range import GeneratedHeights
    feats, rpc, _ = T._inputs(1, 3, 16, 8, 48, 96, seed=21)
    f = [T._t(x, dev) for x in feats]
    rng = np.random.default_rng(3)
    prev = T._t((200.0 + rng.normal(0, 5.0, (1, 24, 48))).astype(np.float32), dev)
    gen = GeneratedHeights(prev, 8, 5.0, (96, 192), (48, 96))
    got = warping.variance_cost_volume(f, T._t(rpc, dev), gen, "rpc")
    want = warping.variance_cost_volume(f, T._t(rpc, dev), gen.materialize(), "rpc")
    assert torch.equal(got, want)


def test_fused_fuzz(dev):
    cmd = [sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_costvol_fwd.py"), "60", "11", "fused"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "MISMATCH" not in p.stdout, p.stdout[-3000:]
    last = p.stdout.strip().splitlines()[-1]
    assert last.startswith("60 cases (fused): 0 voxels outside"), last
