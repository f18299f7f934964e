"""GPU parity: the HIP path (through the C ABI, via the reference-named Python surface) against the
CPU oracle on identical seeded inputs, against the committed golden vectors of the Python
reference, and -- at the BASELINE.json sizes -- through size-independent properties.

Tolerances (see also tests/test_oracle_golden.py):
  * float64 geodesy: lat/lon 1e-12 deg, pixels 1e-8 px (GPU uses FMA + reciprocal scales).
  * float32 volumes: the sampler/variance arithmetic is bit-identical to the oracle by
    construction; a voxel can differ only when the float64 source coordinate (|diff| ~1e-13 px)
    straddles a float32 rounding boundary.  We require <= 1e-4 of the voxels to differ and
    max |diff| <= 2e-4 for unit-variance features.
  * regressed height: <= 1e-3 m (north_star).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


def _close_f32(got, want, frac=1e-4, atol=2e-4):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    nbad = int((got != want).sum())
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
    assert nbad <= max(1, frac * got.size), "%d of %d voxels differ (max %g)" % (nbad, got.size, diff)
    assert diff <= atol, diff
    return nbad, diff


def _inputs(B, V, C, D, H, W, seed, jitter=True, geo="rpc"):
    from satmvs_amd import rpc_synth
    rng = np.random.default_rng(seed)
    feats = [rng.standard_normal((B, C, H, W)).astype(np.float32) for _ in range(V)]
    if geo == "rpc":
        gp = np.stack([rpc_synth.make_view_rpcs(V, H, W, seed=seed + 7 * b) for b in range(B)])
        lo, hi = 0.0, 400.0
    else:
        gp = np.zeros((B, V, 4, 4))
        for b in range(B):
            for v in range(V):
                f = 1.1 * W
                K = np.array([[f, 0, W / 2.0, 0], [0, f, H / 2.0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1]])
                E = np.eye(4)
                a = rng.normal(0, 0.02) * (v > 0)
                E[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
                E[:3, 3] = [25.0 * v * (-1) ** v, 3.0 * v, 0.5 * v]
                gp[b, v] = K @ E
        lo, hi = 400.0, 700.0
    if jitter:
        depth = (np.linspace(lo, hi, D).reshape(1, D, 1, 1) + rng.normal(0, 2.0, (B, D, H, W))).astype(np.float32)
    else:
        depth = np.linspace(lo, hi, D, dtype=np.float32)[None].repeat(B, 0)
    return feats, gp, depth


# ---------------------------------------------------------------------------------------------------
def test_library_loads_on_gpu(dev):
    from satmvs_amd import _lib
    assert "gfx950" in _lib.version()


def test_cpu_tensors_are_rejected(dev):
    from satmvs_amd import _lib
    from satmvs_amd.modules import warping
    with pytest.raises(_lib.SatMVSNativeError):
        warping.rpc_warping(torch.zeros(1, 2, 4, 4), torch.zeros(1, 170, dtype=torch.float64),
                            torch.zeros(1, 170, dtype=torch.float64), torch.zeros(1, 2), None)


def test_rpc_project_golden(dev, golden, oracle):
    from satmvs_amd.modules import warping
    g = golden("rpc_project")
    hei = _t(g["hei"], dev)[None]
    for v in range(3):
        rpc = _t(g["rpc"][v:v + 1], dev)
        lat, lon = warping.RPC_Photo2Obj(_t(g["samp"], dev)[None], _t(g["line"], dev)[None], hei, rpc, None)
        np.testing.assert_allclose(lat[0].cpu().numpy(), g["lat%d" % v], rtol=0, atol=1e-12)
        np.testing.assert_allclose(lon[0].cpu().numpy(), g["lon%d" % v], rtol=0, atol=1e-12)
        s, l = warping.RPC_Obj2Photo(_t(g["lat%d" % v], dev)[None], _t(g["lon%d" % v], dev)[None], hei, rpc, None)
        np.testing.assert_allclose(s[0].cpu().numpy(), g["samp_back%d" % v], rtol=0, atol=1e-8)
        np.testing.assert_allclose(l[0].cpu().numpy(), g["line_back%d" % v], rtol=0, atol=1e-8)
        olat, olon = oracle.rpc_project(g["rpc"][v], g["samp"], g["line"], g["hei"], 0)
        np.testing.assert_allclose(lat[0].cpu().numpy(), olat, rtol=0, atol=1e-12)


@pytest.mark.parametrize("kind", ["4", "2"])
def test_rpc_warping_golden(dev, golden, oracle, kind):
    from satmvs_amd.modules import warping
    g = golden("rpc_warp")
    out = warping.rpc_warping(_t(g["src_fea"], dev), _t(g["rpc"][:, 1], dev), _t(g["rpc"][:, 0], dev),
                              _t(g["depth" + kind], dev), None)
    _close_f32(out, g["warped" + kind])
    _close_f32(out, oracle.rpc_warping(g["src_fea"], g["rpc"][:, 1], g["rpc"][:, 0], g["depth" + kind]))


def test_rpc_warping_enisum_golden(dev, golden):
    from satmvs_amd.modules import warping
    from satmvs_amd import rpc_synth
    g = golden("rpc_warp_qc")

    def qc(r):
        keys = ["line_off", "samp_off", "lat_off", "lon_off", "height_off", "line_scale", "samp_scale", "lat_scale",
                "lon_scale", "height_scale"]
        d = {k: _t(r[:, i], dev) for i, k in enumerate(keys)}
        for j, nm in enumerate(["line_num", "line_den", "samp_num", "samp_den", "lat_num", "lat_den", "lon_num", "lon_den"]):
            d[nm + "_tensor"] = _t(np.stack([rpc_synth.coeffs_to_qc_tensor(x[10 + 20 * j:30 + 20 * j]) for x in r]), dev)
        return d

    out = warping.rpc_warping_enisum(_t(g["src_fea"], dev), qc(g["rpc"][:, 1]), qc(g["rpc"][:, 0]), _t(g["depth4"], dev))
    _close_f32(out, g["warped"])


@pytest.mark.parametrize("kind", ["4", "2"])
def test_homo_warping_golden(dev, golden, kind):
    from satmvs_amd.modules import warping
    g = golden("homo_warp")
    out = warping.homo_warping(_t(g["src_fea"], dev), _t(g["proj"][:, 1], dev), _t(g["proj"][:, 0], dev),
                               _t(g["depth" + kind], dev))
    _close_f32(out, g["warped" + kind])
    comp = warping._compose_homography(_t(g["proj"][:, 1], dev), _t(g["proj"][:, 0], dev))
    np.testing.assert_allclose(comp.cpu().numpy(), g["composed"], rtol=1e-11, atol=1e-9)


def test_costvol_golden(dev, golden):
    from satmvs_amd.modules import warping
    g = golden("costvol")
    feats = [_t(f, dev) for f in g["feats"]]
    var = warping.variance_cost_volume(feats, _t(g["rpc"], dev), _t(g["depth"], dev), "rpc")
    _close_f32(var, g["variance_rpc"])
    varp = warping.variance_cost_volume(feats, _t(g["proj"], dev), _t(g["depth_pin"], dev), "pinhole")
    _close_f32(varp, g["variance_pin"])


@pytest.mark.parametrize("cfg", [
    dict(B=1, V=3, C=32, D=32, H=128, W=256, jitter=True),     # BASELINE config 1 shape
    dict(B=2, V=5, C=16, D=5, H=33, W=70, jitter=True),        # ragged tile, 5 views, batch 2
    dict(B=1, V=2, C=10, D=9, H=17, W=130, jitter=False),      # generic channel count, (B,D) heights
    dict(B=1, V=8, C=8, D=3, H=8, W=64, jitter=True),          # maximum view count
    dict(B=1, V=3, C=8, D=1, H=4, W=3, jitter=True),           # smaller than one tile
    dict(B=2, V=3, C=16, D=11, H=37, W=70, jitter=True),       # staged kernel: ragged tile, batch 2, odd plane count
    dict(B=1, V=2, C=32, D=6, H=20, W=40, jitter=False),       # staged kernel: one source, (B,D) heights
    dict(B=1, V=3, C=16, D=3, H=5, W=9, jitter=True),          # staged kernel: smaller than one wave patch
    dict(B=1, V=6, C=16, D=5, H=40, W=72, jitter=True),        # staged kernel, 5 sources (2 planes per wave, odd plane count)
    dict(B=2, V=7, C=32, D=4, H=24, W=66, jitter=False),       # staged kernel, 6 sources, batch 2, (B,D) heights, ragged width
    dict(B=1, V=8, C=8, D=1, H=16, W=96, jitter=True),         # staged kernel, 7 sources, single plane
    dict(B=2, V=5, C=16, D=8, H=38, W=70, jitter=True),        # shared-box form (4 sources, 8 planes): ragged tile whose last workgroup has a row pair below the image, batch 2
    dict(B=1, V=4, C=8, D=16, H=24, W=100, jitter=False),      # shared-box form, 3 sources, (B,D) heights, two plane chunks
    dict(B=1, V=5, C=32, D=8, H=6, W=20, jitter=True),         # shared-box form, smaller than one workgroup patch
])
def test_costvol_vs_oracle(dev, oracle, cfg):
    from satmvs_amd.modules import warping
    feats, rpc, depth = _inputs(cfg["B"], cfg["V"], cfg["C"], cfg["D"], cfg["H"], cfg["W"], seed=3, jitter=cfg["jitter"])
    want = oracle.costvol_variance(feats, rpc, depth, "rpc")
    got = warping.variance_cost_volume([_t(f, dev) for f in feats], _t(rpc, dev), _t(depth, dev), "rpc")
    _close_f32(got, want)


def test_costvol_unaligned_feature_pointers(dev, oracle):
    """Feature maps that are contiguous VIEWS at odd element offsets of a larger allocation (base pointers 4-byte but not
    16-byte aligned) and a width that is not a multiple of 4: the 16-byte LDS-DMA chunks of the staged kernel are then
    unaligned in memory everywhere.  Same bits as the oracle."""
    from satmvs_amd.modules import warping
    B, V, C, D, H, W = 1, 3, 16, 8, 40, 90
    feats, rpc, depth = _inputs(B, V, C, D, H, W, seed=9, jitter=True)
    views = []
    for k, f in enumerate(feats):
        big = torch.zeros(f.size + 7, dtype=torch.float32, device=dev)
        off = 1 + 2 * k                                            # 4, 12, 20 bytes past a 256-byte aligned allocation
        big[off:off + f.size] = _t(f, dev).reshape(-1)
        v = big[off:off + f.size].view(B, C, H, W)
        assert v.is_contiguous() and v.data_ptr() % 16 != 0
        views.append(v)
    want = oracle.costvol_variance(feats, rpc, depth, "rpc")
    got = warping.variance_cost_volume(views, _t(rpc, dev), _t(depth, dev), "rpc")
    _close_f32(got, want)


def test_costvol_pinhole_vs_oracle(dev, oracle):
    from satmvs_amd.modules import warping
    feats, proj, depth = _inputs(1, 3, 16, 12, 48, 96, seed=4, geo="pinhole")
    want = oracle.costvol_variance(feats, proj, depth, "pinhole")
    got = warping.variance_cost_volume([_t(f, dev) for f in feats], _t(proj, dev), _t(depth, dev), "pinhole")
    _close_f32(got, want)


def test_costvol_shared_box_paths_agree(dev, oracle, arith):
    """3-4 sources, sweeps that divide into eights: a workgroup of 2 x 2 waves stages ONE box per source for 4 rows x 8 planes
    (csrc/costvol_kernels.h, shared form).  Plane windows take the shared form (8 planes), the per-wave form (4 / 2 / 1 planes)
    or, with planes too far apart for the box, the direct gathers inside it: identical bits in both arithmetic modes, so
    a sharded build equals the unsharded one; the exact mode also equals the oracle."""
    from satmvs_amd.modules import warping
    feats, rpc, depth = _inputs(1, 5, 16, 24, 42, 72, seed=8)
    depth[0, 5, 7, 9] = np.nan
    depth[0, 17, :4] = 9000.0                                       # off the source images
    f = [_t(x, dev) for x in feats]
    r, d = _t(rpc, dev), _t(depth, dev)
    full = warping.variance_cost_volume(f, r, d, "rpc")
    for lo, hi in ((0, 8), (8, 24), (3, 11), (16, 24), (5, 6), (0, 4), (10, 12), (1, 24)):
        part = warping.variance_cost_volume(f, r, d, "rpc", d_begin=lo, d_end=hi)
        assert torch.equal(torch.nan_to_num(part, nan=-7.0), torch.nan_to_num(full[:, :, lo:hi], nan=-7.0)), (lo, hi)
    if arith == "exact":
        want = oracle.costvol_variance(feats, rpc, depth, "rpc")
        got = full.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want))
        _close_f32(got[~np.isnan(want)], want[~np.isnan(want)])
    # planes 57 m apart: the workgroup's boxes do not fit -> every wave takes the direct gathers for its planes
    wide = np.broadcast_to(np.linspace(0.0, 3000.0, 8, dtype=np.float32).reshape(1, 8, 1, 1), (1, 8, 42, 72)).copy()
    w = _t(wide, dev)
    whole = warping.variance_cost_volume(f, r, w, "rpc")
    for pl in range(8):
        assert torch.equal(warping.variance_cost_volume(f, r, w, "rpc", d_begin=pl, d_end=pl + 1), whole[:, :, pl:pl + 1]), pl


# ---- plane-constant heights: collapsed (bivariate) source cubics, smvs_rpc_plane_coef + smvs_rpc_costvol_fwd_pc -------------
def _build_raw(dev, feats, rpc, depth, use_pc, d_begin=0, d_end=None, pc_from=None):
    """The volume through the C ABI: smvs_rpc_costvol_fwd (use_pc=False) or smvs_rpc_plane_coef + smvs_rpc_costvol_fwd_pc."""
    from satmvs_amd import _lib
    f = [_t(x, dev) for x in feats]
    r, d = _t(rpc, dev), _t(depth, dev)
    B, C, H, W = f[0].shape
    D = d.shape[1]
    is4d = 1 if d.dim() == 4 else 0
    d_end = D if d_end is None else d_end
    out = torch.full((B, C, d_end - d_begin, H, W), 7.0, dtype=torch.float32, device=dev)
    st = _lib.current_stream(dev)
    srcs = _lib.ptr_array(f[1:])
    if not use_pc:
        _lib.call("smvs_rpc_costvol_fwd", _lib.ptr(f[0]), srcs, len(f) - 1, _lib.ptr(r), _lib.ptr(d), is4d, _lib.ptr(out),
                  B, C, D, H, W, d_begin, d_end, d_end - d_begin, 0, st)
        return out, None
    pc = torch.zeros(_lib.load().smvs_rpc_plane_coef_bytes(B, len(f) - 1, D) // 8, dtype=torch.float64, device=dev)
    src_d = d if pc_from is None else _t(pc_from, dev)
    _lib.call("smvs_rpc_plane_coef", _lib.ptr(r), _lib.ptr(src_d), 1 if src_d.dim() == 4 else 0, _lib.ptr(pc), B, len(f) - 1, D, H, W, 0, D, st)
    _lib.call("smvs_rpc_costvol_fwd_pc", _lib.ptr(f[0]), srcs, len(f) - 1, _lib.ptr(r), _lib.ptr(d), is4d, _lib.ptr(pc), _lib.ptr(out),
              B, C, D, H, W, d_begin, d_end, d_end - d_begin, 0, st)
    torch.cuda.synchronize()
    return out, pc


def test_plane_coef_records(dev):
    """smvs_rpc_plane_coef: the planes' heights at b D + d (padded to a multiple of 8 doubles), then per (b, d) and source the
    four cubics SNUM, SDEN, LNUM, LDEN folded at the plane's normalised height (the 6 height-dependent of their 10 bivariate
    coefficients each, laid out [b][source][cubic][d][6]) -- against a float64 numpy evaluation of the same sums (/root/reference/modules/warping.py:183-207
    monomial order)."""
    from satmvs_amd import _lib
    B, V, D, H, W = 2, 4, 5, 12, 20
    _, rpc, depth = _inputs(B, V, 8, D, H, W, seed=21, jitter=False)
    depth4 = np.ascontiguousarray(np.broadcast_to(depth[:, :, None, None], (B, D, H, W)))
    hdr = (B * D + 7) // 8 * 8
    n = _lib.load().smvs_rpc_plane_coef_bytes(B, V - 1, D) // 8
    body, co = B * D * 24 * (V - 1), hdr + 24 * B      # (behind the heights: 24 doubles per batch item for the views' reciprocal scales)
    assert n == co + body + 64          # (+ what a cut-short plane group may read behind the last plane)
    for dd in (depth, depth4):
        pc = torch.zeros(n, dtype=torch.float64, device=dev)
        d = _t(dd, dev)
        _lib.call("smvs_rpc_plane_coef", _lib.ptr(_t(rpc, dev)), _lib.ptr(d), 1 if d.dim() == 4 else 0, _lib.ptr(pc), B, V - 1, D, H, W, 0, D,
                  _lib.current_stream(dev))
        flat = pc.cpu().numpy()
        assert np.array_equal(flat[:B * D].reshape(B, D), depth.astype(np.float64)) and not flat[B * D:hdr].any()
        rec = flat[co:co + body].reshape(B, V - 1, 4, D, 6).transpose(0, 3, 1, 2, 4)          # [b][source][cubic][d][6] -> (b, d, s, i, j)
        sc = flat[hdr:co].reshape(B, 8, 3)
        for b in range(B):
            assert np.array_equal(sc[b, 0], 1.0 / rpc[b, 0][[6, 5, 9]])                 # ref view: 1/SAMP_SCALE, 1/LINE_SCALE, 1/HEIGHT_SCALE (IEEE divisions)
            for s in range(V - 1):
                assert np.array_equal(sc[b, s + 1], 1.0 / rpc[b, s + 1][[7, 8, 9]])     # source views: 1/LAT_SCALE, 1/LONG_SCALE, 1/HEIGHT_SCALE
        for b in range(B):
            for s in range(V - 1):
                r = rpc[b, s + 1]
                Hn = (depth[b].astype(np.float64) - r[4]) / r[9]
                for i, base in enumerate((50, 70, 10, 30)):
                    c = r[base:base + 20]
                    want = np.stack([c[0] + Hn * c[3] + Hn ** 2 * c[9] + Hn ** 3 * c[19], c[1] + Hn * c[5] + Hn ** 2 * c[13],
                                     c[2] + Hn * c[6] + Hn ** 2 * c[16], c[4] + Hn * c[10], c[7] + Hn * c[17], c[8] + Hn * c[18]], axis=1)
                    np.testing.assert_allclose(rec[b, :, s, i], want, rtol=1e-14, atol=1e-18)
    # a plane window writes the records of its planes only
    pc = torch.full_like(pc, -3.0)
    _lib.call("smvs_rpc_plane_coef", _lib.ptr(_t(rpc, dev)), _lib.ptr(_t(depth, dev)), 0, _lib.ptr(pc), B, V - 1, D, H, W, 1, 3, _lib.current_stream(dev))
    w = pc.cpu().numpy()
    wr = w[co:co + body].reshape(B, V - 1, 4, D, 6).transpose(0, 3, 1, 2, 4)
    assert (wr[:, [0, 3, 4]] == -3.0).all() and np.array_equal(wr[:, 1:3], rec[:, 1:3])
    assert np.array_equal(w[:B * D].reshape(B, D)[:, 1:3], depth[:, 1:3].astype(np.float64)) and (w[:B * D].reshape(B, D)[:, [0, 3, 4]] == -3.0).all()


def test_reciprocal_scales_equal_the_ieee_quotients(dev):
    """recip_scale (v_rcp_f64 + two Newton steps: every rpc kernel's 1/SCALE since round 6) against the IEEE quotient the oracle and
    the reference compute, read back through smvs_rpc_plane_coef's header: bit-equal over 77 000 scales spread log-uniformly over
    1e-4 .. 1e5 (camera models hold 0.05 .. 20 000) and over the synthetic scenes' own values."""
    from satmvs_amd import _lib
    from satmvs_amd import rpc_synth
    rng = np.random.default_rng(41)
    B, V = 64, 4
    n = _lib.load().smvs_rpc_plane_coef_bytes(B, V - 1, 1) // 8
    hdr = (B + 7) // 8 * 8
    depth = _t(np.zeros((B, 1), np.float32), dev)
    bad = 0
    base = np.stack([rpc_synth.make_view_rpcs(V, 96, 160, seed=1000 + b) for b in range(B)])
    for it in range(100):
        rpc = base.copy()
        if it:                                                    # round 0: the synthetic scenes' own scales; then arbitrary ones
            rpc[:, :, 5:10] = np.exp(rng.uniform(np.log(1e-4), np.log(1e5), (B, V, 5))) * rng.choice([-1.0, 1.0], (B, V, 5))
        pc = torch.zeros(n, dtype=torch.float64, device=dev)
        _lib.call("smvs_rpc_plane_coef", _lib.ptr(_t(rpc, dev)), _lib.ptr(depth), 0, _lib.ptr(pc), B, V - 1, 1, 96, 160, 0, 1, _lib.current_stream(dev))
        sc = pc.cpu().numpy()[hdr:hdr + 24 * B].reshape(B, 8, 3)[:, :V]
        want = np.concatenate([1.0 / rpc[:, :1][:, :, [6, 5, 9]], 1.0 / rpc[:, 1:][:, :, [7, 8, 9]]], axis=1)
        bad += int((sc != want).sum())
    assert bad == 0, bad


@pytest.mark.parametrize("cfg", [
    dict(B=1, V=3, C=32, D=16, H=40, W=72, four=True),      # the headline instance (8 planes per wave), (B,D,H,W) broadcast planes
    dict(B=2, V=3, C=16, D=11, H=37, W=70, four=False),     # 4 planes per wave, ragged tile, batch 2, odd plane count, (B,D) heights
    dict(B=1, V=5, C=32, D=8, H=24, W=66, four=True),       # shared-box form, 4 sources
    dict(B=1, V=2, C=8, D=1, H=16, W=40, four=False),       # one source, one plane
    dict(B=1, V=7, C=16, D=5, H=20, W=72, four=True),       # 6 sources: 2 planes per wave
    dict(B=1, V=3, C=10, D=6, H=17, W=33, four=True),       # generic channel count: the direct-gather kernel ignores the coefficients
])
def test_plane_coefficients_match_trivariate(dev, oracle, arith, cfg):
    """Plane-constant heights: waves take the collapsed bivariate source cubics (smvs_device.h, o2p_pc_xn).  Same polynomials
    re-associated -> float64 coordinates move by ~1e-13 px, so the volume equals the trivariate build's except where a
    coordinate straddles a float32 rounding boundary (the same allowance as against the oracle), in both arithmetic modes;
    the exact mode equals the oracle."""
    feats, rpc, depth = _inputs(cfg["B"], cfg["V"], cfg["C"], cfg["D"], cfg["H"], cfg["W"], seed=12, jitter=False)
    if cfg["four"]:
        depth = np.ascontiguousarray(np.broadcast_to(depth[:, :, None, None], depth.shape + (cfg["H"], cfg["W"])))
    tri, _ = _build_raw(dev, feats, rpc, depth, False)
    got, _ = _build_raw(dev, feats, rpc, depth, True)
    _close_f32(got, tri.cpu().numpy())
    if arith == "exact":
        _close_f32(got, oracle.costvol_variance(feats, rpc, depth, "rpc"))
    # plane windows on the same coefficients: bit-identical to the whole sweep
    D = cfg["D"]
    for lo, hi in ((0, 1), (D // 2, D), (1, max(2, D - 1))):
        if hi <= D and lo < hi:
            part, _ = _build_raw(dev, feats, rpc, depth, True, lo, hi)
            assert torch.equal(part, got[:, :, lo:hi]), (lo, hi)


def test_plane_coefficients_are_checked_not_trusted(dev, oracle):
    """The kernel compares every wave's heights with the folded planes' and takes the bivariate cubics only where ALL match:
    (1) one jittered pixel sends its wave down the trivariate chain, and the volume still equals the oracle's; (2) coefficients
    folded for OTHER heights are never used: the result is bit-identical to smvs_rpc_costvol_fwd; (3) per-voxel heights:
    identical bits with and without the workspace."""
    B, V, C, D, H, W = 1, 3, 32, 8, 24, 96
    feats, rpc, planes = _inputs(B, V, C, D, H, W, seed=14, jitter=False)
    depth = np.ascontiguousarray(np.broadcast_to(planes[:, :, None, None], (B, D, H, W))).copy()
    rng = np.random.default_rng(3)
    for _ in range(6):                                             # one jittered voxel in six (wave, plane) places
        depth[0, rng.integers(D), rng.integers(H), rng.integers(W)] += 1.5
    depth[0, 3, 5, 40] = np.nan
    want = oracle.costvol_variance(feats, rpc, depth, "rpc")
    got, _ = _build_raw(dev, feats, rpc, depth, True)
    got = got.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    _close_f32(got[~np.isnan(want)], want[~np.isnan(want)])
    clean = np.ascontiguousarray(np.broadcast_to(planes[:, :, None, None], (B, D, H, W)))
    tri, _ = _build_raw(dev, feats, rpc, clean, False)
    stale, _ = _build_raw(dev, feats, rpc, clean, True, pc_from=clean + 0.25)
    assert torch.equal(stale, tri)
    jit = _inputs(B, V, C, D, H, W, seed=14, jitter=True)[2]
    a, _ = _build_raw(dev, feats, rpc, jit, False)
    b, _ = _build_raw(dev, feats, rpc, jit, True)
    assert torch.equal(a, b)


def test_variance_cost_volume_plane_constant_policy(dev, arith, monkeypatch):
    """variance_cost_volume sends (B,D) heights and H/W-broadcast views through the folded cubics, materialised (B,D,H,W)
    tensors through the trivariate chain unless told (and nothing below _FOLD_MIN_VOXELS: lifted here); the bits are the
    same whichever way (both modes)."""
    from satmvs_amd.modules import warping
    monkeypatch.setattr(warping, "_FOLD_MIN_VOXELS", 0)
    calls = []
    real = warping.plane_coefficients
    monkeypatch.setattr(warping, "plane_coefficients", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    B, V, C, D, H, W = 1, 3, 32, 8, 24, 96
    feats, rpc, planes = _inputs(B, V, C, D, H, W, seed=15, jitter=False)
    f, r = [_t(x, dev) for x in feats], _t(rpc, dev)
    p2 = _t(planes, dev)
    view = p2[:, :, None, None].expand(B, D, H, W)
    outs = [warping.variance_cost_volume(f, r, p2, "rpc"), warping.variance_cost_volume(f, r, view, "rpc"),
            warping.variance_cost_volume(f, r, view.contiguous(), "rpc"),
            warping.variance_cost_volume(f, r, view.contiguous(), "rpc", plane_constant=True),
            warping.variance_cost_volume(f, r, p2, "rpc", plane_constant=False)]
    assert len(calls) == 3                                       # (B,D), the broadcast view, the materialised tensor with plane_constant=True
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    jit = _t(_inputs(B, V, C, D, H, W, seed=15, jitter=True)[2], dev)
    assert torch.equal(warping.variance_cost_volume(f, r, jit, "rpc"), warping.variance_cost_volume(f, r, jit, "rpc", plane_constant=True))


@pytest.mark.parametrize("C", [8, 16])                          # direct kernel / staged kernel
def test_costvol_out_of_image_and_nan(dev, oracle, C):
    """Large parallax pushes taps off the source image (zero padding); NaN heights must not fault."""
    from satmvs_amd.modules import warping
    feats, rpc, depth = _inputs(1, 3, C, 6, 32, 64, seed=5)
    depth[:, 0] = -4000.0
    depth[:, 1] = 6000.0
    depth[0, 2, 3, 5] = np.nan
    want = oracle.costvol_variance(feats, rpc, depth, "rpc")
    got = warping.variance_cost_volume([_t(f, dev) for f in feats], _t(rpc, dev), _t(depth, dev), "rpc").cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    m = ~np.isnan(want)
    _close_f32(got[m], want[m])


def test_costvol_plane_ranges(dev):
    """[d_begin,d_end) builds exactly those planes, bit-identical to the whole-volume launch."""
    from satmvs_amd.modules import warping
    feats, rpc, depth = _inputs(1, 3, 16, 20, 40, 72, seed=6)
    f = [_t(x, dev) for x in feats]
    r, d = _t(rpc, dev), _t(depth, dev)
    full = warping.variance_cost_volume(f, r, d, "rpc")
    for lo, hi in ((0, 1), (3, 11), (19, 20), (8, 20)):
        part = warping.variance_cost_volume(f, r, d, "rpc", d_begin=lo, d_end=hi)
        assert part.shape[2] == hi - lo
        assert torch.equal(part, full[:, :, lo:hi])


def test_regression_golden(dev, golden, oracle):
    from satmvs_amd.modules import module as M
    g = golden("regress")
    with torch.no_grad():
        depth, conf = M.softmax_depth_regression(_t(g["reg"], dev), _t(g["depth_values"], dev))
    np.testing.assert_allclose(depth.cpu().numpy(), g["sm_depth"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(conf.cpu().numpy(), g["sm_conf"], rtol=1e-5, atol=1e-6)
    B, D, H, W = g["reg"].shape
    acc = M.StreamingRegression(B, H, W, dev)
    for d in range(D):
        acc.step(_t(g["reg"][:, d], dev), _t(g["depth_values"], dev), d)
    st = acc.state.cpu().numpy()
    np.testing.assert_allclose(st[0], g["st_exp_sum"][:, 0], rtol=1e-13)
    np.testing.assert_allclose(st[1], g["st_depth_img"][:, 0], rtol=1e-12)
    np.testing.assert_allclose(st[2], g["st_max"][:, 0], rtol=1e-13)
    depth, conf = acc.result()
    np.testing.assert_allclose(depth.cpu().numpy(), g["st_depth"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(conf.cpu().numpy(), g["st_conf"], rtol=1e-6)
    c = golden("costvol")
    with torch.no_grad():
        depth, conf = M.softmax_depth_regression(_t(c["reg_rpc"], dev), _t(c["depth"], dev))
    np.testing.assert_allclose(depth.cpu().numpy(), c["depth_rpc"], rtol=0, atol=1e-3)


def test_window_regression_golden(dev, golden, oracle):
    """smvs_window_regress_fwd (casmvs / ucs flavour) vs the reference's outputs and vs the oracle, and the torch
    composite of the same function (what runs under autograd) vs the kernel."""
    from satmvs_amd.modules import module as M
    g = golden("regress")
    reg, dv, lamb = _t(g["reg"], dev), _t(g["depth_values"], dev), float(g["ucs_lamb"])
    with torch.no_grad():
        depth, conf, var = M.window_depth_regression(reg, dv, lamb=lamb)
        d2, c2 = M.window_depth_regression(reg, dv)
    assert torch.equal(d2, depth) and torch.equal(c2, conf)
    np.testing.assert_allclose(depth.cpu().numpy(), g["sm_depth"], rtol=0, atol=1e-3)
    assert (np.abs(conf.cpu().numpy() - g["w4_conf"]) > 1e-5).mean() <= 0.01
    np.testing.assert_allclose(var.cpu().numpy(), g["ucs_variance"], rtol=2e-5, atol=1e-3)
    od, oc, ov = oracle.window_regress(g["reg"], g["depth_values"], lamb=lamb)
    np.testing.assert_allclose(depth.cpu().numpy(), od, rtol=0, atol=1e-4)
    assert (np.abs(conf.cpu().numpy() - oc) > 1e-6).mean() <= 0.005
    np.testing.assert_allclose(var.cpu().numpy(), ov, rtol=1e-5, atol=1e-4)
    # the differentiable composite (autograd path) computes the same numbers
    rg = reg.clone().requires_grad_(True)
    dc, cc, vc = M.window_depth_regression(rg, dv, lamb=lamb)
    np.testing.assert_allclose(dc.detach().cpu().numpy(), depth.cpu().numpy(), rtol=0, atol=1e-3)
    assert (np.abs(cc.detach().cpu().numpy() - conf.cpu().numpy()) > 1e-5).mean() <= 0.01
    np.testing.assert_allclose(vc.detach().cpu().numpy(), var.cpu().numpy(), rtol=1e-4, atol=1e-3)
    # (B,D) heights
    planes = torch.linspace(0, 300, reg.shape[1], device=dev).repeat(reg.shape[0], 1)
    with torch.no_grad():
        dp, cp, vp = M.window_depth_regression(reg, planes, lamb=lamb)
    od, oc, ov = oracle.window_regress(g["reg"], planes.cpu().numpy(), lamb=lamb)
    np.testing.assert_allclose(dp.cpu().numpy(), od, rtol=0, atol=1e-4)
    np.testing.assert_allclose(vp.cpu().numpy(), ov, rtol=1e-5, atol=1e-4)


def test_generated_heights(dev, golden, oracle):
    """hypotheses evaluated inside the kernels.  (1) smvs_height_hypotheses == the oracle bit for bit and
    the reference's (B,D,H,W) tensors (tests/golden/depth_range.npz: r2 exact -- sample arithmetic + trilinear resize --,
    r_a / r_b within 1 ulp -- bilinear resize of the previous map, see tests/test_oracle_golden.py); (2) every consumer
    fed the generator produces the bits it produces from the materialised tensor: cost volume (staged and direct
    kernels, plane ranges), softmax and window regressions; (3) stage-1 planes equal the reference's r1."""
    from satmvs_amd.modules import module as M
    from satmvs_amd.modules import warping
    from satmvs_amd.modules.depth_range import GeneratedHeights, stage1_planes
    g = golden("depth_range")
    _, H, W = g["cur"].shape
    cases = [("prev_a", 6, 5.0, (H // 2, W // 2), "r_a", 6.2e-5), ("prev_b", 8, 2.5, (H, W), "r_b", 6.2e-5),
             ("cur", 6, 5.0, (H // 2, W // 2), "r2", 0.0)]
    for name, nd, interval, stage, ref_key, tol in cases:
        gen = GeneratedHeights(_t(g[name], dev), nd, interval, (H, W), stage)
        got = gen.materialize().cpu().numpy()
        assert np.array_equal(got, oracle.height_hypotheses(g[name], nd, interval, (H, W), stage)), name
        assert np.abs(got - g[ref_key]).max() <= tol, name
    planes = stage1_planes(_t(g["dv"], dev), 8).cpu().numpy()
    assert np.array_equal(np.broadcast_to(planes[:, :, None, None], g["r1"].shape), g["r1"])
    # consumers: generator vs materialised tensor, identical bits
    rng = np.random.default_rng(21)
    for C, (h, w), up, nd, interval in ((16, (40, 72), 2, 6, 5.0), (8, (40, 72), 1, 8, 2.5), (32, (24, 40), 2, 5, 5.0)):
        ih, iw = h * up, w * up
        feats, rpc, _ = _inputs(1, 3, C, nd, h, w, seed=9)
        prev = (200.0 + rng.normal(0, 6.0, (1, ih // 2, iw // 2))).astype(np.float32)
        gen = GeneratedHeights(_t(prev, dev), nd, interval, (ih, iw), (h, w))
        dv = gen.materialize()
        f = [_t(x, dev) for x in feats]
        r = _t(rpc, dev)
        with torch.no_grad():
            a = warping.variance_cost_volume(f, r, gen, "rpc")
            b = warping.variance_cost_volume(f, r, dv, "rpc")
            assert torch.equal(a, b), (C, h, w)
            a1 = warping.variance_cost_volume(f, r, gen, "rpc", d_begin=2, d_end=3)
            assert torch.equal(a1, b[:, :, 2:3])
            reg = -a.mean(1)
            assert all(torch.equal(x, y) for x, y in zip(M.softmax_depth_regression(reg, gen), M.softmax_depth_regression(reg, dv)))
            assert all(torch.equal(x, y) for x, y in zip(M.window_depth_regression(reg, gen, lamb=1.5),
                                                           M.window_depth_regression(reg, dv, lamb=1.5)))
        want = oracle.costvol_variance(feats, rpc, dv.cpu().numpy(), "rpc")
        _close_f32(a, want)


def test_generated_heights_ucs(dev, golden, oracle):
    """SURVEY 8f-1, UCS-Net flavour: uncertainty_aware_samples behind its two bilinear resizes evaluated inside the kernels.
    (1) smvs_height_hypotheses == the oracle bit for bit and the reference's tensors within 1 ulp of the resize
    (tests/golden/ucs_samples.npz, both range clamps firing); (2) the cost volume and the window regression fed the generator
    give the bits they give from the materialised tensor."""
    from satmvs_amd.modules import module as M
    from satmvs_amd.modules import warping
    from satmvs_amd.modules.depth_range import GeneratedHeights
    g = golden("ucs_samples")
    H, W = g["s8"].shape[2:]
    for key, nd in (("s8", 8), ("s12", 12)):
        gen = GeneratedHeights.ucs(_t(g["prev"], dev), _t(g["var"], dev), _t(g["dmin"], dev), _t(g["dmax"], dev), nd, (H, W))
        got = gen.materialize().cpu().numpy()
        assert np.array_equal(got, oracle.ucs_hypotheses(g["prev"], g["var"], g["dmin"], g["dmax"], nd, (H, W))), key
        assert np.abs(got - g[key]).max() <= 6.2e-5, key
    rng = np.random.default_rng(22)
    for C, (h, w), nd in ((16, (40, 72), 6), (8, (48, 80), 8)):
        feats, rpc, _ = _inputs(1, 3, C, nd, h, w, seed=10)
        prev = (200.0 + rng.normal(0, 6.0, (1, h // 2, w // 2))).astype(np.float32)
        var = rng.uniform(0.5, 12.0, (1, h // 2, w // 2)).astype(np.float32)
        lo, hi = np.array([190.0], np.float32), np.array([215.0], np.float32)
        gen = GeneratedHeights.ucs(_t(prev, dev), _t(var, dev), _t(lo, dev), _t(hi, dev), nd, (h, w))
        dv = gen.materialize()
        f = [_t(x, dev) for x in feats]
        r = _t(rpc, dev)
        with torch.no_grad():
            a = warping.variance_cost_volume(f, r, gen, "rpc")
            b = warping.variance_cost_volume(f, r, dv, "rpc")
            assert torch.equal(a, b), (C, h, w)
            reg = -a.mean(1)
            assert all(torch.equal(x, y) for x, y in zip(M.window_depth_regression(reg, gen, lamb=1.5),
                                                           M.window_depth_regression(reg, dv, lamb=1.5)))


def test_photo_consistent_peaky_problem(dev, golden, oracle, arith):
    """Photo-consistent features + peaky regulariser (gen_golden.py::gen_photo): heights within 1e-3 m of the
    reference's; and the check means something -- moving one source image by 0.05 px moves the heights by far more."""
    from satmvs_amd.modules import module as M
    from satmvs_amd.modules import warping
    from satmvs_amd.rpc_synth import SAMP_OFF
    g = golden("photo")
    feats = [_t(g["feats"][v], dev) for v in range(g["feats"].shape[0])]
    dv, lam = _t(g["depth_values"], dev), float(g["lam"])
    with torch.no_grad():
        var = warping.variance_cost_volume(feats, _t(g["rpc"], dev), dv, "rpc")
        depth, conf = M.softmax_depth_regression(-lam * var.mean(1), dv)
        rpc2 = g["rpc"].copy()
        rpc2[0, 1, SAMP_OFF] += 0.05
        var2 = warping.variance_cost_volume(feats, _t(rpc2, dev), dv, "rpc")
        depth2, _ = M.softmax_depth_regression(-lam * var2.mean(1), dv)
    if arith == "exact":
        assert np.abs(depth.cpu().numpy() - g["depth"]).max() <= 1e-3
    else:
        # lam = 2e6 amplifies the ROUNDING NOISE of the variance into the height: the reference's own golden sits 1.16e-3 m from
        # a float64 evaluation of its formula on the same float32 taps (oracle.costvol_variance_f64 -> softmax -> expectation,
        # all float64).  The fused arithmetic does not reproduce that noise; it is held to 1e-3 m (measured 2.4e-4) of the
        # float64 evaluation, and lands as far from the golden as the golden is from float64.
        truth, _ = oracle.costvol_variance_f64([g["feats"][v] for v in range(g["feats"].shape[0])], g["rpc"], g["depth_values"], "rpc")
        reg = -lam * truth.mean(1)
        p = np.exp(reg - reg.max(1, keepdims=True))
        h64 = (p / p.sum(1, keepdims=True) * g["depth_values"].astype(np.float64)).sum(1)
        ref_noise = np.abs(g["depth"] - h64).max()
        assert 1e-3 < ref_noise < 1.3e-3                        # the golden's own distance from float64
        assert np.abs(depth.cpu().numpy() - h64).max() <= 1e-3 / 2
        assert np.abs(depth.cpu().numpy() - g["depth"]).max() <= ref_noise + 1e-3 / 2
    np.testing.assert_allclose(conf.cpu().numpy(), g["conf"], rtol=0, atol=5e-4)
    assert float((depth2 - depth).abs().max()) > 0.05          # 50x the tolerance: the golden is sensitive to the warp


def test_costvol_backward_matches_reference_gradients(dev, golden, arith):
    """d loss / d features through the native forward + smvs_costvol_bwd against gradients captured from the
    reference's own differentiable path (gen_golden.py::gen_grad: grid_sample backward, in-place variance
    accumulation, softmax, depth_regression).  float32 atomics sum in another order: rtol 2e-4 on the gradient scale."""
    from satmvs_amd.modules.module import depth_regression
    from satmvs_amd.modules import warping
    g = golden("grad")
    feats = [_t(g["feats"][v], dev).requires_grad_(True) for v in range(g["feats"].shape[0])]
    dv, lam, w = _t(g["depth_values"], dev), float(g["lam"]), _t(g["wmap"], dev)
    var = warping.variance_cost_volume(feats, _t(g["rpc"], dev), dv, "rpc")
    p = torch.softmax(-lam * var.mean(1), dim=1)
    depth = depth_regression(p, depth_values=dv)
    np.testing.assert_allclose(depth.detach().cpu().numpy(), g["depth"], rtol=0, atol=1e-3)
    loss = (depth * w).sum()
    assert abs(loss.item() - float(g["loss"])) <= 1e-3 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    scale = np.abs(g["grads"]).max()
    for v, f in enumerate(feats):
        np.testing.assert_allclose(f.grad.cpu().numpy(), g["grads"][v], rtol=0, atol=2e-4 * scale)


def test_warp_backward_matches_torch(dev, oracle):
    """grad w.r.t. src_fea == autograd of F.grid_sample on the same (oracle-built) grid."""
    from satmvs_amd.modules import warping
    B, C, D, H, W = 1, 4, 3, 16, 24
    feats, rpc, depth = _inputs(B, 2, C, D, H, W, seed=8)
    src = _t(feats[1], dev).requires_grad_(True)
    out = warping.rpc_warping(src, _t(rpc[:, 1], dev), _t(rpc[:, 0], dev), _t(depth, dev), None)
    gout = torch.randn_like(out)
    out.backward(gout)
    _, _, samp, line = oracle.rpc_warp_coords(rpc[:, 1], rpc[:, 0], depth, H, W)
    gx = samp.astype(np.float32) / np.float32((W - 1) / 2.0) - np.float32(1)
    gy = line.astype(np.float32) / np.float32((H - 1) / 2.0) - np.float32(1)
    grid = _t(np.stack([gx, gy], -1).reshape(B, D * H, W, 2), dev)
    src2 = _t(feats[1], dev).requires_grad_(True)
    ref = torch.nn.functional.grid_sample(src2, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    ref.view(B, C, D, H, W).backward(gout)
    torch.testing.assert_close(src.grad, src2.grad, rtol=1e-4, atol=1e-4)


def test_costvol_backward_matches_torch(dev, oracle):
    """Backward of the fused volume == autograd through the reference's composite
    (grid_sample per source, sum / sq accumulation, variance) evaluated with torch on the GPU."""
    from satmvs_amd.modules import warping
    B, V, C, D, H, W = 1, 3, 4, 3, 16, 24
    feats, rpc, depth = _inputs(B, V, C, D, H, W, seed=9)
    fs = [_t(f, dev).requires_grad_(True) for f in feats]
    var = warping.variance_cost_volume(fs, _t(rpc, dev), _t(depth, dev), "rpc")
    gout = torch.randn_like(var)
    var.backward(gout)
    fs2 = [_t(f, dev).requires_grad_(True) for f in feats]
    vol = fs2[0].unsqueeze(2).repeat(1, 1, D, 1, 1)
    s, q = vol, vol ** 2
    for v in range(1, V):
        _, _, samp, line = oracle.rpc_warp_coords(rpc[:, v], rpc[:, 0], depth, H, W)
        gx = samp.astype(np.float32) / np.float32((W - 1) / 2.0) - np.float32(1)
        gy = line.astype(np.float32) / np.float32((H - 1) / 2.0) - np.float32(1)
        grid = _t(np.stack([gx, gy], -1).reshape(B, D * H, W, 2), dev)
        w = torch.nn.functional.grid_sample(fs2[v], grid, mode="bilinear", padding_mode="zeros",
                                            align_corners=False).view(B, C, D, H, W)
        s = s + w
        q = q + w ** 2
    var2 = q / V - (s / V) ** 2
    torch.testing.assert_close(var, var2, rtol=1e-5, atol=1e-5)
    var2.backward(gout)
    for a, b in zip(fs, fs2):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cfg", [
    dict(B=2, V=3, C=8, D=12, H=40, W=70),                                  # ragged tiles, D not a multiple of the 8-plane chunk, boxed waves
    dict(B=1, V=2, C=4, D=8, H=64, W=96),                                   # one source view
    dict(B=1, V=3, C=6, D=5, H=33, W=130, jitter=False),                    # per-plane heights (B,D), odd sizes
    dict(B=1, V=5, C=4, D=6, H=48, W=80),                                   # four source views: 4-plane chunks, 64 KB of boxes per workgroup
    dict(B=1, V=6, C=4, D=5, H=24, W=72),                                   # five source views: no boxed path, 64 x 1 patches
    dict(B=1, V=8, C=2, D=3, H=16, W=40),                                   # seven source views
    dict(B=1, V=3, C=4, D=8, H=48, W=96, span=(0.0, 30000.0)),              # tens of cells of parallax per plane: boxes overflow, register scheme
    dict(B=1, V=3, C=4, D=6, H=40, W=72, geo="pinhole"),                    # homography volume
    dict(B=2, V=4, C=4, D=9, H=36, W=66, geo="pinhole", jitter=False),
])
def test_costvol_backward_matrix(dev, cfg):
    """smvs_costvol_bwd over view counts, geometries, ragged sizes and both scatter schemes (wave-private LDS boxes /
    register runs) against autograd through the per-view warp operators (their own backward kernel, itself checked against
    torch's grid_sample backward above) and torch arithmetic for the variance.  Summation order differs: rtol 1e-4 on the
    gradient scale."""
    from satmvs_amd.modules import warping
    B, V, C, D, H, W = (cfg[k] for k in "BVCDHW")
    geo = cfg.get("geo", "rpc")
    feats, gp, depth = _inputs(B, V, C, D, H, W, seed=31 + V, jitter=cfg.get("jitter", True), geo=geo)
    if "span" in cfg:
        lo, hi = cfg["span"]
        depth = np.broadcast_to(np.linspace(lo, hi, D, dtype=np.float32).reshape(1, D, 1, 1), (B, D, H, W)).copy()
    gpt, dt = _t(gp, dev), _t(depth, dev)
    fs = [_t(f, dev).requires_grad_(True) for f in feats]
    var = warping.variance_cost_volume(fs, gpt, dt, geo)
    gout = torch.randn_like(var)
    var.backward(gout)
    fs2 = [_t(f, dev).requires_grad_(True) for f in feats]
    s = fs2[0].unsqueeze(2).repeat(1, 1, D, 1, 1)
    q = s ** 2
    for v in range(1, V):
        if geo == "rpc":
            w = warping.rpc_warping(fs2[v], gpt[:, v], gpt[:, 0], dt, None)
        else:
            w = warping.homo_warping(fs2[v], gpt[:, v], gpt[:, 0], dt)
        s = s + w
        q = q + w ** 2
    var2 = q / V - (s / V) ** 2
    torch.testing.assert_close(var, var2, rtol=1e-5, atol=1e-5)
    var2.backward(gout)
    for v, (a, b) in enumerate(zip(fs, fs2)):
        scale = float(b.grad.abs().max())
        assert scale > 0
        err = float((a.grad - b.grad).abs().max())
        assert err <= 1e-4 * scale, "view %d: max error %.3g on a gradient scale of %.3g" % (v, err, scale)


@pytest.mark.parametrize("cfg", [
    dict(V=3, C=32, D=64, H=384, W=768),                                     # config 2 (headline metric shape)
    dict(V=3, C=16, D=32, H=192, W=384, span=(150.0, 305.0)),                # config 3, stage 2
    dict(V=5, C=32, D=8, H=768, W=1536, span=(0.0, 44.4)),                   # config 4: one rank's 8-plane shard, four source views
    dict(V=3, C=32, D=64, H=384, W=768, geo="pinhole"),                      # config 5
])
def test_costvol_backward_full_size_adjoint(dev, cfg):
    """smvs_costvol_bwd at BASELINE.json's full sizes through a size-independent identity.  The variance volume is a
    QUADRATIC function of the feature maps, so its central difference is exact: var(f + d) - var(f - d) = 2 J(f) d for any
    direction d, and therefore  < g, (var(f+d) - var(f-d)) / 2 >  =  < (J^T g)_v, d_v >  for a direction d that moves view v
    only -- the left side needs only the forward kernel (pinned bit for bit by the oracle at these sizes,
    test_full_size_properties), the right side is the backward kernel.  Per view, two directions: the gradient itself
    (right side = |grad_v|^2 up to scale: no cancellation, 1e-4 relative -- any error in the gradient's magnitude shows) and
    a random one (tolerance relative to the root-sum-square of the terms -- errors orthogonal to the gradient show)."""
    from satmvs_amd.modules import warping
    V, C, D, H, W = cfg["V"], cfg["C"], cfg["D"], cfg["H"], cfg["W"]
    geo = cfg.get("geo", "rpc")
    feats, gp, depth = _inputs(1, V, C, D, H, W, seed=17, geo=geo)
    if "span" in cfg:
        lo, hi = cfg["span"]
        rng = np.random.default_rng(18)
        depth = (np.linspace(lo, hi, D).reshape(1, D, 1, 1) + rng.normal(0, 0.5, (1, D, H, W))).astype(np.float32)
    gpt, dt = _t(gp, dev), _t(depth, dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    fs = [_t(f, dev).requires_grad_(True) for f in feats]
    var = warping.variance_cost_volume(fs, gpt, dt, geo)
    g = torch.randn(var.shape, generator=gen, device=dev)
    var.backward(g)
    del var
    base = [f.detach() for f in fs]

    def forward_side(v, d):
        with torch.no_grad():
            plus = warping.variance_cost_volume([f + d if i == v else f for i, f in enumerate(base)], gpt, dt, geo)
            minus = warping.variance_cost_volume([f - d if i == v else f for i, f in enumerate(base)], gpt, dt, geo)
            plus -= minus
            del minus
            return 0.5 * torch.dot(plus.double().flatten(), g.double().flatten()).item()

    for v in range(V):
        grad = fs[v].grad
        assert torch.isfinite(grad).all()
        d_grad = grad / grad.pow(2).mean().sqrt()                       # unit rms, like the features
        rhs = torch.dot(grad.double().flatten(), d_grad.double().flatten()).item()
        lhs = forward_side(v, d_grad)
        assert rhs > 0 and abs(lhs - rhs) <= 1e-4 * rhs, (v, lhs, rhs)
        d_rand = torch.randn(grad.shape, generator=gen, device=dev)
        terms = grad.double() * d_rand.double()
        rhs, rss = terms.sum().item(), terms.pow(2).sum().sqrt().item()
        lhs = forward_side(v, d_rand)
        assert abs(lhs - rhs) <= 1e-3 * rss, (v, lhs, rhs, rss)


# ---- BASELINE.json full sizes: properties + oracle spot checks ------------------------------------
@pytest.mark.parametrize("cfg", [
    dict(V=3, C=32, D=64, H=384, W=768, planes=(0, 31, 63)),      # config 2 (headline metric shape)
    dict(V=5, C=32, D=8, H=768, W=1536, planes=(5,)),             # config 4: one GPU's 8-plane shard, planes 50 m apart (boxes overflow: fallback path)
    dict(V=5, C=32, D=8, H=768, W=1536, planes=(2,), span=(0.0, 44.4)),   # config 4 shard at its real spacing: 8 of 64 planes (staged 4-source kernel)
    dict(V=3, C=32, D=48, H=96, W=192, planes=(0, 47)),           # config 3, stage 1: 48 planes on the 1/4-resolution grid
    dict(V=3, C=16, D=32, H=192, W=384, planes=(7,), span=(150.0, 305.0)),    # config 3, stage 2: 32 planes 5 m apart
    dict(V=3, C=8, D=8, H=384, W=768, planes=(0, 7), span=(190.0, 207.5)),    # config 3, stage 3: 8 planes 2.5 m apart, C=8
    dict(V=3, C=32, D=64, H=384, W=768, planes=(0, 40), geo="pinhole"),       # config 5: homography volume at full size
])
def test_full_size_properties(dev, oracle, cfg):
    from satmvs_amd.modules import warping
    V, C, D, H, W = cfg["V"], cfg["C"], cfg["D"], cfg["H"], cfg["W"]
    geo = cfg.get("geo", "rpc")
    feats, rpc, depth = _inputs(1, V, C, D, H, W, seed=11, geo=geo)
    if "span" in cfg:                                          # per-pixel jittered planes over the span a real launch sees
        lo, hi = cfg["span"]
        rng = np.random.default_rng(12)
        depth = (np.linspace(lo, hi, D).reshape(1, D, 1, 1) + rng.normal(0, 0.5, (1, D, H, W))).astype(np.float32)
    f = [_t(x, dev) for x in feats]
    r, d = _t(rpc, dev), _t(depth, dev)
    full = warping.variance_cost_volume(f, r, d, geo)
    assert full.shape == (1, C, D, H, W)
    assert torch.isfinite(full).all()
    # (1) variance is non-negative up to rounding: |min| << typical value
    assert full.min().item() > -1e-4
    # (2) shard consistency: two half-range launches reproduce the whole volume bit for bit, and
    #     the checksum of checksums matches
    a = warping.variance_cost_volume(f, r, d, geo, d_begin=0, d_end=D // 2)
    b = warping.variance_cost_volume(f, r, d, geo, d_begin=D // 2, d_end=D)
    assert torch.equal(torch.cat([a, b], 2), full)
    assert a.double().sum().item() + b.double().sum().item() == pytest.approx(full.double().sum().item(), rel=1e-12)
    # (3) idempotence: same launch twice, identical bits (no atomics / races in the forward)
    assert torch.equal(warping.variance_cost_volume(f, r, d, geo), full)
    # (4) spatially constant feature maps (one constant per channel and view): every in-image
    #     bilinear footprint returns that constant (weights sum to 1), so the volume equals the
    #     across-view variance of the constants wherever all taps are inside the image -- whatever
    #     the geometry.  (Identical views do NOT give zero variance: the reference's sampler is
    #     offset by up to half a pixel, SURVEY.md Q1, and we reproduce that.)
    consts = torch.arange(1, V * C + 1, dtype=torch.float32, device=dev).view(V, C) / 7.0
    cf = [consts[v].view(1, C, 1, 1).expand(1, C, H, W).contiguous() for v in range(V)]
    #     Evaluated on one plane near the cameras' height offset (small parallax), away from the image border.
    mid = torch.full((1, 1, H, W), 200.0 if geo == "rpc" else 550.0, dtype=torch.float32, device=dev)
    z = warping.variance_cost_volume(cf, r, mid, geo)
    want_c = (consts ** 2).mean(0) - consts.mean(0) ** 2
    m = min(64 if geo == "rpc" else 96, H // 4)          # the synthetic pinhole rig shifts view 2 by ~80 px at this depth
    err = (z[0, :, 0, m:-m, m:-m] - want_c.view(C, 1, 1)).abs().max().item()
    assert err < 1e-3 * float(want_c.max()), err
    # (5) oracle spot check on whole planes of the full-size volume
    for pl in cfg["planes"]:
        want = oracle.costvol_variance(feats, rpc, depth, geo, d_begin=pl, d_end=pl + 1)[:, :, pl]
        _close_f32(full[:, :, pl], want)
