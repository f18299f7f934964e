"""Geometric-consistency filter (/root/reference/tools/rpc_filter.py:11-112).

Pin status: tests/golden/filter.npz holds the outputs of the reference's OWN rpc_filter.py (reproject_with_depth,
check_geometric_consistency, filter_depth) on top of its own cupy projector rpc_tensor.py, run unmodified by
tests/golden/gen_golden.py::gen_filter with two import-time stand-ins for modules the image lacks: cupy -> numpy, and
cv2.remap -> the oracle's restatement of OpenCV's published fixed-point bilinear remap.  So the projector halves, the
control flow, the thresholds, the masking and the averaging are pinned by a reference run; cv2.remap itself is the one
step that is not (opencv-python 4.5.5.62 is absent here).  CPU: oracle vs that fixture.  GPU: kernel vs that fixture.

Tolerances: float64 coordinates 1e-8 px (the reference contracts quaternary-cubic tensors, the oracle and the kernel sum
20 products / run Horner chains); the remap rounds coordinates to 1/32 px, so a coordinate within 1e-8 px of a rounding
boundary may pick the neighbouring fraction: <= 1e-3 of the pixels may differ there, the rest agree to 1e-4 m."""
import numpy as np
import pytest
import torch


def _check_pair(g, v, dep, xb, yb, xs, ys, m, dm):
    np.testing.assert_allclose(xs, g["v%d.x_src" % v], rtol=0, atol=1e-8)
    np.testing.assert_allclose(ys, g["v%d.y_src" % v], rtol=0, atol=1e-8)
    want = g["v%d.sampled" % v]
    assert dep.dtype == want.dtype and dep.shape == want.shape
    close = np.abs(dep - want) <= 1e-4
    assert close.mean() >= 0.999
    ok = close & (want > -900)
    np.testing.assert_allclose(xb[ok], g["v%d.x_back" % v][ok], rtol=0, atol=1e-6)
    np.testing.assert_allclose(yb[ok], g["v%d.y_back" % v][ok], rtol=0, atol=1e-6)
    wm = g["v%d.mask" % v]
    assert m.dtype == wm.dtype and (m != wm).mean() <= 1e-3
    same = m == wm
    np.testing.assert_allclose(dm[same], g["v%d.depth_masked" % v][same], rtol=0, atol=1e-4)
    assert (dm[~m] == 0).all()


def _check_filter(g, fd):
    depths, rpc, prob = g["depths"], g["rpc"], g["prob"]
    p, d, n, c = float(g["p_ratio"]), float(g["d_ratio"]), int(g["geo_consist_num"]), float(g["confidence_ratio"])
    for key_m, key_a, kw, nn in (("final_mask", "averaged", dict(prob=prob, confidence_ratio=c), n),
                                 ("final_mask_noprob", "averaged_noprob", {}, 1)):
        f, a = fd(depths, rpc, p, d, nn, **kw)
        assert f.dtype == g[key_m].dtype and a.dtype == g[key_a].dtype
        assert (f != g[key_m]).mean() <= 1e-3
        np.testing.assert_allclose(a[f & g[key_m]], g[key_a][f & g[key_m]], rtol=0, atol=1e-3)
    f, _ = fd(depths, rpc, p, d, n, prob=prob, confidence_ratio=c)
    assert not f[10:20, 30:50].any() and not f[:8, :12].any() and f[30:50, 55:90].mean() > 0.5   # blunder and low confidence rejected


def test_oracle_filter_vs_reference(oracle, golden):
    g = golden("filter")
    depths, rpc = g["depths"], g["rpc"]
    for v in (1, 2):
        dep, xb, yb, xs, ys = oracle.reproject_with_depth(depths[0], rpc[0], depths[v], rpc[v])
        m, dm, xs2, ys2 = oracle.check_geometric_consistency(depths[0], rpc[0], depths[v], rpc[v], float(g["p_ratio"]), float(g["d_ratio"]))
        assert np.array_equal(xs, xs2) and np.array_equal(ys, ys2)
        _check_pair(g, v, dep, xb, yb, xs, ys, m, dm)
    _check_filter(g, oracle.filter_depth)
    # remap: integer coordinates return the pixel, the border value appears outside, halves interpolate
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert oracle.remap_linear_const(img, np.array([[1.0]]), np.array([[2.0]]))[0, 0] == img[2, 1]
    assert oracle.remap_linear_const(img, np.array([[-5.0]]), np.array([[0.0]]))[0, 0] == -999.0
    assert oracle.remap_linear_const(img, np.array([[1.5]]), np.array([[0.0]]))[0, 0] == 1.5


@pytest.mark.gpu
def test_filter_kernel_matches_reference(oracle, golden):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from satmvs_amd import rpc_filter
    g = golden("filter")
    depths, rpc = g["depths"], g["rpc"]
    for v in (1, 2):
        dep, xb, yb, xs, ys = rpc_filter.reproject_with_depth(depths[0], rpc[0], depths[v], rpc[v])
        m, dm, xs2, ys2 = rpc_filter.check_geometric_consistency(depths[0], rpc[0], depths[v], rpc[v], float(g["p_ratio"]), float(g["d_ratio"]))
        assert np.array_equal(xs, xs2) and np.array_equal(ys, ys2)
        _check_pair(g, v, dep, xb, yb, xs, ys, m, dm)
        # kernel == oracle on the same inputs, coordinate for coordinate
        od, oxb, oyb, oxs, oys = oracle.reproject_with_depth(depths[0], rpc[0], depths[v], rpc[v])
        np.testing.assert_allclose(xs, oxs, rtol=0, atol=1e-8)
        assert (np.abs(dep - od) <= 1e-4).mean() >= 0.999
    _check_filter(g, rpc_filter.filter_depth)


@pytest.mark.gpu
def test_reproject_returns_raw_samples_for_non_finite_heights(golden):
    """rpc_filter.py:30-47 hands back cv2.remap's value untouched: NaN source heights / the -999 border must come through
    reproject_with_depth, only check_geometric_consistency zeroes what fails the mask."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from satmvs_amd import rpc_filter
    g = golden("filter")
    depths, rpc = g["depths"].copy(), g["rpc"]
    depths[1, 20:30, 40:60] = np.nan
    dep, xb, yb, xs, ys = rpc_filter.reproject_with_depth(depths[0], rpc[0], depths[1], rpc[1])
    assert np.isnan(dep).any() and (dep == -999.0).any() == (g["v1.sampled"] == -999.0).any()
    m, dm, _, _ = rpc_filter.check_geometric_consistency(depths[0], rpc[0], depths[1], rpc[1], 1.0, 2.5)
    assert not np.isnan(dm).any() and not m[np.isnan(dep)].any()


# ---- the pinhole twin: /root/reference/tools/pinhole_filter.py:7-67 ------------------------------------------------------------
# tests/golden/filter_pinhole.npz = the reference's own reproject_with_depth / check_geometric_consistency run unmodified
# (gen_golden.py::gen_filter_pinhole; cv2.remap stood in for by the oracle's restatement, default border 0): pinned except cv2.remap.
# Tolerances: float32 coordinates to 1 ulp-ish (2e-4 px: the float64 matrix products of numpy's BLAS and of the kernel round
# differently in the last bit before the cast); where a coordinate sits on a 1/32-px rounding boundary of the remap the sampled
# depth may come from the neighbouring fraction: <= 1e-3 of the pixels, the rest agree to 1e-3 (depths ~400).
def _check_pinhole_pair(g, v, dep, xb, yb, xs, ys, m, dm):
    for got, key in ((xs, "x_src"), (ys, "y_src")):
        want = g["v%d.%s" % (v, key)]
        assert got.dtype == want.dtype == np.float32 and got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)
    want = g["v%d.sampled" % v]
    close = np.abs(dep - want) <= 1e-3
    assert dep.dtype == want.dtype and close.mean() >= 0.999
    np.testing.assert_allclose(xb[close], g["v%d.x_back" % v][close], rtol=0, atol=5e-4)
    np.testing.assert_allclose(yb[close], g["v%d.y_back" % v][close], rtol=0, atol=5e-4)
    wm = g["v%d.mask" % v]
    assert m.dtype == wm.dtype and (m != wm).mean() <= 2e-3
    same = m == wm
    np.testing.assert_allclose(dm[same], g["v%d.depth_masked" % v][same], rtol=0, atol=1e-3)
    assert (dm[~m] == 0).all()


def test_oracle_pinhole_filter_vs_reference(oracle, golden):
    g = golden("filter_pinhole")
    depths, K, E = g["depths"], g["K"], g["E"]
    for v in (1, 2):
        dep, xb, yb, xs, ys = oracle.pinhole_reproject_with_depth(depths[0], K[0], E[0], depths[v], K[v], E[v])
        m, dm, xs2, ys2 = oracle.pinhole_check_geometric_consistency(depths[0], K[0], E[0], depths[v], K[v], E[v],
                                                                     float(g["p_thre"]), float(g["relative_d_thre"]))
        assert np.array_equal(xs, xs2) and np.array_equal(ys, ys2)
        _check_pinhole_pair(g, v, dep, xb, yb, xs, ys, m, dm)
    assert g["v1.mask"].mean() > 0.8 and g["v2.mask"][12:22, 28:52].mean() < 0.6      # the 3 % blunder patch of the last view is rejected where it matters


@pytest.mark.gpu
def test_pinhole_filter_kernel_matches_reference(oracle, golden):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from satmvs_amd import pinhole_filter
    g = golden("filter_pinhole")
    depths, K, E = g["depths"], g["K"], g["E"]
    for v in (1, 2):
        dep, xb, yb, xs, ys = pinhole_filter.reproject_with_depth(depths[0], K[0], E[0], depths[v], K[v], E[v])
        m, dm, xs2, ys2 = pinhole_filter.check_geometric_consistency(depths[0], K[0], E[0], depths[v], K[v], E[v],
                                                                     float(g["p_thre"]), float(g["relative_d_thre"]))
        assert np.array_equal(xs, xs2) and np.array_equal(ys, ys2)
        _check_pinhole_pair(g, v, dep, xb, yb, xs, ys, m, dm)
        od, oxb, oyb, oxs, oys = oracle.pinhole_reproject_with_depth(depths[0], K[0], E[0], depths[v], K[v], E[v])
        np.testing.assert_allclose(xs, oxs, rtol=0, atol=2e-4)
        assert (np.abs(dep - od) <= 1e-3).mean() >= 0.999
    # the reference's defaults (p_thre=1, relative_d_thre=0.01) and GPU tensors in
    m2, _, _, _ = pinhole_filter.check_geometric_consistency(torch.from_numpy(depths[0]).cuda(), K[0], E[0], torch.from_numpy(depths[1]).cuda(), K[1], E[1])
    assert (m2 != g["v1.mask"]).mean() <= 2e-3
