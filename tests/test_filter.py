"""Geometric-consistency filter (/root/reference/tools/rpc_filter.py:11-112).

Pin status: the projector halves are pinned by tests/golden/rpc_project.npz; cv2.remap is NOT (cv2 and cupy are absent
from the build image, so tools/rpc_filter.py cannot be imported): the oracle restates OpenCV's published fixed-point
bilinear remap (oracle/oracle.py::remap_linear_const) and these tests check (CPU) that the oracle behaves like the
reference's docstring promises on a consistent scene, and (GPU) that the kernel equals the oracle."""
import numpy as np
import pytest
import torch


def _scene(H=96, W=160, V=3, seed=0):
    """V consistent height maps of one smooth surface, one per view, through our RPC synthesiser (inputs only)."""
    from satmvs_amd import rpc_synth
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
    return np.stack(depths), rpc


def test_oracle_filter_on_a_consistent_scene(oracle):
    depths, rpc = _scene()
    mask, dep, xs, ys = oracle.check_geometric_consistency(depths[0], rpc[0], depths[1], rpc[1], 1.0, 2.5)
    inner = mask[8:-8, 16:-16]
    assert inner.mean() > 0.99                                   # consistent maps reproject onto themselves
    assert np.abs(dep[mask] - depths[0][mask]).max() < 0.5
    bad = depths.copy()
    bad[1] += 10.0                                               # a 10 m blunder in the source map fails the 2.5 m test
    m2, d2, _, _ = oracle.check_geometric_consistency(bad[0], rpc[0], bad[1], rpc[1], 1.0, 2.5)
    assert m2.mean() < 0.01 and (d2[~m2] == 0).all()
    final, avg = oracle.filter_depth(depths, rpc, 1.0, 2.5, 2)
    assert final[8:-8, 16:-16].mean() > 0.99 and np.abs(avg - depths[0])[final].max() < 0.5
    # remap: integer coordinates return the pixel, the border value appears outside
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert oracle.remap_linear_const(img, np.array([[1.0]]), np.array([[2.0]]))[0, 0] == img[2, 1]
    assert oracle.remap_linear_const(img, np.array([[-5.0]]), np.array([[0.0]]))[0, 0] == -999.0
    assert oracle.remap_linear_const(img, np.array([[1.5]]), np.array([[0.0]]))[0, 0] == 1.5


@pytest.mark.gpu
def test_filter_kernel_matches_oracle(oracle):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from satmvs_amd import rpc_filter
    depths, rpc = _scene(seed=3)
    depths[2, 10:20, 30:50] += 8.0                               # an inconsistent patch in one source view
    for v in (1, 2):
        dep_o, xb_o, yb_o, xs_o, ys_o = oracle.reproject_with_depth(depths[0], rpc[0], depths[v], rpc[v])
        dep, xb, yb, xs, ys = rpc_filter.reproject_with_depth(depths[0], rpc[0], depths[v], rpc[v])
        np.testing.assert_allclose(xs, xs_o, rtol=0, atol=1e-8)
        np.testing.assert_allclose(ys, ys_o, rtol=0, atol=1e-8)
        # the remap rounds coordinates to 1/32 px: a coordinate within 1e-8 px of a rounding boundary may pick the
        # neighbouring fraction -- allow 1e-3 of the pixels to differ there
        close = np.abs(dep - dep_o) <= 1e-4
        assert close.mean() >= 0.999
        ok = close & (dep_o > -900)
        np.testing.assert_allclose(xb[ok], xb_o[ok], rtol=0, atol=1e-6)
        np.testing.assert_allclose(yb[ok], yb_o[ok], rtol=0, atol=1e-6)
        m_o, d_o, _, _ = oracle.check_geometric_consistency(depths[0], rpc[0], depths[v], rpc[v], 1.0, 2.5)
        m, d, _, _ = rpc_filter.check_geometric_consistency(depths[0], rpc[0], depths[v], rpc[v], 1.0, 2.5)
        assert (m != m_o).mean() <= 1e-3
        same = m == m_o
        np.testing.assert_allclose(d[same], d_o[same], rtol=0, atol=1e-4)
    f_o, a_o = oracle.filter_depth(depths, rpc, 1.0, 2.5, 2)
    f, a = rpc_filter.filter_depth(depths, rpc, 1.0, 2.5, 2)
    assert (f != f_o).mean() <= 1e-3 and a.dtype == a_o.dtype
    same = f == f_o
    np.testing.assert_allclose(a[same], a_o[same], rtol=0, atol=1e-3)
    assert not f[10:20, 30:50].any() and f[30:60, 60:120].all()   # the blunder is rejected, the rest accepted
