"""ctypes front-end for oracle/oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle.c header).  The product package satmvs_amd/ never does.

All functions take/return numpy arrays; layouts follow the reference tensors
(features (B,C,H,W) f32, rpc (B,170)/(B,V,170) f64, depth (B,D) or (B,D,H,W) f32).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile oracle.c with gcc (seconds).  Returns the .so path."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_homo_compose.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def num_threads():
    return int(lib().orc_num_threads())


def rpc_project(rpc170, a, b, h, direction):
    """direction 0: photo->object (samp,line,h)->(lat,lon); 1: object->photo."""
    rpc170, a, b, h = _f64(rpc170), _f64(a).ravel(), _f64(b).ravel(), _f64(h).ravel()
    o0, o1 = np.empty_like(a), np.empty_like(a)
    lib().orc_rpc_project(_p(rpc170), _p(a), _p(b), _p(h), _p(o0), _p(o1), C.c_size_t(a.size), C.c_int(direction))
    return o0, o1


def grid_sample(inp, grid):
    inp, grid = _f32(inp), _f32(grid)
    B, Cc, H, W = inp.shape
    _, Ho, Wo, _ = grid.shape
    out = np.empty((B, Cc, Ho, Wo), np.float32)
    lib().orc_grid_sample(_p(inp), _p(grid), _p(out), B, Cc, H, W, Ho, Wo)
    return out


def _depth_args(depth, B, D, H, W):
    depth = _f32(depth)
    if depth.ndim == 2:
        assert depth.shape == (B, D)
        return depth, 0
    assert depth.shape == (B, D, H, W), (depth.shape, (B, D, H, W))
    return depth, 1


def rpc_warping(src_fea, src_rpc, ref_rpc, depth):
    src_fea, src_rpc, ref_rpc = _f32(src_fea), _f64(src_rpc), _f64(ref_rpc)
    B, Cc, H, W = src_fea.shape
    D = depth.shape[1]
    depth, is4 = _depth_args(depth, B, D, H, W)
    out = np.empty((B, Cc, D, H, W), np.float32)
    lib().orc_rpc_warping(_p(src_fea), _p(src_rpc), _p(ref_rpc), _p(depth), is4, _p(out), B, Cc, D, H, W)
    return out


def rpc_warp_coords(src_rpc, ref_rpc, depth, H, W):
    src_rpc, ref_rpc = _f64(src_rpc), _f64(ref_rpc)
    B = src_rpc.shape[0]
    D = depth.shape[1]
    depth, is4 = _depth_args(depth, B, D, H, W)
    outs = [np.empty((B, D, H, W), np.float64) for _ in range(4)]
    lib().orc_rpc_warp_coords(_p(src_rpc), _p(ref_rpc), _p(depth), is4, *[_p(o) for o in outs], B, D, H, W)
    return outs  # lat, lon, samp, line


def homo_compose(src_proj, ref_proj):
    src_proj, ref_proj = _f64(src_proj), _f64(ref_proj)
    B = src_proj.shape[0]
    out = np.empty((B, 4, 4), np.float64)
    rc = lib().orc_homo_compose(_p(src_proj), _p(ref_proj), _p(out), B)
    if rc:
        raise np.linalg.LinAlgError("singular ref_proj")
    return out


def homo_warping(src_fea, src_proj, ref_proj, depth):
    src_fea = _f32(src_fea)
    B, Cc, H, W = src_fea.shape
    D = depth.shape[1]
    depth, is4 = _depth_args(depth, B, D, H, W)
    proj = homo_compose(src_proj, ref_proj)
    out = np.empty((B, Cc, D, H, W), np.float32)
    lib().orc_homo_warping(_p(src_fea), _p(proj), _p(depth), is4, _p(out), B, Cc, D, H, W)
    return out


def costvol_variance(features, geo_params, depth, geo_model="rpc", d_begin=0, d_end=None, out=None):
    """features: list of V arrays (B,C,H,W), view 0 = reference.
    geo_params: rpc -> (B,V,170) f64;  pinhole -> (B,V,4,4) f64 projection matrices."""
    feats = [_f32(f) for f in features]
    V = len(feats)
    B, Cc, H, W = feats[0].shape
    D = depth.shape[1]
    depth, is4 = _depth_args(depth, B, D, H, W)
    d_end = D if d_end is None else d_end
    if geo_model == "rpc":
        gp = _f64(geo_params)
        assert gp.shape == (B, V, 170)
        geo = 0
    else:
        P = _f64(geo_params)
        assert P.shape == (B, V, 4, 4)
        gp = np.stack([homo_compose(P[:, v], P[:, 0]) for v in range(1, V)], axis=1)  # (B,V-1,4,4)
        gp = _f64(gp)
        geo = 1
    if out is None:
        out = np.zeros((B, Cc, D, H, W), np.float32)
    ptrs = (C.c_void_p * V)(*[f.ctypes.data for f in feats])
    lib().orc_costvol_variance(ptrs, _p(gp), geo, _p(depth), is4, _p(out), B, V, Cc, D, H, W, d_begin, d_end)
    return out


def costvol_variance_f64(features, geo_params, depth, geo_model="rpc"):
    """float64 evaluation of the variance volume from the reference's float32 tap positions (orc_costvol_variance_f64):
    returns (variance, error scale = sum(X^2)/V with X = sum |corner| * weight), both float64 (B,C,D,H,W).  What the float32 sequences approximate; not a reference function."""
    feats = [_f32(f) for f in features]
    V = len(feats)
    B, Cc, H, W = feats[0].shape
    D = depth.shape[1]
    depth, is4 = _depth_args(depth, B, D, H, W)
    if geo_model == "rpc":
        gp = _f64(geo_params)
        geo = 0
    else:
        P = _f64(geo_params)
        gp = _f64(np.stack([homo_compose(P[:, v], P[:, 0]) for v in range(1, V)], axis=1))
        geo = 1
    out = np.zeros((B, Cc, D, H, W), np.float64)
    scale = np.zeros((B, Cc, D, H, W), np.float64)
    ptrs = (C.c_void_p * V)(*[f.ctypes.data for f in feats])
    lib().orc_costvol_variance_f64(ptrs, _p(gp), geo, _p(depth), is4, _p(out), _p(scale), B, V, Cc, D, H, W)
    return out, scale


def softmax_regress(reg, depth):
    reg = _f32(reg)
    B, D, H, W = reg.shape
    depth, is4 = _depth_args(depth, B, D, H, W)
    od = np.empty((B, H, W), np.float32)
    oc = np.empty((B, H, W), np.float32)
    lib().orc_softmax_regress(_p(reg), _p(depth), is4, _p(od), _p(oc), B, D, H, W)
    return od, oc


def height_hypotheses(prev, ndepth, interval, img_hw, stage_hw):
    """(B,D,H,W) hypotheses of a generated cascade stage from the previous height map -- orc_height_hypotheses."""
    import ctypes
    prev = _f32(prev)
    B, hp, wp = prev.shape
    (ih, iw), (H, W) = img_hw, stage_hw
    out = np.empty((B, ndepth, H, W), np.float32)
    f = lib().orc_height_hypotheses
    f.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    f.restype = ctypes.c_int
    if f(_p(prev), B, hp, wp, ih, iw, ndepth, float(interval), H, W, _p(out)) != 0:
        raise ValueError("image / stage size ratio must be 1 or 2")
    return out


def ucs_hypotheses(prev, prev_var, range_min, range_max, ndepth, stage_hw):
    """(B,D,H,W) UCS-Net hypotheses of a later stage from the previous height and standard-deviation maps -- orc_ucs_hypotheses."""
    import ctypes
    prev, prev_var, rmin, rmax = _f32(prev), _f32(prev_var), _f32(range_min), _f32(range_max)
    B, hp, wp = prev.shape
    H, W = stage_hw
    out = np.empty((B, ndepth, H, W), np.float32)
    f = lib().orc_ucs_hypotheses
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    f.restype = ctypes.c_int
    if f(_p(prev), _p(prev_var), _p(rmin), _p(rmax), B, hp, wp, ndepth, H, W, _p(out)) != 0:
        raise ValueError("ndepth must be at least 2")
    return out


def stage1_planes(depth_range, ndepth):
    """(B,D) planes of stage 1 (modules/depth_range.py:26-33): min + arange * (max - min) / (ndepth - 1), float32."""
    dr = _f32(depth_range)
    lo, hi = dr[:, 0], dr[:, -1]
    step = ((hi - lo).astype(np.float32) / np.float32(ndepth - 1)).astype(np.float32)
    return (lo[:, None] + (np.arange(ndepth, dtype=np.float32)[None] * step[:, None]).astype(np.float32)).astype(np.float32)


def remap_linear_const(img, x, y, border=-999.0):
    """cv2.remap(img, x, y, INTER_LINEAR, BORDER_CONSTANT, border) for float32 images, restated from OpenCV's published
    algorithm (opencv-python 4.5.5.62 is pinned by the reference's environment.yml:156 but absent here: parity unpinned for
    this step): coordinates go to fixed point with 5 fractional bits (cvRound = round half to even), the bilinear weights
    come from those fractions, taps outside the image take the border value."""
    img = _f32(img)
    H, W = img.shape
    sx = np.rint(np.asarray(x, np.float32) * np.float32(32)).astype(np.int64)
    sy = np.rint(np.asarray(y, np.float32) * np.float32(32)).astype(np.int64)
    ix, iy = sx >> 5, sy >> 5
    ax, ay = ((sx & 31).astype(np.float32) / np.float32(32)), ((sy & 31).astype(np.float32) / np.float32(32))

    def at(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        return np.where(ok, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(border)).astype(np.float32)
    one = np.float32(1)
    w00, w01, w10, w11 = (one - ax) * (one - ay), ax * (one - ay), (one - ax) * ay, ax * ay
    return (at(iy, ix) * w00 + at(iy, ix + 1) * w01 + at(iy + 1, ix) * w10 + at(iy + 1, ix + 1) * w11).astype(np.float32)


def reproject_with_depth(depth_ref, rpc_ref, depth_src, rpc_src):
    """tools/rpc_filter.py:11-48 of the reference: -> sampled source heights, reprojected (x, y), source (x, y)."""
    depth_ref = _f32(depth_ref)
    H, W = depth_ref.shape
    xr, yr = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    h = depth_ref.reshape(-1).astype(np.float64)
    lat, lon = rpc_project(rpc_ref, xr.reshape(-1), yr.reshape(-1), h, 0)
    xs, ys = rpc_project(rpc_src, lat, lon, h, 1)
    xs, ys = xs.reshape(H, W), ys.reshape(H, W)
    sampled = remap_linear_const(depth_src, xs.astype(np.float32), ys.astype(np.float32))
    sh = sampled.reshape(-1).astype(np.float64)
    lat, lon = rpc_project(rpc_src, xs.reshape(-1), ys.reshape(-1), sh, 0)
    xb, yb = rpc_project(rpc_ref, lat, lon, sh, 1)
    return sampled, xb.reshape(H, W), yb.reshape(H, W), xs, ys


def check_geometric_consistency(depth_ref, rpc_ref, depth_src, rpc_src, p_ratio, d_ratio):
    """tools/rpc_filter.py:51-70."""
    depth_ref = _f32(depth_ref)
    H, W = depth_ref.shape
    xr, yr = np.meshgrid(np.arange(W), np.arange(H))
    dep, xb, yb, xs, ys = reproject_with_depth(depth_ref, rpc_ref, depth_src, rpc_src)
    dist = np.sqrt((xb - xr) ** 2 + (yb - yr) ** 2)
    mask = np.logical_and(dist < p_ratio, np.abs(dep - depth_ref) < d_ratio)
    dep = dep.copy()
    dep[~mask] = 0
    return mask, dep, xs, ys


def filter_depth(depths, rpcs, p_ratio, d_ratio, geo_consist_num, prob=None, confidence_ratio=0.0):
    """tools/rpc_filter.py:73-112."""
    ref = _f32(depths[0])
    photo = (np.asarray(prob) > confidence_ratio) if prob is not None else np.ones(ref.shape, bool)
    geo_sum, ests = 0, []
    for v in range(1, len(depths)):
        m, dep, _, _ = check_geometric_consistency(ref, rpcs[0], depths[v], rpcs[v], p_ratio, d_ratio)
        geo_sum = geo_sum + m.astype(np.int32)
        ests.append(dep)
    averaged = (sum(ests) + ref) / (geo_sum + 1)
    return np.logical_and(photo, geo_sum >= geo_consist_num), averaged


def _pinhole_mats(K_ref, E_ref, K_src, E_src):
    """P = [K @ E[:3]; 0 0 0 1] and its inverse for both views (tools/pinhole_filter.py:17-24), float64."""
    bottom = np.array([[0.0, 0.0, 0.0, 1.0]])
    P_ref = np.concatenate((np.matmul(np.asarray(K_ref, np.float64), np.asarray(E_ref, np.float64)[:3]), bottom), axis=0)
    P_src = np.concatenate((np.matmul(np.asarray(K_src, np.float64), np.asarray(E_src, np.float64)[:3]), bottom), axis=0)
    return P_ref, np.linalg.inv(P_ref), P_src, np.linalg.inv(P_src)


def pinhole_reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src):
    """tools/pinhole_filter.py:7-46 of the reference: reference pixel * depth -> world -> source pixel (float64 matrix
    products, float32 coordinates into cv2.remap with its default border: constant 0), the sampled source depth back into
    the reference view.  -> sampled depth, reprojected (x, y) float32, source (x, y) float32."""
    depth_ref = _f32(depth_ref)
    H, W = depth_ref.shape
    row, col = np.meshgrid(range(H), range(W), indexing="ij")
    col, row = col.reshape(1, -1), row.reshape(1, -1)
    d = depth_ref.reshape(1, -1)
    P_ref, inv_ref, P_src, inv_src = _pinhole_mats(K_ref, E_ref, K_src, E_src)
    tmp = np.vstack((d * col, d * row, d, np.ones((1, W * H))))
    xy = np.matmul(P_src, np.matmul(inv_ref, tmp))
    xy = xy[:2] / xy[2]
    xs = xy[0].reshape(H, W).astype(np.float32)
    ys = xy[1].reshape(H, W).astype(np.float32)
    sampled = remap_linear_const(depth_src, xs, ys, border=0.0)
    sv = sampled.reshape(1, -1)
    tmp = np.vstack((sv * xy[0], sv * xy[1], sv, np.ones((1, W * H))))
    back = np.matmul(P_ref, np.matmul(inv_src, tmp))
    back = back[:2] / back[2]
    return sampled, back[0].reshape(H, W).astype(np.float32), back[1].reshape(H, W).astype(np.float32), xs, ys


def pinhole_check_geometric_consistency(depth_ref, K_ref, E_ref, depth_src, K_src, E_src, p_thre=1, relative_d_thre=0.01):
    """tools/pinhole_filter.py:49-67: mask = |reprojected - pixel| < p_thre and |sampled - depth| / depth < relative_d_thre."""
    depth_ref = _f32(depth_ref)
    H, W = depth_ref.shape
    xr, yr = np.meshgrid(np.arange(0, W), np.arange(0, H))
    dep, xb, yb, xs, ys = pinhole_reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src)
    dist = np.sqrt((xb - xr) ** 2 + (yb - yr) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(dep - depth_ref) / depth_ref
    mask = np.logical_and(dist < p_thre, rel < relative_d_thre)
    dep = dep.copy()
    dep[~mask] = 0
    return mask, dep, xs, ys


def window_regress(reg, depth, lamb=None):
    """casmvs / ucs regression: (depth, window-4 confidence[, lamb * std-dev]) -- orc_window_regress."""
    reg = _f32(reg)
    B, D, H, W = reg.shape
    depth, is4 = _depth_args(depth, B, D, H, W)
    od = np.empty((B, H, W), np.float32)
    oc = np.empty((B, H, W), np.float32)
    ov = np.empty((B, H, W), np.float32) if lamb is not None else None
    import ctypes
    f = lib().orc_window_regress
    f.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_float] + [ctypes.c_int] * 4
    f(_p(reg), _p(depth), is4, _p(od), _p(oc), _p(ov) if ov is not None else None, float(lamb or 0.0), B, D, H, W)
    return (od, oc, ov) if lamb is not None else (od, oc)


class StreamRegress:
    """The three float64 accumulators of networks/casred.py:182-184 and their updates."""

    def __init__(self, B, H, W):
        self.B, self.H, self.W = B, H, W
        self.exp_sum = np.zeros((B, 1, H, W), np.float64)
        self.depth_img = np.zeros((B, 1, H, W), np.float64)
        self.max_prob = np.zeros((B, 1, H, W), np.float64)

    def step(self, reg_plane, depth, d):
        reg_plane = _f32(reg_plane)
        depth = _f32(depth)
        D = depth.shape[1]
        is_plane = 1 if depth.ndim == 4 else 0
        lib().orc_stream_regress_step(_p(reg_plane), _p(depth), is_plane, _p(self.exp_sum), _p(self.depth_img),
                                      _p(self.max_prob), self.B, self.H, self.W, D, d)

    def final(self):
        n = self.B * self.H * self.W
        od = np.empty((self.B, self.H, self.W), np.float32)
        oc = np.empty((self.B, self.H, self.W), np.float32)
        lib().orc_stream_regress_final(_p(self.exp_sum), _p(self.depth_img), _p(self.max_prob), _p(od), _p(oc),
                                       C.c_size_t(n))
        return od, oc


# ---- RED regulariser (modules/module.py:6-58, :595-693) -----------------------------------------------------
def conv2d3x3(x, w, bias=None, stride=1):
    x, w = _f32(x), _f32(w)
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = np.empty((B, Cout, Ho, Wo), np.float32)
    bp = _p(_f32(bias)) if bias is not None else None
    lib().orc_conv2d3x3(_p(x), _p(w), bp, _p(out), B, Cin, Cout, H, W, stride)
    return out


def convT2d3x3(x, w, bias=None, stride=2, out_pad=1):
    x, w = _f32(x), _f32(w)
    B, Cin, H, W = x.shape
    Cout = w.shape[1]
    Ho, Wo = (H - 1) * stride + 1 + out_pad, (W - 1) * stride + 1 + out_pad
    out = np.empty((B, Cout, Ho, Wo), np.float32)
    bp = _p(_f32(bias)) if bias is not None else None
    lib().orc_convT2d3x3(_p(x), _p(w), bp, _p(out), B, Cin, Cout, H, W, stride, out_pad)
    return out


def groupnorm1(x, gamma, beta, eps=1e-5):
    x = np.array(x, dtype=np.float32, order="C", copy=True)
    B, Cc = x.shape[:2]
    lib().orc_groupnorm1(_p(x), _p(_f32(gamma)), _p(_f32(beta)), C.c_float(eps), B, Cc, int(np.prod(x.shape[2:])))
    return x


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)


def conv_gru(wt, prefix, x, h):
    """ConvGRUCell2.forward (module.py:49-58) with weights wt[prefix + ...]."""
    g = conv2d3x3(np.concatenate([x, h], 1), wt[prefix + "gate_conv.weight"], wt[prefix + "gate_conv.bias"])
    hc = h.shape[1]
    r = _sigmoid(groupnorm1(g[:, :hc], wt[prefix + "reset_gate_norm.weight"], wt[prefix + "reset_gate_norm.bias"]))
    u = _sigmoid(groupnorm1(g[:, hc:], wt[prefix + "update_gate_norm.weight"], wt[prefix + "update_gate_norm.bias"]))
    o = conv2d3x3(np.concatenate([x, r * h], 1), wt[prefix + "output_conv.weight"], wt[prefix + "output_conv.bias"])
    y = np.tanh(groupnorm1(o, wt[prefix + "output_norm.weight"], wt[prefix + "output_norm.bias"]).astype(np.float64)).astype(np.float32)
    return u * h + (1 - u) * y


def red_step(wt, cost, states):
    """slice_RED_Regularization.forward (module.py:672-693): one plane.  wt: name -> array (state_dict)."""
    relu = lambda a: np.maximum(a, 0)  # noqa: E731
    neg = -_f32(cost)
    s1, s2, s3, s4 = (_f32(s) for s in states)
    e1 = relu(conv2d3x3(neg, wt["conv1.conv.weight"], None, 2))
    e2 = relu(conv2d3x3(e1, wt["conv2.conv.weight"], None, 2))
    e3 = relu(conv2d3x3(e2, wt["conv3.conv.weight"], None, 2))
    s4 = conv_gru(wt, "conv_gru4.", e3, s4)
    u3 = relu(convT2d3x3(s4, wt["upconv3.conv.weight"], None, 2, 1))
    s3 = conv_gru(wt, "conv_gru3.", e2, s3)
    u2 = relu(convT2d3x3(u3 + s3, wt["upconv2.conv.weight"], None, 2, 1))
    s2 = conv_gru(wt, "conv_gru2.", e1, s2)
    u1 = relu(convT2d3x3(u2 + s2, wt["upconv1.conv.weight"], None, 2, 1))
    s1 = conv_gru(wt, "conv_gru1.", neg, s1)
    out = convT2d3x3(u1 + s1, wt["upconv2d.weight"], wt["upconv2d.bias"], 1, 0)
    return out, [s1, s2, s3, s4]


# ---- CostRegNet (modules/module.py:546-577), eval mode ---------------------------------------------------------
def conv3d3(x, w, stride=1):
    x, w = _f32(x), _f32(w)
    B, Cin, D, H, W = x.shape
    Cout = w.shape[0]
    out = np.empty((B, Cout, (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1), np.float32)
    lib().orc_conv3d3(_p(x), _p(w), _p(out), B, Cin, Cout, D, H, W, stride)
    return out


def convT3d3s2(x, w):
    x, w = _f32(x), _f32(w)
    B, Cin, D, H, W = x.shape
    Cout = w.shape[1]
    out = np.empty((B, Cout, 2 * D, 2 * H, 2 * W), np.float32)
    lib().orc_convT3d3s2(_p(x), _p(w), _p(out), B, Cin, Cout, D, H, W)
    return out


def _bn_eval(x, wt, prefix, eps=1e-5):
    shp = (1, -1, 1, 1, 1)
    g, b = wt[prefix + "weight"].astype(np.float64), wt[prefix + "bias"].astype(np.float64)
    m, v = wt[prefix + "running_mean"].astype(np.float64), wt[prefix + "running_var"].astype(np.float64)
    y = (x.astype(np.float64) - m.reshape(shp)) / np.sqrt(v.reshape(shp) + eps) * g.reshape(shp) + b.reshape(shp)
    return y.astype(np.float32)


def costregnet(wt, x):
    """CostRegNet.forward in eval mode.  wt: name -> array (state_dict incl. BN running stats)."""
    relu = lambda a: np.maximum(a, 0)  # noqa: E731

    def block(name, t, stride=1, transposed=False):
        y = convT3d3s2(t, wt[name + ".conv.weight"]) if transposed else conv3d3(t, wt[name + ".conv.weight"], stride)
        return relu(_bn_eval(y, wt, name + ".bn."))

    c0 = block("conv0", _f32(x))
    c2 = block("conv2", block("conv1", c0, 2))
    c4 = block("conv4", block("conv3", c2, 2))
    t = block("conv6", block("conv5", c4, 2))
    t = c4 + block("conv7", t, transposed=True)
    t = c2 + block("conv9", t, transposed=True)
    t = c0 + block("conv11", t, transposed=True)
    return conv3d3(t, wt["prob.weight"], 1)


# ---- FeatureNet (modules/module.py:442-543), eval mode, arch_mode="unet" ----------------------------------------
def conv2d_k(x, w, stride=1):
    x, w = _f32(x), _f32(w)
    B, Cin, H, W = x.shape
    Cout, K = w.shape[0], w.shape[2]
    pad = K // 2
    out = np.empty((B, Cout, (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1), np.float32)
    lib().orc_conv2d_k(_p(x), _p(w), None, _p(out), B, Cin, Cout, H, W, K, stride)
    return out


def _bn2d_eval(x, wt, prefix, eps=1e-5):
    shp = (1, -1, 1, 1)
    g, b = wt[prefix + "weight"].astype(np.float64), wt[prefix + "bias"].astype(np.float64)
    m, v = wt[prefix + "running_mean"].astype(np.float64), wt[prefix + "running_var"].astype(np.float64)
    y = (x.astype(np.float64) - m.reshape(shp)) / np.sqrt(v.reshape(shp) + eps) * g.reshape(shp) + b.reshape(shp)
    return y.astype(np.float32)


def featurenet(wt, img, arch_mode="unet"):
    """FeatureNet.forward (3 stages, arch_mode "unet" or "fpn") in eval mode.  wt: name -> array (state_dict incl.
    BN running stats).  Returns (stage1, stage2, stage3) = (B,4c,H/4,W/4), (B,2c,H/2,W/2), (B,c,H,W)."""
    relu = lambda a: np.maximum(a, 0)  # noqa: E731

    def cbr(name, t, stride=1):
        return relu(_bn2d_eval(conv2d_k(t, wt[name + ".conv.weight"], stride), wt, name + ".bn."))

    def fuse(name, pre, t):
        up = relu(_bn2d_eval(convT2d3x3(t, wt[name + ".deconv.conv.weight"], None, 2, 1), wt, name + ".deconv.bn."))
        return cbr(name + ".conv", np.concatenate([up, pre], 1))

    def lateral(name, pre, t):                             # F.interpolate(t, 2, "nearest") + innerK(pre), module.py:527-536
        up = np.repeat(np.repeat(t, 2, axis=2), 2, axis=3)
        return up + (conv2d_k(pre, wt[name + ".weight"]) + _f32(wt[name + ".bias"]).reshape(1, -1, 1, 1))

    c0 = cbr("conv0.1", cbr("conv0.0", _f32(img)))
    c1 = cbr("conv1.2", cbr("conv1.1", cbr("conv1.0", c0, 2)))
    c2 = cbr("conv2.2", cbr("conv2.1", cbr("conv2.0", c1, 2)))
    s1 = conv2d_k(c2, wt["out1.weight"])
    if arch_mode == "unet":
        f = fuse("deconv1", c1, c2)
        s2 = conv2d_k(f, wt["out2.weight"])
        f = fuse("deconv2", c0, f)
        s3 = conv2d_k(f, wt["out3.weight"])
    else:
        f = lateral("inner1", c1, c2)
        s2 = conv2d_k(f, wt["out2.weight"])
        f = lateral("inner2", c0, f)
        s3 = conv2d_k(f, wt["out3.weight"])
    return s1, s2, s3
