"""Drop-in operator surface of the reference's modules/warping.py, backed by the HIP library.

Same names, argument order and tensor layouts as /root/reference/modules/warping.py:
    homo_warping(src_fea, src_proj, ref_proj, depth_values)                    (:6)
    rpc_warping(src_fea, src_rpc, ref_rpc, depth_values, coef)                 (:310)
    rpc_warping_enisum(src_fea, src_rpc, ref_rpc, depth_values)                (:139)
    RPC_Photo2Obj(insamp, inline, inhei, rpc, coef)                            (:255)
    RPC_Obj2Photo(inlat, inlon, inhei, rpc, coef)                              (:218)
    RPC_Photo2Obj_enisum / RPC_Obj2Photo_enisum                                (:96, :60)
plus the fused operator the networks use instead of the per-source loop:
    variance_cost_volume(features, proj_matrices, depth_values, geo_model, use_qc)
        == the body of networks/casred.py:22-53 up to `volume_variance`.

Everything runs on the MI355X through include/satmvs.h; CPU tensors raise (no fallback).
`coef` (the reference's (B, N, 20) float64 scratch) is accepted and ignored.
Gradients flow to the feature maps only -- the sampling grid is built under no_grad in the
reference too (warping.py:322).
"""
from __future__ import annotations

import torch

from .. import _lib
from ..rpc_synth import qc_tensor_to_coeffs  # noqa: F401  (host-side layout helper)

__all__ = ["homo_warping", "rpc_warping", "rpc_warping_enisum", "RPC_Photo2Obj", "RPC_Obj2Photo",
           "RPC_Photo2Obj_enisum", "RPC_Obj2Photo_enisum", "variance_cost_volume", "qc_dict_to_rpc",
           "prepare_geometry", "plane_coefficients"]


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def _f64c(t):
    return t.detach().to(torch.float64).contiguous()


def _depth_arg(depth_values, B, H, W):
    """(B,D) or (B,D,H,W) float32 contiguous + the depth_is_4d flag (warping.py:329-332)."""
    d = _f32c(depth_values)
    if d.dim() == 2:
        if d.shape[0] != B:
            raise ValueError("depth_values batch %d != %d" % (d.shape[0], B))
        return d, 0, d.shape[1]
    if d.dim() == 4:
        if d.shape[0] != B or d.shape[2] != H or d.shape[3] != W:
            raise ValueError("depth_values %s does not match features (B=%d,H=%d,W=%d)" % (tuple(d.shape), B, H, W))
        return d, 1, d.shape[1]
    raise ValueError("depth_values must be (B,D) or (B,D,H,W), got %s" % (tuple(depth_values.shape),))


# ---- QC dict <-> 170-vector -----------------------------------------------------------------------
_QC_SCALARS = ["line_off", "samp_off", "lat_off", "lon_off", "height_off",
               "line_scale", "samp_scale", "lat_scale", "lon_scale", "height_scale"]
_QC_TENSORS = ["line_num", "line_den", "samp_num", "samp_den", "lat_num", "lat_den", "lon_num", "lon_den"]
# (i,j,k) of the 20 monomials in the symmetric (4,4,4) tensor and their multiplicities
_QC_IDX = [(0, 0, 0), (0, 0, 1), (0, 0, 2), (0, 0, 3), (0, 1, 2), (0, 1, 3), (0, 2, 3), (0, 1, 1), (0, 2, 2),
           (0, 3, 3), (1, 2, 3), (1, 1, 1), (1, 2, 2), (1, 3, 3), (1, 1, 2), (2, 2, 2), (2, 3, 3), (1, 1, 3),
           (2, 2, 3), (3, 3, 3)]
_QC_MULT = [1, 3, 3, 3, 6, 6, 6, 3, 3, 3, 6, 1, 3, 3, 3, 1, 3, 3, 3, 1]


def qc_dict_to_rpc(rpc):
    """The `use_qc` dict of dataset/data_io.py:123-150 -> (B,170) float64 on the same device.

    T[i,j,k] holds coefficient/multiplicity (data_io.py:95-120), so c_m = T[idx_m] * mult_m.
    """
    cols = [rpc[k].reshape(-1).to(torch.float64) for k in _QC_SCALARS]
    B = cols[0].shape[0]
    out = torch.empty((B, 170), dtype=torch.float64, device=cols[0].device)
    for i, c in enumerate(cols):
        out[:, i] = c
    mult = torch.tensor(_QC_MULT, dtype=torch.float64, device=out.device)
    ii = torch.tensor([t[0] for t in _QC_IDX], device=out.device)
    jj = torch.tensor([t[1] for t in _QC_IDX], device=out.device)
    kk = torch.tensor([t[2] for t in _QC_IDX], device=out.device)
    for n, name in enumerate(_QC_TENSORS):
        T = rpc[name + "_tensor"].to(torch.float64).reshape(B, 4, 4, 4)
        out[:, 10 + 20 * n: 30 + 20 * n] = T[:, ii, jj, kk] * mult
    return out


# ---- single-source warps ---------------------------------------------------------------------------
class _WarpFn(torch.autograd.Function):
    """geo 0: rpc (src_geo, ref_geo = (B,170));  geo 1: homography (src_geo = composed (B,4,4))."""

    @staticmethod
    def forward(ctx, src_fea, src_geo, ref_geo, depth, is4d, geo):
        dev = _lib.require_device(src_fea, src_geo, depth)
        fea = _f32c(src_fea)
        B, C, H, W = fea.shape
        D = depth.shape[1]
        out = torch.empty((B, C, D, H, W), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            if geo == 0:
                _lib.call("smvs_rpc_warp_fwd", _lib.ptr(fea), _lib.ptr(src_geo), _lib.ptr(ref_geo), _lib.ptr(depth),
                          is4d, _lib.ptr(out), B, C, D, H, W, st)
            else:
                _lib.call("smvs_homo_warp_fwd", _lib.ptr(fea), _lib.ptr(src_geo), _lib.ptr(depth), is4d,
                          _lib.ptr(out), B, C, D, H, W, st)
        ctx.save_for_backward(src_geo, ref_geo if ref_geo is not None else src_geo, depth)
        ctx.meta = (is4d, geo, (B, C, D, H, W))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        src_geo, ref_geo, depth = ctx.saved_tensors
        is4d, geo, (B, C, D, H, W) = ctx.meta
        g = _f32c(grad_out)
        dev = g.device
        grad_src = torch.zeros((B, C, H, W), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            if geo == 0:
                _lib.call("smvs_rpc_warp_bwd", _lib.ptr(g), _lib.ptr(src_geo), _lib.ptr(ref_geo), _lib.ptr(depth),
                          is4d, _lib.ptr(grad_src), B, C, D, H, W, st)
            else:
                _lib.call("smvs_homo_warp_bwd", _lib.ptr(g), _lib.ptr(src_geo), _lib.ptr(depth), is4d,
                          _lib.ptr(grad_src), B, C, D, H, W, st)
        return grad_src, None, None, None, None, None


def _compose_homography(src_proj, ref_proj):
    """src_proj @ inverse(ref_proj) on the device (warping.py:19), any leading shape (...,4,4)."""
    dev = _lib.require_device(src_proj, ref_proj)
    s, r = _f64c(src_proj), _f64c(ref_proj)
    n = s.numel() // 16
    out = torch.empty_like(s)
    with torch.cuda.device(dev):
        _lib.call("smvs_homo_compose", _lib.ptr(s), _lib.ptr(r), _lib.ptr(out), n, _lib.current_stream(dev))
    return out


def rpc_warping(src_fea, src_rpc, ref_rpc, depth_values, coef=None):
    """(B,C,H,W), (B,170), (B,170), (B,D)|(B,D,H,W) -> (B,C,D,H,W).  reference: warping.py:310-365."""
    B, _, H, W = src_fea.shape
    depth, is4d, _ = _depth_arg(depth_values, B, H, W)
    return _WarpFn.apply(src_fea, _f64c(src_rpc), _f64c(ref_rpc), depth, is4d, 0)


def rpc_warping_enisum(src_fea, src_rpc, ref_rpc, depth_values):
    """QC-tensor variant (warping.py:139-178): same kernel after T -> 20 coefficients."""
    return rpc_warping(src_fea, qc_dict_to_rpc(src_rpc), qc_dict_to_rpc(ref_rpc), depth_values, None)


def homo_warping(src_fea, src_proj, ref_proj, depth_values):
    """(B,C,H,W), (B,4,4), (B,4,4), (B,D)|(B,D,H,W) -> (B,C,D,H,W).  reference: warping.py:6-44."""
    B, _, H, W = src_fea.shape
    depth, is4d, _ = _depth_arg(depth_values, B, H, W)
    proj = _compose_homography(src_proj, ref_proj)
    return _WarpFn.apply(src_fea, proj, None, depth, is4d, 1)


# ---- flat projectors ----------------------------------------------------------------------------------
def _project(a, b, h, rpc, direction):
    dev = _lib.require_device(a, b, h, rpc)
    shape = a.shape
    rpc = _f64c(rpc).reshape(-1, 170)
    B = rpc.shape[0]
    a2, b2, h2 = (_f64c(t).reshape(B, -1) for t in (a, b, h))
    o0, o1 = torch.empty_like(a2), torch.empty_like(a2)
    n = a2.shape[1]
    with torch.cuda.device(dev):
        st = _lib.current_stream(dev)
        for i in range(B):
            _lib.call("smvs_rpc_project", _lib.ptr(rpc[i]), _lib.ptr(a2[i]), _lib.ptr(b2[i]), _lib.ptr(h2[i]),
                      _lib.ptr(o0[i]), _lib.ptr(o1[i]), n, direction, st)
    return o0.reshape(shape), o1.reshape(shape)


def RPC_Photo2Obj(insamp, inline, inhei, rpc, coef=None):
    """(B,N) samp, line, height float64 + rpc (B,170) -> lat, lon (B,N).  reference: warping.py:255-307."""
    return _project(insamp, inline, inhei, rpc, 0)


def RPC_Obj2Photo(inlat, inlon, inhei, rpc, coef=None):
    """(B,N) lat, lon, height float64 + rpc (B,170) -> samp, line (B,N).  reference: warping.py:218-252."""
    return _project(inlat, inlon, inhei, rpc, 1)


def RPC_Photo2Obj_enisum(insamp, inline, inhei, rpc):
    return _project(insamp, inline, inhei, qc_dict_to_rpc(rpc), 0)


def RPC_Obj2Photo_enisum(inlat, inlon, inhei, rpc):
    return _project(inlat, inlon, inhei, qc_dict_to_rpc(rpc), 1)


# ---- fused cost volume ---------------------------------------------------------------------------------
def plane_coefficients(geo, depth, is4d, n_src, H, W, d_begin=0, d_end=None):
    """smvs_rpc_plane_coef: the source views' cubics folded at the height of every plane in [d_begin, d_end) --
    the workspace smvs_rpc_costvol_fwd_pc takes.  geo (B,V,170) f64, depth (B,D) or (B,D,H,W) f32, both on the GPU."""
    dev = _lib.require_device(geo, depth)
    B, D = depth.shape[0], depth.shape[1]
    d_end = D if d_end is None else d_end
    nbytes = _lib.load().smvs_rpc_plane_coef_bytes(B, n_src, D)
    pc = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.call("smvs_rpc_plane_coef", _lib.ptr(geo), _lib.ptr(depth), is4d, _lib.ptr(pc), B, n_src, D, H, W,
                  d_begin, d_end, _lib.current_stream(dev))
    return pc


# below this many voxels the fold's own launch (~2-5 us on the stream) costs more than the ~4 % of the build it saves
# (bench extra stage3_rpc_3view_768x384x8_c8: 2.4 M voxels, 41.9 us build, 46.9 us with the fold)
_FOLD_MIN_VOXELS = 8 << 20


class _CostVolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geo_kind, geo, depth, is4d, d_begin, d_end, plane_constant, ref_fea, *src_feas):
        dev = _lib.require_device(ref_fea, geo, depth, *src_feas)
        ref = _f32c(ref_fea)
        srcs = [_f32c(s) for s in src_feas]
        B, C, H, W = ref.shape
        D = depth.shape[1]
        for s in srcs:
            if s.shape != ref.shape:
                raise ValueError("source feature %s != reference feature %s" % (tuple(s.shape), tuple(ref.shape)))
        nd = d_end - d_begin
        out = torch.empty((B, C, nd, H, W), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            if geo_kind == 0 and plane_constant and B * nd * H * W >= _FOLD_MIN_VOXELS:
                # plane-constant heights (stage 1 of every cascade) collapse the source cubics to bivariate ones: fold them for
                # the planes of this launch.  The kernel runs its geometry from the folded records and checks its own heights
                # against them afterwards, wave by wave -- a wave whose heights differ redoes it with the trivariate chain,
                # which is why heights that vary inside a plane (stages 2 / 3) are not sent this way
                pc = plane_coefficients(geo, depth, is4d, len(srcs), H, W, d_begin, d_end)
                _lib.call("smvs_rpc_costvol_fwd_pc", _lib.ptr(ref), _lib.ptr_array(srcs), len(srcs), _lib.ptr(geo), _lib.ptr(depth),
                          is4d | _lib.call_arith_bits(), _lib.ptr(pc), _lib.ptr(out), B, C, D, H, W, d_begin, d_end, nd, 0, st)
            else:
                name = "smvs_rpc_costvol_fwd" if geo_kind == 0 else "smvs_homo_costvol_fwd"
                _lib.call(name, _lib.ptr(ref), _lib.ptr_array(srcs), len(srcs), _lib.ptr(geo), _lib.ptr(depth), is4d | _lib.call_arith_bits(),
                          _lib.ptr(out), B, C, D, H, W, d_begin, d_end, nd, 0, st)
        ctx.save_for_backward(geo, depth, ref, *srcs)
        ctx.meta = (geo_kind, is4d, d_begin, d_end, (B, C, D, H, W))
        return out

    @staticmethod
    def backward(ctx, grad_var):
        geo, depth, ref, *srcs = ctx.saved_tensors
        geo_kind, is4d, d_begin, d_end, (B, C, D, H, W) = ctx.meta
        if d_begin != 0 or d_end != D:
            raise _lib.SatMVSNativeError("backward of a depth-sharded cost volume is not supported")
        g = _f32c(grad_var)
        dev = g.device
        g_ref = torch.zeros_like(ref)
        g_srcs = [torch.zeros_like(s) for s in srcs]
        with torch.cuda.device(dev):
            _lib.call("smvs_costvol_bwd", geo_kind, _lib.ptr(g), _lib.ptr(ref), _lib.ptr_array(srcs), len(srcs),
                      _lib.ptr(geo), _lib.ptr(depth), is4d, _lib.ptr(g_ref), _lib.ptr_array(g_srcs),
                      B, C, D, H, W, _lib.current_stream(dev))
        return (None, None, None, None, None, None, None, g_ref, *g_srcs)


def prepare_geometry(features, proj_matrices, geo_model="rpc", use_qc=False):
    """(geo_kind, geo tensor) in the layout the C ABI takes: rpc (B,V,170) f64, or composed homographies
    (B,V-1,4,4) f64 -- from exactly what the reference networks receive as `proj_matrices`."""
    B = features[0].shape[0]
    V = len(features)
    if geo_model == "rpc":
        geo = torch.stack([qc_dict_to_rpc(p) for p in proj_matrices], dim=1) if use_qc else proj_matrices
        geo = _f64c(geo)
        if tuple(geo.shape) != (B, V, 170):
            raise ValueError("rpc proj_matrices must be (B,V,170) = %s, got %s" % ((B, V, 170), tuple(geo.shape)))
        return 0, geo
    if geo_model == "pinhole":
        P = _f64c(proj_matrices)
        if tuple(P.shape) != (B, V, 4, 4):
            raise ValueError("pinhole proj_matrices must be (B,V,4,4), got %s" % (tuple(P.shape),))
        return 1, _compose_homography(P[:, 1:].contiguous(), P[:, :1].expand(B, V - 1, 4, 4).contiguous())
    raise ValueError("geo_model must be 'rpc' or 'pinhole', got %r" % (geo_model,))


def variance_cost_volume(features, proj_matrices, depth_values, geo_model="rpc", use_qc=False,
                         d_begin=0, d_end=None, plane_constant=None):
    """Fused per-channel variance cost volume: (B,C,d_end-d_begin,H,W) float32.

    features: list of V tensors (B,C,H,W), view 0 = reference (networks/casred.py:22).
    proj_matrices: rpc -> (B,V,170) float64 (or, with use_qc, the list of V QC dicts);
                   pinhole -> (B,V,4,4) float64 -- exactly what the reference networks receive.
    depth_values: (B,D) or (B,D,H,W) float32.  [d_begin,d_end) selects the planes to build
    (the pred loop passes d,d+1; a depth shard passes its own range).
    plane_constant: every pixel of a plane holds the same height (stage 1), so the rpc build may use the folded
    bivariate source cubics (smvs_rpc_plane_coef + smvs_rpc_costvol_fwd_pc).  None = decide from the argument: (B,D)
    heights and (B,D,H,W) views broadcast over H and W are; a materialised (B,D,H,W) tensor is taken as per-pixel.
    True on heights that are not plane-constant is still correct (the kernel checks), only slower.  Builds of fewer than
    8 Mi voxels never fold: the extra launch costs more than it saves.
    """
    ref_fea = features[0]
    B, _, H, W = ref_fea.shape
    from .depth_range import GeneratedHeights
    if isinstance(depth_values, GeneratedHeights):
        gen = depth_values
        if tuple(gen.shape) != (B, gen.ndepth, H, W):
            raise ValueError("generated heights %s do not match features (B=%d,H=%d,W=%d)" % (tuple(gen.shape), B, H, W))
        if torch.is_grad_enabled() and any(f.requires_grad for f in features):
            depth_values = gen.materialize()             # autograd path: the backward kernel takes the tensor
        else:
            return _costvol_generated(features, proj_matrices, gen, geo_model, use_qc, d_begin, d_end)
    if plane_constant is None:
        plane_constant = depth_values.dim() == 2 or (depth_values.dim() == 4 and (H == 1 or depth_values.stride(2) == 0)
                                                     and (W == 1 or depth_values.stride(3) == 0))
    depth, is4d, D = _depth_arg(depth_values, B, H, W)
    d_end = D if d_end is None else d_end
    if not (0 <= d_begin <= d_end <= D):
        raise ValueError("bad plane range [%d,%d) of %d" % (d_begin, d_end, D))
    kind, geo = prepare_geometry(features, proj_matrices, geo_model, use_qc)
    return _CostVolFn.apply(kind, geo, depth, is4d, d_begin, d_end, bool(plane_constant), ref_fea, *features[1:])


def _costvol_generated(features, proj_matrices, gen, geo_model, use_qc, d_begin, d_end):
    """variance_cost_volume with the hypotheses evaluated inside the kernel (smvs_*_costvol_fwd_gen); no autograd."""
    import ctypes
    ref = _f32c(features[0])
    srcs = [_f32c(s) for s in features[1:]]
    B, C, H, W = ref.shape
    D = gen.ndepth
    d_end = D if d_end is None else d_end
    if not (0 <= d_begin <= d_end <= D):
        raise ValueError("bad plane range [%d,%d) of %d" % (d_begin, d_end, D))
    kind, geo = prepare_geometry(features, proj_matrices, geo_model, use_qc)
    dev = _lib.require_device(ref, geo, gen.prev, *srcs)
    nd = d_end - d_begin
    out = torch.empty((B, C, nd, H, W), dtype=torch.float32, device=dev)
    gs = gen.c_struct()
    with torch.cuda.device(dev):
        _lib.call("smvs_rpc_costvol_fwd_gen" if kind == 0 else "smvs_homo_costvol_fwd_gen", _lib.ptr(ref),
                  _lib.ptr_array(srcs), len(srcs), _lib.ptr(geo), ctypes.addressof(gs), _lib.ptr(out),
                  B, C, D, H, W, d_begin, d_end, nd, 0, _lib.current_stream(dev))
    return out
