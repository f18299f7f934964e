"""Geometric-consistency filtering of the per-view height maps, the reference's post-processing step
(/root/reference/tools/rpc_filter.py:11-112), on the native projector + remap kernel.

Same names, arguments and return values as the reference (numpy in, numpy out); torch tensors on the GPU are accepted
too and stay there until the final conversion.  One launch per (reference, source) pair does what the reference does
with two cupy projector calls, a cv2.remap and two more projector calls.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def _dev():
    if not torch.cuda.is_available():
        raise _lib.SatMVSNativeError("rpc_filter needs an MI355X (no CPU fallback for the product path)")
    return torch.device("cuda", torch.cuda.current_device())


def _f32(a, dev):
    t = torch.as_tensor(a)
    return t.to(device=dev, dtype=torch.float32).contiguous()


def _f64(a, dev):
    t = torch.as_tensor(a)
    return t.to(device=dev, dtype=torch.float64).contiguous()


def _pair(depth_ref, rpc_ref, depth_src, rpc_src, p_ratio, d_ratio, want_back):
    dev = _dev()
    dr, ds = _f32(depth_ref, dev), _f32(depth_src, dev)
    rr, rs = _f64(rpc_ref, dev).reshape(-1), _f64(rpc_src, dev).reshape(-1)
    if rr.numel() != 170 or rs.numel() != 170:
        raise ValueError("rpc vectors must hold 170 values")
    H, W = dr.shape
    Hs, Ws = ds.shape
    mask = torch.empty((H, W), dtype=torch.uint8, device=dev)
    dep = torch.empty((H, W), dtype=torch.float32, device=dev)
    xs = torch.empty((H, W), dtype=torch.float64, device=dev)
    ys = torch.empty_like(xs)
    xb = torch.empty_like(xs) if want_back else None
    yb = torch.empty_like(xs) if want_back else None
    with torch.cuda.device(dev):
        _lib.call("smvs_rpc_geo_consistency", _lib.ptr(dr), _lib.ptr(rr), _lib.ptr(ds), _lib.ptr(rs), H, W, Hs, Ws,
                  float(p_ratio), float(d_ratio), _lib.ptr(mask), _lib.ptr(dep), _lib.ptr(xs), _lib.ptr(ys),
                  _lib.ptr(xb) if want_back else None, _lib.ptr(yb) if want_back else None, _lib.current_stream(dev))
    return mask, dep, xs, ys, xb, yb


def reproject_with_depth(depth_ref, rpc_ref, depth_src, rpc_src):
    """rpc_filter.py:11-48 -> (sampled_depth_src, x_reprojected, y_reprojected, x_src, y_src)."""
    # asking for the back-projection makes the kernel return the raw remap value everywhere (no masking, NaN included)
    _, dep, xs, ys, xb, yb = _pair(depth_ref, rpc_ref, depth_src, rpc_src, np.inf, np.inf, True)
    return dep.cpu().numpy(), xb.cpu().numpy(), yb.cpu().numpy(), xs.cpu().numpy(), ys.cpu().numpy()


def check_geometric_consistency(depth_ref, rpc_ref, depth_src, rpc_src, p_ratio, d_ratio):
    """rpc_filter.py:51-70 -> (mask bool, depth_reprojected (0 outside the mask), x2d_src, y2d_src)."""
    mask, dep, xs, ys, _, _ = _pair(depth_ref, rpc_ref, depth_src, rpc_src, p_ratio, d_ratio, False)
    return mask.bool().cpu().numpy(), dep.cpu().numpy(), xs.cpu().numpy(), ys.cpu().numpy()


def filter_depth(depths, rpcs, p_ratio, d_ratio, geo_consist_num, prob=None, confidence_ratio=0.0):
    """rpc_filter.py:73-112: view 0 is the reference; -> (final_mask bool, depth_est_averaged)."""
    dev = _dev()
    ref_depth = _f32(depths[0], dev)
    photo_mask = (_f32(prob, dev) > confidence_ratio) if prob is not None else torch.ones_like(ref_depth, dtype=torch.bool)
    geo_sum = torch.zeros(ref_depth.shape, dtype=torch.int32, device=dev)
    acc = torch.zeros_like(ref_depth)
    for v in range(1, len(depths)):
        mask, dep, _, _, _, _ = _pair(ref_depth, rpcs[0], depths[v], rpcs[v], p_ratio, d_ratio, False)
        geo_sum += mask.to(torch.int32)
        acc = acc + dep
    averaged = (acc + ref_depth).double() / (geo_sum + 1).double()          # numpy: float32 / int32 -> float64
    final = photo_mask & (geo_sum >= geo_consist_num)
    return final.cpu().numpy(), averaged.cpu().numpy()
