"""Geometric-consistency check between pinhole depth maps, the reference's post-processing step for geo_model="pinhole"
(/root/reference/tools/pinhole_filter.py:7-67), on the native remap kernel (csrc/filter.hip).

Same names, arguments and return values as the reference (numpy in, numpy out; GPU tensors are accepted too).  The 4 x 4
projection matrices are formed and inverted on the host in numpy, as the reference does; one launch per (reference,
source) pair does its four matrix products over the pixels, the cv2.remap and the mask.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .rpc_filter import _dev, _f32


def _mats(intrinsics_ref, extrinsics_ref, intrinsics_src, extrinsics_src):
    """P_ref, inverse(P_ref), P_src, inverse(P_src) with P = [K @ E[:3]; 0 0 0 1] (pinhole_filter.py:17-24), (4,4,4) float64."""
    def to_np(a):
        return a.detach().cpu().numpy().astype(np.float64) if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    bottom = np.array([[0.0, 0.0, 0.0, 1.0]])
    P_ref = np.concatenate((np.matmul(to_np(intrinsics_ref), to_np(extrinsics_ref)[:3]), bottom), axis=0)
    P_src = np.concatenate((np.matmul(to_np(intrinsics_src), to_np(extrinsics_src)[:3]), bottom), axis=0)
    return np.ascontiguousarray(np.stack([P_ref, np.linalg.inv(P_ref), P_src, np.linalg.inv(P_src)]))


def _pair(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src, p_thre, relative_d_thre, want_back):
    dev = _dev()
    dr, ds = _f32(depth_ref, dev), _f32(depth_src, dev)
    mats = torch.from_numpy(_mats(intrinsics_ref, extrinsics_ref, intrinsics_src, extrinsics_src)).to(dev)
    H, W = dr.shape
    Hs, Ws = ds.shape
    mask = torch.empty((H, W), dtype=torch.uint8, device=dev)
    dep, xs, ys = (torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(3))
    xb = torch.empty_like(xs) if want_back else None
    yb = torch.empty_like(xs) if want_back else None
    with torch.cuda.device(dev):
        _lib.call("smvs_pinhole_geo_consistency", _lib.ptr(dr), _lib.ptr(ds), _lib.ptr(mats), H, W, Hs, Ws, float(p_thre), float(relative_d_thre),
                  _lib.ptr(mask), _lib.ptr(dep), _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(xb) if want_back else None,
                  _lib.ptr(yb) if want_back else None, _lib.current_stream(dev))
    return mask, dep, xs, ys, xb, yb


def reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    """pinhole_filter.py:7-46 -> (depth_reprojected, x_reprojected, y_reprojected, x_src, y_src), float32 (H,W) each."""
    _, dep, xs, ys, xb, yb = _pair(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src, np.inf, np.inf, True)
    return dep.cpu().numpy(), xb.cpu().numpy(), yb.cpu().numpy(), xs.cpu().numpy(), ys.cpu().numpy()


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src,
                                p_thre=1, relative_d_thre=0.01):
    """pinhole_filter.py:49-67 -> (mask bool, depth_reprojected (0 outside the mask), x2d_src, y2d_src)."""
    mask, dep, xs, ys, _, _ = _pair(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src,
                                    p_thre, relative_d_thre, False)
    return mask.bool().cpu().numpy(), dep.cpu().numpy(), xs.cpu().numpy(), ys.cpu().numpy()
