"""Height-hypothesis samplers, the step right before the hot path.

Mirror of /root/reference/modules/depth_range.py (get_cur_depth_range_samples :4,
get_depth_range_samples :23, uncertainty_aware_samples :45) as PyTorch ops (training, UCS-Net), plus the fused
form of SURVEY.md section 8(f) item 1: stage-1 hypotheses as exact (B,D) planes and `GeneratedHeights`, the
description of a later stage's hypotheses that the native kernels evaluate per pixel (no (B,D,H,W) tensor).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn.functional as F


def stage1_planes(depth_range, ndepth):
    """(B,D) hypothesis planes of the first stage (depth_range.py:26-33 of the reference).  The reference repeats
    them to (B,D,img_h,img_w) and resizes trilinearly to the stage grid; every pixel of a plane holds the same value
    and the resize weights (1/2 + 1/2, 1 + 0) reproduce it exactly, so the planes ARE its depth_values."""
    lo, hi = depth_range[:, 0], depth_range[:, -1]
    # a TENSOR divisor: with a python scalar torch's GPU kernels multiply by the rounded reciprocal, which is 1 ulp
    # off the true quotient the reference gets on the CPU (and the in-kernel generator computes)
    step = (hi - lo) / torch.full_like(lo, float(ndepth - 1))
    idx = torch.arange(0, ndepth, device=depth_range.device, dtype=depth_range.dtype).reshape(1, -1)
    return lo.unsqueeze(1) + idx * step.unsqueeze(1)


class _HeightGenStruct(ctypes.Structure):       # smvs_height_gen, include/satmvs.h
    _fields_ = [("prev_height", ctypes.c_void_p), ("prev_h", ctypes.c_int), ("prev_w", ctypes.c_int),
                ("img_h", ctypes.c_int), ("img_w", ctypes.c_int), ("ndepth", ctypes.c_int), ("interval", ctypes.c_double),
                ("prev_var", ctypes.c_void_p), ("range_min", ctypes.c_void_p), ("range_max", ctypes.c_void_p),
                ("arith", ctypes.c_int)]


class GeneratedHeights:
    """Hypotheses of cascade stage 2 / 3 described instead of materialised (networks/casred.py:134-145 +
    depth_range.py:4-20): previous height map (B,hp,wp), ndepth, interval, image size, this stage's size.

    Accepted wherever the native paths take `depth_values` (variance_cost_volume, the RED plane pipelines, the
    regressions); `.materialize()` gives the reference's (B,D,H,W) tensor (one kernel; torch composite on the CPU)."""

    def __init__(self, prev_height, ndepth, interval, img_hw, stage_hw, prev_var=None, range_min=None, range_max=None):
        self.prev = prev_height.detach().to(torch.float32).contiguous()
        self.ndepth, self.interval = int(ndepth), float(interval)
        self.img_h, self.img_w = int(img_hw[0]), int(img_hw[1])
        self.H, self.W = int(stage_hw[0]), int(stage_hw[1])
        # UCS-Net sampler (uncertainty_aware_samples behind networks/ucs.py:49-58): previous variance map + the height range
        self.var = None
        if prev_var is not None:
            self.var = prev_var.detach().to(torch.float32).contiguous()
            self.rmin = range_min.detach().to(torch.float32).contiguous()
            self.rmax = range_max.detach().to(torch.float32).contiguous()
            assert (self.img_h, self.img_w) == (self.H, self.W), "the UCS sampler resizes straight to the stage grid"

    @classmethod
    def ucs(cls, prev_depth, prev_var, range_min, range_max, ndepth, stage_hw):
        """Hypotheses of UCS-Net stage 2 / 3: prev_depth -+ prev_var resized to the stage grid, clamped to the range."""
        return cls(prev_depth, ndepth, 0.0, stage_hw, stage_hw, prev_var, range_min, range_max)

    @staticmethod
    def supported(img_hw, stage_hw):
        (ih, iw), (h, w) = img_hw, stage_hw
        return ih % h == 0 and iw % w == 0 and ih // h == iw // w and ih // h in (1, 2)

    @property
    def shape(self):
        return torch.Size((self.prev.shape[0], self.ndepth, self.H, self.W))

    @property
    def device(self):
        return self.prev.device

    def dim(self):
        return 4

    def c_struct(self):
        """ctypes smvs_height_gen; keep the returned object (and self) alive until the call has been enqueued.  Carries the
        arithmetic of the calling thread's arith_scope (satmvs_amd/_lib.py) for the cost-volume entry points."""
        from .. import _lib
        v = self.var is not None
        return _HeightGenStruct(self.prev.data_ptr(), self.prev.shape[1], self.prev.shape[2], self.img_h, self.img_w,
                                self.ndepth, self.interval, self.var.data_ptr() if v else None, self.rmin.data_ptr() if v else None,
                                self.rmax.data_ptr() if v else None, _lib.call_arith_bits())

    def materialize(self):
        if self.prev.is_cuda:
            from .. import _lib
            out = torch.empty(tuple(self.shape), dtype=torch.float32, device=self.prev.device)
            gs = self.c_struct()
            with torch.cuda.device(self.prev.device):
                _lib.call("smvs_height_hypotheses", ctypes.addressof(gs), _lib.ptr(out), out.shape[0], self.H, self.W,
                          _lib.current_stream(self.prev.device))
            return out
        if self.var is not None:
            cur = F.interpolate(self.prev.unsqueeze(1), [self.H, self.W], mode="bilinear", align_corners=False)
            ev = F.interpolate(self.var.unsqueeze(1), [self.H, self.W], mode="bilinear", align_corners=False)
            return uncertainty_aware_samples(cur, self.rmin, self.rmax, ev, self.ndepth, cur.device, cur.dtype,
                                             [cur.shape[0], self.H, self.W])
        cur = F.interpolate(self.prev.unsqueeze(1), [self.img_h, self.img_w], mode="bilinear", align_corners=False).squeeze(1)
        samples = get_cur_depth_range_samples(cur, self.ndepth, self.interval, list(cur.shape))
        return F.interpolate(samples.unsqueeze(1), [self.ndepth, self.H, self.W], mode="trilinear",
                             align_corners=False).squeeze(1)


def get_cur_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, shape):
    """cur_depth (B,H,W) -> (B,D,H,W): ndepth hypotheses centred on the previous stage's height."""
    assert cur_depth.shape == torch.Size(shape), "cur_depth:{}, input shape:{}".format(cur_depth.shape, shape)
    lo = cur_depth - ndepth / 2 * depth_inteval_pixel
    hi = cur_depth + ndepth / 2 * depth_inteval_pixel
    step = (hi - lo) / (ndepth - 1)
    idx = torch.arange(0, ndepth, device=cur_depth.device, dtype=cur_depth.dtype).reshape(1, -1, 1, 1)
    return lo.unsqueeze(1) + idx * step.unsqueeze(1)


def get_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, device, dtype, shape):
    """cur_depth (B,2+) [first stage: min..max] or (B,H,W) -> (B,D,H,W)."""
    if cur_depth.dim() == 2:
        lo, hi = cur_depth[:, 0], cur_depth[:, -1]
        step = (hi - lo) / (ndepth - 1)
        planes = lo.unsqueeze(1) + torch.arange(0, ndepth, device=device, dtype=dtype).reshape(1, -1) * step.unsqueeze(1)
        return planes.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, shape[1], shape[2])
    return get_cur_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, shape)


def uncertainty_aware_samples(cur_depth, depth_min, depth_max, exp_var, ndepth, device, dtype, shape):
    """UCS-Net sampler: first stage as above; later stages span cur_depth +- exp_var clipped to the range."""
    eps = 1e-12
    if cur_depth.dim() == 2:
        lo, hi = cur_depth[:, 0], cur_depth[:, -1]
        step = (hi - lo) / (ndepth - 1)
        planes = lo.unsqueeze(1) + torch.arange(0, ndepth, device=device, dtype=dtype).reshape(1, -1) * step.unsqueeze(1)
        return planes.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, shape[1], shape[2])
    assert ndepth > 1
    b = cur_depth.shape[0]
    low = torch.maximum(cur_depth - exp_var, depth_min.view(b, 1, 1, 1).to(cur_depth.dtype))
    high = torch.minimum(cur_depth + exp_var, depth_max.view(b, 1, 1, 1).to(cur_depth.dtype))
    step = (high - low) / (float(ndepth) - 1)
    return torch.cat([low + step * i + eps for i in range(int(ndepth))], 1)


def stage_hypotheses(prev_depth, depth_range, ndepth, interval, img_hw, stage_hw, dtype, device, batch):
    """`depth_values` of one cascade stage, in the cheapest exact form (networks/casred.py:134-145):
      * first stage (prev_depth None): the (B,D) planes;
      * later stages without autograd on a GPU: a GeneratedHeights description (evaluated inside the kernels);
      * otherwise the reference's composite -- bilinear resize, samples, trilinear resize -- as a (B,D,H,W) tensor."""
    if prev_depth is None:
        return stage1_planes(depth_range.to(dtype), ndepth)
    if (not torch.is_grad_enabled() and prev_depth.is_cuda and prev_depth.dtype == torch.float32
            and GeneratedHeights.supported(img_hw, stage_hw)):
        return GeneratedHeights(prev_depth, ndepth, interval, img_hw, stage_hw)
    cur = F.interpolate(prev_depth.unsqueeze(1), list(img_hw), mode="bilinear", align_corners=False).squeeze(1)
    samples = get_depth_range_samples(cur_depth=cur, ndepth=ndepth, depth_inteval_pixel=interval, dtype=dtype,
                                      device=device, shape=[batch, img_hw[0], img_hw[1]])
    return F.interpolate(samples.unsqueeze(1), [ndepth, stage_hw[0], stage_hw[1]], mode="trilinear",
                         align_corners=False).squeeze(1)
