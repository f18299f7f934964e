"""Height-hypothesis samplers, the step right before the hot path.

Mirror of /root/reference/modules/depth_range.py (get_cur_depth_range_samples :4,
get_depth_range_samples :23, uncertainty_aware_samples :45).  Stock PyTorch elementwise ops on the
device; fusing them into the kernel prologue is SURVEY.md section 8(f) item 1.
"""
from __future__ import annotations

import torch


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
