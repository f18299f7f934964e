"""UCS-Net (uncertainty-aware cascade) on the native cost-volume engine.

Same public surface as /root/reference/networks/ucs.py: compute_depth (:9), UCSNet (:79).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..modules.depth_range import GeneratedHeights, uncertainty_aware_samples
from ..modules.module import CostRegNet, FeatureNet, window_depth_regression
from .. import _lib
from ..modules.warping import variance_cost_volume


def compute_depth(feats, proj_mats, depth_samps, cost_reg, lamb, geo_model, is_training=False, use_qc=False):
    num_depth = depth_samps.shape[1]
    n_proj = len(proj_mats) if use_qc else proj_mats.shape[1]
    assert n_proj == len(feats), "Different number of images and projection matrices"
    with _lib.pipeline_arith_scope():
        volume_variance = variance_cost_volume(feats, proj_mats, depth_samps, geo_model, use_qc)
    reg = cost_reg(volume_variance).squeeze(1)
    depth, prob_conf, exp_variance = window_depth_regression(reg, depth_samps, lamb=lamb)      # ucs.py:60-74
    return {"depth": depth, "photometric_confidence": prob_conf, "variance": exp_variance}


class UCSNet(nn.Module):
    def __init__(self, geo_model, lamb=1.5, stage_configs=[64, 32, 8], grad_method="detach", base_chs=[8, 8, 8],
                 feat_ext_ch=8, use_qc=False, arith=None):
        super().__init__()
        self.arith = arith                      # this model's arithmetic of the variance build: "exact" / "fused"; None = the enclosing
        # _lib.arith_scope if there is one, else "exact" (_lib.pipeline_arith_scope)
        assert geo_model in ["rpc", "pinhole"]
        self.geo_model, self.stage_configs, self.grad_method = geo_model, stage_configs, grad_method
        self.base_chs, self.lamb, self.num_stage, self.use_qc = base_chs, lamb, len(stage_configs), use_qc
        self.ds_ratio = {"stage1": 4.0, "stage2": 2.0, "stage3": 1.0}
        self.feature_extraction = FeatureNet(base_channels=feat_ext_ch, num_stage=self.num_stage)
        self.cost_regularization = nn.ModuleList([
            CostRegNet(in_channels=self.feature_extraction.out_channels[i], base_channels=self.base_chs[i])
            for i in range(self.num_stage)])

    def forward(self, imgs, proj_matrices, depth_values):
        from .. import _lib
        with _lib.pipeline_arith_scope(getattr(self, "arith", None)):
            return self._forward(imgs, proj_matrices, depth_values)

    def _forward(self, imgs, proj_matrices, depth_values):
        features = self.feature_extraction.forward_views(imgs)
        outputs = {}
        depth, exp_var = None, None
        depth_min, depth_max = depth_values[:, 0], depth_values[:, -1]
        for stage_idx in range(self.num_stage):
            key = "stage{}".format(stage_idx + 1)
            feats = [f[key] for f in features]
            scale = int(self.ds_ratio[key])
            cur_h, cur_w = imgs.shape[3] // scale, imgs.shape[4] // scale
            nd = self.stage_configs[stage_idx]
            if (depth is not None and not torch.is_grad_enabled() and depth.is_cuda and depth.dtype == torch.float32):
                # inference on the GPU: the two bilinear resizes and the sampler are evaluated inside the kernels
                # (no (B,D,H,W) hypothesis tensor, no D small launches of the concatenating sampler)
                samples = GeneratedHeights.ucs(depth, exp_var, depth_min, depth_max, nd, (cur_h, cur_w))
            else:
                if depth is not None:
                    if self.grad_method == "detach":
                        cur_depth, exp_var = depth.detach(), exp_var.detach()
                    else:
                        cur_depth = depth
                    cur_depth = F.interpolate(cur_depth.unsqueeze(1), [cur_h, cur_w], mode="bilinear", align_corners=False)
                    exp_var = F.interpolate(exp_var.unsqueeze(1), [cur_h, cur_w], mode="bilinear", align_corners=False)
                else:
                    cur_depth = depth_values
                samples = uncertainty_aware_samples(cur_depth=cur_depth, depth_min=depth_min, depth_max=depth_max,
                                                    exp_var=exp_var, ndepth=nd,
                                                    dtype=imgs.dtype, device=imgs.device, shape=[imgs.shape[0], cur_h, cur_w])
            out = compute_depth(feats, proj_matrices[key], depth_samps=samples,
                                cost_reg=self.cost_regularization[stage_idx], lamb=self.lamb, geo_model=self.geo_model,
                                is_training=self.training, use_qc=self.use_qc)
            depth, exp_var = out["depth"], out["variance"]
            outputs[key] = out
        return outputs
