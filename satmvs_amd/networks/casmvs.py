"""Cascade MVSNet (3-D conv regulariser) on the native cost-volume engine.

Same public surface as /root/reference/networks/casmvs.py: DepthNet (:11), CascadeMVSNet (:79).
The per-source warp + variance accumulation is the fused HIP launch; in inference CostRegNet runs on its native
kernels (smvs_costreg_fwd) and softmax + expected height + window-4 confidence are one kernel
(smvs_window_regress_fwd); training keeps the differentiable torch composites on the same parameters.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..modules.depth_range import stage_hypotheses
from ..modules.module import CostRegNet, FeatureNet, depth_regression, window_depth_regression
from .. import _lib
from ..modules.warping import variance_cost_volume

Align_Corners_Range = False


class DepthNet(nn.Module):
    def forward(self, features, proj_matrices, depth_values, num_depth, cost_regularization, geo_model, use_qc=False):
        n_proj = len(proj_matrices) if use_qc else proj_matrices.shape[1]
        assert len(features) == n_proj, "Different number of images and projection matrices"
        assert depth_values.shape[1] == num_depth, "depth_values.shape[1]:{}  num_depth:{}".format(
            depth_values.shape[1], num_depth)
        with _lib.pipeline_arith_scope():
            volume_variance = variance_cost_volume(features, proj_matrices, depth_values, geo_model, use_qc)
        reg = cost_regularization(volume_variance).squeeze(1)
        depth, conf = window_depth_regression(reg, depth_values)      # casmvs.py:66-74; native when no gradient is wanted
        return {"depth": depth, "photometric_confidence": conf}


class CascadeMVSNet(nn.Module):
    def __init__(self, geo_model, refine=False, min_interval=2.5, ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1],
                 share_cr=False, grad_method="detach", arch_mode="fpn", cr_base_chs=[8, 8, 8], use_qc=False, arith=None):
        super().__init__()
        self.arith = arith                      # this model's arithmetic of the variance build: "exact" / "fused"; None = the enclosing
        # _lib.arith_scope if there is one, else "exact" (_lib.pipeline_arith_scope)
        assert geo_model in ["rpc", "pinhole"]
        assert len(ndepths) == len(depth_interals_ratio)
        if refine:
            raise NotImplementedError("RefineNet is dead code in the reference (module.py:580-592, F.cat does not exist)")
        self.geo_model, self.refine, self.share_cr = geo_model, refine, share_cr
        self.ndepths, self.depth_interals_ratio = ndepths, depth_interals_ratio
        self.grad_method, self.arch_mode, self.cr_base_chs = grad_method, arch_mode, cr_base_chs
        self.num_stage, self.min_interval, self.use_qc = len(ndepths), min_interval, use_qc
        self.stage_infos = {"stage1": {"scale": 4.0}, "stage2": {"scale": 2.0}, "stage3": {"scale": 1.0}}
        self.feature = FeatureNet(base_channels=8, stride=4, num_stage=self.num_stage, arch_mode=self.arch_mode)
        if self.share_cr:
            self.cost_regularization = CostRegNet(in_channels=self.feature.out_channels, base_channels=8)
        else:
            self.cost_regularization = nn.ModuleList([
                CostRegNet(in_channels=self.feature.out_channels[i], base_channels=self.cr_base_chs[i])
                for i in range(self.num_stage)])
        self.DepthNet = DepthNet()

    def forward(self, imgs, proj_matrices, depth_values):
        from .. import _lib
        with _lib.pipeline_arith_scope(getattr(self, "arith", None)):
            return self._forward(imgs, proj_matrices, depth_values)

    def _forward(self, imgs, proj_matrices, depth_values):
        features = self.feature.forward_views(imgs)
        h, w = int(imgs.shape[3]), int(imgs.shape[4])
        outputs = {}
        depth = None
        for stage_idx in range(self.num_stage):
            key = "stage{}".format(stage_idx + 1)
            feats = [f[key] for f in features]
            scale = int(self.stage_infos[key]["scale"])
            cur = None if depth is None else (depth.detach() if self.grad_method == "detach" else depth)
            dv = stage_hypotheses(cur, depth_values, self.ndepths[stage_idx],
                                  self.depth_interals_ratio[stage_idx] * self.min_interval, (h, w),
                                  (h // scale, w // scale), imgs.dtype, imgs.device, imgs.shape[0])
            reg = self.cost_regularization if self.share_cr else self.cost_regularization[stage_idx]
            out = self.DepthNet(feats, proj_matrices[key], depth_values=dv, num_depth=self.ndepths[stage_idx],
                                cost_regularization=reg, geo_model=self.geo_model, use_qc=self.use_qc)
            depth = out["depth"]
            outputs[key] = out
            outputs.update(out)
        return outputs
