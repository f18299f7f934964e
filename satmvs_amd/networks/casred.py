"""Cascade RED-Net on the native cost-volume engine.

Same public surface as /root/reference/networks/casred.py:
    compute_depth_when_train(features, proj_matrices, depth_values, num_depth, cost_regularization,
                             geo_model, use_qc)                                              (:10)
    compute_depth_when_pred(...same...)                                                     (:161)
    CascadeREDNet(geo_model, min_interval, ndepths, depth_interals_ratio, cr_base_chs, use_qc) (:68)
    Infer_CascadeREDNet(...)                                                                (:242)
so train.py / predict.py style drivers run unchanged.  What differs is inside:
  * the per-source rpc_warping loop + volume_sum/volume_sq_sum passes are ONE fused HIP launch
    (variance_cost_volume); no warped volume and no (B,N,20) float64 `coef` tensor exist;
  * the pred loop's float64 accumulators are updated by smvs_stream_regress_step;
  * with world_size > 1 (one process per GPU) the pred loop can shard the height planes
    (satmvs_amd/shard.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..modules.depth_range import GeneratedHeights, stage_hypotheses
from ..modules.module import (FeatureNet, RED_Regularization, StreamingRegression, guard_miopen_find, restore_miopen_find, slice_RED_Regularization,
                              softmax_depth_regression)
from .. import _lib
from ..modules.warping import variance_cost_volume

_STAGE_SCALES = {3: {"stage1": 4.0, "stage2": 2.0, "stage3": 1.0}, 2: {"stage1": 4.0, "stage2": 1.0}}


def _check(features, proj_matrices, depth_values, num_depth, use_qc):
    n_proj = len(proj_matrices) if use_qc else proj_matrices.shape[1]
    assert len(features) == n_proj, "Different number of images and projection matrices"
    assert depth_values.shape[1] == num_depth, "depth_values.shape[1]:{}  num_depth:{}".format(
        depth_values.shape[1], num_depth)


def compute_depth_when_train(features, proj_matrices, depth_values, num_depth, cost_regularization, geo_model,
                             use_qc):
    """Whole-volume path: variance volume -> regulariser -> softmax -> expected height."""
    _check(features, proj_matrices, depth_values, num_depth, use_qc)
    if (not torch.is_grad_enabled() and hasattr(cost_regularization, "native_volume")
            and cost_regularization._use_native(features[0])):
        # inference: plane pipeline (variance plane -> RED step) straight into the (B,D,H,W) regularised cost;
        # the (B,C,D,H,W) variance volume is never materialised
        dv = depth_values if isinstance(depth_values, GeneratedHeights) else depth_values.detach().to(torch.float32).contiguous()
        reg = cost_regularization.native_volume(features, proj_matrices, dv, geo_model, use_qc)
    else:
        with _lib.pipeline_arith_scope():                          # (the native pipeline defaults the same way, include/satmvs.h)
            volume_variance = variance_cost_volume(features, proj_matrices, depth_values, geo_model, use_qc)
        reg = cost_regularization(volume_variance)                 # (B,D,H,W)
    depth, confidence = softmax_depth_regression(reg, depth_values)
    return {"depth": depth, "photometric_confidence": confidence}


def compute_depth_when_pred(features, proj_matrices, depth_values, num_depth, cost_regularization, geo_model,
                            use_qc):
    """Plane-at-a-time path with recurrent state and streaming regression (memory-lean inference)."""
    _check(features, proj_matrices, depth_values, num_depth, use_qc)
    ref = features[0]
    b, _, h, w = ref.shape
    states = cost_regularization.initial_states(b, h, w, ref.device)
    acc = StreamingRegression(b, h, w, ref.device)
    gen = isinstance(depth_values, GeneratedHeights)
    dv = depth_values if gen else depth_values.detach().to(torch.float32).contiguous()
    if hasattr(cost_regularization, "native_pred_planes") and cost_regularization._use_native(ref):
        # the whole plane loop in one native call: variance plane -> RED step -> regression update
        cost_regularization.native_pred_planes(features, proj_matrices, dv, geo_model, use_qc, states, acc.state,
                                               0, num_depth)
    else:
        if gen:
            dv = dv.materialize()
        for d in range(num_depth):
            with _lib.pipeline_arith_scope():
                plane = variance_cost_volume(features, proj_matrices, dv, geo_model, use_qc, d_begin=d, d_end=d + 1)
            reg, *states = cost_regularization(plane.squeeze(2), *states)
            acc.step(reg, dv, d)
    depth, confidence = acc.result()
    return {"depth": depth, "photometric_confidence": confidence}


class _CascadeRED(nn.Module):
    regulariser_cls = None
    compute = None

    def __init__(self, geo_model, min_interval=2.5, ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1],
                 cr_base_chs=[8, 8, 8], use_qc=False, arith=None):
        super().__init__()
        self.arith = arith                      # this model's arithmetic of the variance build: "exact" / "fused"; None = the enclosing
        # _lib.arith_scope if there is one, else "exact" (_lib.pipeline_arith_scope)
        assert geo_model in ["rpc", "pinhole"]
        assert len(ndepths) == len(depth_interals_ratio)
        self.geo_model = geo_model
        self.ndepths = ndepths
        self.depth_interals_ratio = depth_interals_ratio
        self.cr_base_chs = cr_base_chs
        self.num_stage = len(ndepths)
        self.min_interval = min_interval
        self.use_qc = use_qc
        self.stage_infos = {k: {"scale": v} for k, v in _STAGE_SCALES[self.num_stage].items()}
        self.feature = FeatureNet(base_channels=8, stride=4, num_stage=self.num_stage, arch_mode="unet")
        self.cost_regularization = nn.ModuleList([
            type(self).regulariser_cls(in_channels=self.feature.out_channels[i], base_channels=self.cr_base_chs[i])
            for i in range(self.num_stage)])

    def forward(self, imgs, proj_matrices, depth_values):
        from .. import _lib
        with _lib.pipeline_arith_scope(getattr(self, "arith", None)):
            return self._forward(imgs, proj_matrices, depth_values)

    def _forward(self, imgs, proj_matrices, depth_values):
        """imgs (B,V,3,H,W); proj_matrices {"stageK": (B,V,170)|(B,V,4,4)|QC dicts}; depth_values (B,2)."""
        if imgs.is_cuda:
            guard_miopen_find() if self.training else restore_miopen_find()
        features = self.feature.forward_views(imgs)
        img_h, img_w = int(imgs.shape[3]), int(imgs.shape[4])
        outputs = {}
        depth = None
        for stage_idx in range(self.num_stage):
            key = "stage{}".format(stage_idx + 1)
            feats = [f[key] for f in features]
            scale = int(self.stage_infos[key]["scale"])
            # hypotheses of this stage (casred.py:134-145): (B,D) planes for stage 1, a per-pixel generator evaluated
            # inside the kernels for the later stages in inference, the reference's (B,D,H,W) tensor under autograd
            dv = stage_hypotheses(depth, depth_values, self.ndepths[stage_idx],
                                  self.depth_interals_ratio[stage_idx] * self.min_interval, (img_h, img_w),
                                  (img_h // scale, img_w // scale), imgs.dtype, imgs.device, imgs.shape[0])
            out = type(self).compute(feats, proj_matrices[key], depth_values=dv, num_depth=self.ndepths[stage_idx],
                                     cost_regularization=self.cost_regularization[stage_idx],
                                     geo_model=self.geo_model, use_qc=self.use_qc)
            depth = out["depth"]
            outputs[key] = out
            outputs.update(out)
        return outputs


class CascadeREDNet(_CascadeRED):
    """train + test network (whole volume per stage).  reference: casred.py:68-156."""
    regulariser_cls = RED_Regularization
    compute = staticmethod(compute_depth_when_train)


class Infer_CascadeREDNet(_CascadeRED):
    """predict network (plane-at-a-time).  reference: casred.py:242-333."""
    regulariser_cls = slice_RED_Regularization
    compute = staticmethod(compute_depth_when_pred)
