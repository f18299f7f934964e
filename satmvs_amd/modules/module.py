"""Regularisers, regression and feature extractor with the reference's names and parameter trees.

Mirror of the parts of /root/reference/modules/module.py that the cascade networks use, so that
`state_dict()` keys and shapes are identical and reference checkpoints (train.py:215-220) load:

    ConvGRUCell2 (:6)            RED_Regularization (:595)     slice_RED_Regularization (:653)
    ConvReLU (:178) ConvTransReLU (:208)                       CostRegNet (:546)
    Conv2d (:78) Deconv2d (:120) DeConv2dFuse (:303) Conv3d (:324) Deconv3d (:369)
    FeatureNet (:442)            depth_regression (:433)

This file holds the reference-named nn.Module classes and the INFERENCE paths; the training autograd functions live in
train_fns.py, the A/B switches in switches.py.  What is native here (inference): the RED plane
step and pred loop (smvs_red_step_fwd / smvs_red_pred_planes / smvs_red_volume_planes), CostRegNet
(smvs_costreg_fwd), FeatureNet (smvs_featnet_fwd) and the height regressions (smvs_softmax_regress_fwd,
smvs_window_regress_fwd, smvs_stream_regress_*).  The nn.Module trees only hold the parameters and the
differentiable fallback; sub-modules are created in the same order as the reference so that a given
torch.manual_seed yields the same initial parameters.
"""
from __future__ import annotations

import os
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .switches import SW, guard_miopen_find, restore_miopen_find          # noqa: F401  (re-exported: the networks import them from here)
from .train_fns import (_reparametrize, _NATIVE_KINDS, GroupNorm1, _ConvGRUCellFn, _GroupNormPairFn, _GruBlendFn, _GruMulCatFn, _PlaneViewsFn, _WgradArena, _WgradSink, _conv3x3_cat, _conv3x3, _native_kind, _side_streams, _PARAM_EPOCH, bump_param_epoch, _GroupNorm1Fn, _Conv3x3NativeFn, _Conv3x3WgradFn, _conv_packed, _gn_scratch, _f32c_fast, _TLS, _placeholder, _is_placeholder, _wgrad_now_or_later, _conv3d, _Conv3dWgradFn, _Conv3dNativeFn, _bn3d_relu, _BatchNormReluFn)          # noqa: F401


# ---- native-path plumbing shared by the three modules with HIP kernels ---------------------------------
def _tensors_by_path(module, names):
    """Parameters / buffers by dotted attribute path.  Unlike named_parameters() this also works on the
    replicas nn.DataParallel makes for every forward (train.py:129, predict.py:85): torch.nn.parallel.replicate
    empties `_parameters` and sets the broadcast copies as plain tensor attributes."""
    out = []
    for n in names:
        obj = module
        for part in n.split("."):
            obj = getattr(obj, part)
        out.append(obj)
    return out


_PACK_CACHE = {}        # key -> (packed tensor, weak references to the source storages); module-level so that it survives
_PACK_CACHE_PER_DEVICE = 16   # DataParallel's per-forward replicas (the device-0 replica shares the parent's storages);
                              # entries are counted per device, so 8 replicas do not evict each other every forward
_PACK_CACHE_LOCK = threading.Lock()   # nn.DataParallel runs its replicas on one Python thread per device


def _packed(kind, device, tensors, build):
    """Kernel-layout copy of `tensors`, rebuilt only when a parameter's storage or version changed.

    The key is (address, version) of every source tensor; an entry is only trusted while all the storages it was
    packed from are still alive (weak references), because the caching allocator hands a freed parameter's address
    to the next model's parameters."""
    from torch.multiprocessing.reductions import StorageWeakRef
    dkey = str(device)
    key = (kind, dkey, _PARAM_EPOCH[0]) + tuple((t.data_ptr(), t._version) for t in tensors)
    with _PACK_CACHE_LOCK:
        hit = _PACK_CACHE.get(key)
        if hit is not None and any(r.expired() for r in hit[1]):
            hit = None
    if hit is None:
        packed = build()                                   # device work outside the lock: replicas pack concurrently
        refs = [StorageWeakRef(t.untyped_storage()) for t in tensors]
        with _PACK_CACHE_LOCK:
            for k in [k for k, v in _PACK_CACHE.items() if any(r.expired() for r in v[1])]:
                del _PACK_CACHE[k]
            mine = [k for k in _PACK_CACHE if k[1] == dkey]
            if len(mine) >= _PACK_CACHE_PER_DEVICE:
                del _PACK_CACHE[mine[0]]                   # oldest entry of this device (dicts keep insertion order)
            hit = _PACK_CACHE.setdefault(key, (packed, refs))
    return hit[0]


def _workspace(module, attr, nbytes, dev):
    """Scratch buffer kept on the module; a DataParallel replica (re-created per forward) gets a fresh one
    from torch's caching allocator, which costs no device allocation after the first forward."""
    ws = module.__dict__.get(attr)
    if ws is None or ws.numel() < nbytes or ws.device != dev:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        module.__dict__[attr] = ws
    return ws


def _autograd_needed(x, tensors):
    """True when a gradient could be asked of this call: the native kernels are forward-only."""
    return torch.is_grad_enabled() and (x.requires_grad or any(t.requires_grad for t in tensors))


# ---- small conv wrappers (parameter names: .conv / .bn) ----------------------------------------------
class Conv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.kernel_size, self.stride = kernel_size, stride
        self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x):
        # 3x3 / pad 1 layers: native under autograd where a gradient is wanted on the GPU (forward, input and weight gradient on the RED
        # regulariser's layer kernels, train_fns._conv3x3_cat); anything else (FeatureNet's 5x5 stride-2 layers, inference) is torch's
        x = _conv3x3_cat(self.conv, x) if SW.train_featnet_native and self.conv.bias is None else self.conv(x)
        if self.bn is not None:
            y = _bn3d_relu(self.bn, x, self.relu)     # training form + ReLU as one native operator (train_fns._bn3d_relu), or None
            if y is not None:
                return y
            x = self.bn(x)
        return F.relu(x, inplace=True) if self.relu else x


class Deconv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        super().__init__()
        assert stride in (1, 2)
        self.out_channels, self.stride = out_channels, stride
        self.conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x):
        y = _conv3x3_cat(self.conv, x) if SW.train_featnet_native and self.conv.bias is None else self.conv(x)
        if self.stride == 2:
            h, w = x.shape[2], x.shape[3]
            y = y[:, :, :2 * h, :2 * w].contiguous()
        if self.bn is not None:
            z = _bn3d_relu(self.bn, y, self.relu)
            if z is not None:
                return z
            y = self.bn(y)
        return F.relu(y, inplace=True) if self.relu else y


class DeConv2dFuse(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, relu=True, bn=True, bn_momentum=0.1):
        super().__init__()
        self.deconv = Deconv2d(in_channels, out_channels, kernel_size, stride=2, padding=1, output_padding=1,
                               bn=True, relu=relu, bn_momentum=bn_momentum)
        self.conv = Conv2d(2 * out_channels, out_channels, kernel_size, stride=1, padding=1, bn=bn, relu=relu,
                           bn_momentum=bn_momentum)

    def forward(self, x_pre, x):
        return self.conv(torch.cat((self.deconv(x), x_pre), dim=1))


class Conv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        super().__init__()
        assert stride in (1, 2)
        self.out_channels, self.kernel_size, self.stride = out_channels, kernel_size, stride
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm3d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x):
        x = _conv3d(self.conv, x)             # native under autograd where a gradient is wanted on the GPU (train_fns._conv3d)
        if self.bn is not None:
            y = _bn3d_relu(self.bn, x, self.relu)     # training form + ReLU as one native operator (train_fns._bn3d_relu), or None
            if y is not None:
                return y
            x = self.bn(x)
        return F.relu(x, inplace=True) if self.relu else x


class Deconv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        super().__init__()
        assert stride in (1, 2)
        self.out_channels, self.stride = out_channels, stride
        self.conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm3d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x):
        x = _conv3d(self.conv, x)
        if self.bn is not None:
            y = _bn3d_relu(self.bn, x, self.relu)
            if y is not None:
                return y
            x = self.bn(x)
        return F.relu(x, inplace=True) if self.relu else x


class ConvReLU(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)

    def forward(self, x):
        return _conv3x3_cat(self.conv, x, None, True)


class ConvTransReLU(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1, output_pad=1):
        super().__init__()
        self.conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=pad,
                                       output_padding=output_pad, bias=False)

    def forward(self, x):
        return _conv3x3_cat(self.conv, x, None, True)


# ---- recurrent regulariser (RED) ------------------------------------------------------------------------
class ConvGRUCell2(nn.Module):
    """3x3 convolutional GRU with GroupNorm(1, C) on every gate.  reference: module.py:6-58."""

    def __init__(self, input_channel, output_channel, kernel_size):
        super().__init__()
        cat_ch = input_channel + output_channel
        self.output_channel = output_channel
        self.gate_conv = nn.Conv2d(cat_ch, output_channel * 2, kernel_size, padding=1)
        self.reset_gate_norm = GroupNorm1(1, output_channel, 1e-5, True)
        self.update_gate_norm = GroupNorm1(1, output_channel, 1e-5, True)
        self.output_conv = nn.Conv2d(cat_ch, output_channel, kernel_size, padding=1)
        self.output_norm = GroupNorm1(1, output_channel, 1e-5, True)
        self.activation = nn.Tanh()

    def forward(self, x, h=None):
        if h is None:
            h = torch.zeros((x.shape[0], self.output_channel, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device)
        gc, oc, rn, un, on = self.gate_conv, self.output_conv, self.reset_gate_norm, self.update_gate_norm, self.output_norm
        C = self.output_channel
        if (x.is_cuda and x.shape[0] == 1 and x.dtype is torch.float32 and h.dtype is torch.float32 and torch.is_grad_enabled() and not SW.train_composite_mask
                and _native_kind(gc) == "c1" and _native_kind(oc) == "c1" and gc.bias is not None and oc.bias is not None
                and all(p.dtype is torch.float32 and p.is_contiguous() for p in (gc.weight, oc.weight))
                and x.shape[1] % 2 == 0 and (C * x.shape[2] * x.shape[3]) % 4 == 0 and gc.bias.data_ptr() % 16 == 0 and oc.bias.data_ptr() % 16 == 0
                and (x.shape[1] + 2 * C) * x.shape[2] * x.shape[3] * 4 < 2 ** 31 and rn.affine and un.affine and on.affine and rn.eps == un.eps == on.eps):
            new_h = _ConvGRUCellFn.apply(x, h, gc.weight, gc.bias, rn.weight, rn.bias, un.weight, un.bias, oc.weight, oc.bias, on.weight, on.bias, rn.eps)
            return new_h, new_h
        gates = _conv3x3_cat(self.gate_conv, x, h)
        # the native element-wise paths are float32 kernels with 16-byte vector accesses and 16-bit grid limits: anything else
        # (a .double() model, exotic channel counts that leave a gate half unaligned, B*C > 65535) takes torch's operators
        native = (x.is_cuda and x.dtype is torch.float32 and h.dtype is torch.float32 and gates.dtype is torch.float32
                  and x.shape[0] * gates.shape[1] <= 65535)
        if native and not (SW.train_composite_mask & 1):              # both gate norms + sigmoids: one native call each way
            rn, un = self.reset_gate_norm, self.update_gate_norm
            r, u = torch.split(_GroupNormPairFn.apply(gates, rn.weight, rn.bias, un.weight, un.bias, rn.eps, 1), gates.shape[1] // 2, 1)
        else:
            r, u = torch.split(gates, gates.shape[1] // 2, 1)
            r = self.reset_gate_norm(r, "sigmoid")
            u = self.update_gate_norm(u, "sigmoid")
        # the cell's element-wise steps: one native launch each way
        xc = _GruMulCatFn.apply(x, r, h) if native and not (SW.train_composite_mask & 2) else torch.cat((x, r * h), dim=1)
        cand = self.output_norm(_conv3x3_cat(self.output_conv, xc), "tanh")
        blend_native = native and not (SW.train_composite_mask & 4) and all(
            t.data_ptr() % 16 == 0 or not t.is_contiguous() for t in (u, h, cand))       # non-contiguous operands are copied (aligned) first
        new_h = _GruBlendFn.apply(u, h, cand) if blend_native else u * h + (1 - u) * cand
        return new_h, new_h


class _REDCore(nn.Module):
    """Layers shared by the train and pred variants (same attribute names, same creation order)."""

    def __init__(self, in_channels, base_channels=8):
        super().__init__()
        self.base_channels = base_channels
        self.conv_gru1 = ConvGRUCell2(in_channels, base_channels, 3)
        self.conv_gru2 = ConvGRUCell2(base_channels * 2, base_channels * 2, 3)
        self.conv_gru3 = ConvGRUCell2(base_channels * 4, base_channels * 4, 3)
        self.conv_gru4 = ConvGRUCell2(base_channels * 8, base_channels * 8, 3)
        self.conv1 = ConvReLU(in_channels, base_channels * 2, 3, 2, 1)
        self.conv2 = ConvReLU(base_channels * 2, base_channels * 4, 3, 2, 1)
        self.conv3 = ConvReLU(base_channels * 4, base_channels * 8, 3, 2, 1)
        self.upconv3 = ConvTransReLU(base_channels * 8, base_channels * 4, 3, 2, 1, 1)
        self.upconv2 = ConvTransReLU(base_channels * 4, base_channels * 2, 3, 2, 1, 1)
        self.upconv1 = ConvTransReLU(base_channels * 2, base_channels, 3, 2, 1, 1)
        self.upconv2d = nn.ConvTranspose2d(base_channels, 1, kernel_size=3, stride=1, padding=1, output_padding=0)

    @staticmethod
    def initial_states(b, h, w, device, dtype=torch.float32):
        # hidden sizes are hard-coded 8/16/32/64 in the reference (module.py:617-620)
        return [torch.zeros((b, 8, h, w), device=device, dtype=dtype),
                torch.zeros((b, 16, h // 2, w // 2), device=device, dtype=dtype),
                torch.zeros((b, 32, h // 4, w // 4), device=device, dtype=dtype),
                torch.zeros((b, 64, h // 8, w // 8), device=device, dtype=dtype)]

    # -- native plane step ---------------------------------------------------------------------------------
    _PARAM_ORDER = [g + n for g in ("conv_gru1.", "conv_gru2.", "conv_gru3.", "conv_gru4.")
                    for n in ("gate_conv.weight", "gate_conv.bias", "reset_gate_norm.weight", "reset_gate_norm.bias",
                              "update_gate_norm.weight", "update_gate_norm.bias", "output_conv.weight",
                              "output_conv.bias", "output_norm.weight", "output_norm.bias")] + [
        "conv1.conv.weight", "conv2.conv.weight", "conv3.conv.weight", "upconv1.conv.weight", "upconv2.conv.weight",
        "upconv3.conv.weight", "upconv2d.weight", "upconv2d.bias"]

    def _packed_weights(self, device):
        """Device buffer in the kernel's layout; rebuilt when any parameter changed (optimizer step, load)."""
        tensors = _tensors_by_path(self, self._PARAM_ORDER)
        in_ch = tensors[self._PARAM_ORDER.index("conv1.conv.weight")].shape[1]

        def build():
            lib = _lib.load()
            packed = torch.empty(lib.smvs_red_packed_floats(in_ch), dtype=torch.float32, device=device)
            src = [t.detach().to(device=device, dtype=torch.float32).contiguous() for t in tensors]
            with torch.cuda.device(device):
                _lib.call("smvs_red_pack_weights", _lib.ptr_array(src), in_ch, _lib.ptr(packed), _lib.current_stream(device))
            return packed
        return _packed("red", device, tensors, build), in_ch

    def native_step(self, cost, s1, s2, s3, s4):
        """One plane through smvs_red_step_fwd (HIP).  States are updated in place and returned."""
        dev = _lib.require_device(cost, s1, s2, s3, s4)
        packed, in_ch = self._packed_weights(dev)
        cost = cost.detach().to(torch.float32).contiguous()
        b, c, h, w = cost.shape
        if c != in_ch:
            raise ValueError("cost has %d channels, regulariser expects %d" % (c, in_ch))
        states = [s.detach().to(torch.float32).contiguous() for s in (s1, s2, s3, s4)]
        lib = _lib.load()
        nbytes = lib.smvs_red_workspace_bytes(b, c, h, w)
        if nbytes == 0:
            raise ValueError("plane %dx%d is not a positive multiple of 8" % (h, w))
        ws = _workspace(self, "_ws_step", nbytes, dev)
        out = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("smvs_red_step_fwd", _lib.ptr(packed), _lib.ptr(cost), *[_lib.ptr(s) for s in states],
                      _lib.ptr(out), _lib.ptr(ws), nbytes, b, c, h, w, _lib.current_stream(dev))
        return (out, *states)

    def native_pred_planes(self, features, proj_matrices, depth_values, geo_model, use_qc, states, acc_state,
                           d_begin, d_end, reg_volume=None):
        """Planes [d_begin,d_end) of the pred loop in one native call (smvs_red_pred_planes): per plane
        variance build -> RED step -> float64 regression update.  `states` (list of 4) and `acc_state`
        ((3,B,H,W) float64) are updated in place."""
        from .warping import _depth_arg, prepare_geometry
        ref = features[0]
        dev = _lib.require_device(ref, acc_state, reg_volume, *states)
        packed, in_ch = self._packed_weights(dev)
        feats = [f.detach().to(torch.float32).contiguous() for f in features]
        b, c, h, w = feats[0].shape
        if c != in_ch:
            raise ValueError("features have %d channels, regulariser expects %d" % (c, in_ch))
        kind, geo = prepare_geometry(features, proj_matrices, geo_model, use_qc)
        from .depth_range import GeneratedHeights
        gen = depth_values if isinstance(depth_values, GeneratedHeights) else None
        if gen is not None:
            D = gen.ndepth
            if tuple(gen.shape) != (b, D, h, w):
                raise ValueError("generated heights %s do not match features" % (tuple(gen.shape),))
        else:
            depth, is4d, D = _depth_arg(depth_values, b, h, w)
        lib = _lib.load()
        nbytes = lib.smvs_red_pred_workspace_bytes(b, c, h, w)
        if nbytes == 0:
            raise ValueError("plane %dx%d is not a positive multiple of 8" % (h, w))
        ws = _workspace(self, "_ws_pred", nbytes, dev)
        for s in states:
            if s.dtype != torch.float32 or not s.is_contiguous():
                raise ValueError("states must be contiguous float32")
        target = acc_state if reg_volume is None else reg_volume
        name = "smvs_red_pred_planes" if reg_volume is None else "smvs_red_volume_planes"
        with torch.cuda.device(dev):
            if gen is not None:
                import ctypes
                gs = gen.c_struct()
                _lib.call(name + "_gen", kind, _lib.ptr(feats[0]), _lib.ptr_array(feats[1:]), len(feats) - 1,
                          _lib.ptr(geo), ctypes.addressof(gs), _lib.ptr(packed), *[_lib.ptr(s) for s in states],
                          _lib.ptr(target), _lib.ptr(ws), nbytes, b, c, D, h, w, d_begin, d_end,
                          _lib.current_stream(dev))
            else:
                _lib.call(name, kind, _lib.ptr(feats[0]), _lib.ptr_array(feats[1:]), len(feats) - 1,
                          _lib.ptr(geo), _lib.ptr(depth), is4d | _lib.call_arith_bits(), _lib.ptr(packed), *[_lib.ptr(s) for s in states],
                          _lib.ptr(target), _lib.ptr(ws), nbytes, b, c, D, h, w, d_begin, d_end,
                          _lib.current_stream(dev))

    def native_volume(self, features, proj_matrices, depth_values, geo_model, use_qc):
        """(B,D,H,W) regularised cost of the whole sweep without materialising the variance volume
        (smvs_red_volume_planes): what RED_Regularization.forward(variance_cost_volume(...)) returns."""
        ref = features[0]
        b, _, h, w = ref.shape
        d_num = depth_values.shape[1]
        states = self.initial_states(b, h, w, ref.device)
        reg = torch.empty((b, d_num, h, w), dtype=torch.float32, device=ref.device)
        self.native_pred_planes(features, proj_matrices, depth_values, geo_model, use_qc, states, None, 0, d_num,
                                reg_volume=reg)
        return reg

    def _use_native(self, cost):
        """Native kernels: GPU tensors, no gradient wanted (whatever train()/eval() says -- the ConvGRU has no
        mode-dependent layers), and a plane the kernels support; anything else takes the PyTorch composite."""
        if SW.force_composite("RED"):        # A/B switch: force the stock PyTorch composite
            return False
        if not cost.is_cuda or cost.dtype is not torch.float32 or self.base_channels != 8 or cost.dim() != 4:
            return False
        if _autograd_needed(cost, _tensors_by_path(self, self._PARAM_ORDER)):
            return False
        b, c, h, w = cost.shape
        return _lib.load().smvs_red_workspace_bytes(b, c, h, w) != 0      # 0: not a multiple of 8, or beyond the kernels' limits

    def step(self, cost, s1, s2, s3, s4):
        """One plane: 2-D encoder/decoder with a ConvGRU at each of the 4 scales (module.py:625-644).

        Inference (no autograd) on the GPU runs the native kernels; training keeps the differentiable
        PyTorch composite below (same parameters)."""
        if self._use_native(cost):
            return self.native_step(cost, s1, s2, s3, s4)
        neg = -cost
        if cost.is_cuda and torch.is_grad_enabled() and SW.train_streams:
            return self._step_level_parallel(neg, s1, s2, s3, s4)
        e1 = self.conv1(neg)
        e2 = self.conv2(e1)
        e3 = self.conv3(e2)
        r4, s4 = self.conv_gru4(e3, s4)
        u3 = self.upconv3(r4)
        r3, s3 = self.conv_gru3(e2, s3)
        u2 = self.upconv2(u3 + r3)
        r2, s2 = self.conv_gru2(e1, s2)
        u1 = self.upconv1(u2 + r2)
        r1, s1 = self.conv_gru1(neg, s1)
        return _conv3x3_cat(self.upconv2d, u1 + r1), s1, s2, s3, s4

    def _step_level_parallel(self, neg, s1, s2, s3, s4):
        """The training step of a plane with the ConvGRU cells of levels 1-3 on side streams: the four cells of a plane do not depend
        on each other (a cell needs its encoder level and its own previous state), and each is a chain of ~10 latency-bound launches
        each way.  Autograd runs a node's backward on the stream of its forward and orders the streams along the graph's edges, so the
        backward overlaps the same way; inside a captured training step (train_graph) the forks and joins become graph branches.
        Tensors that cross streams are registered with the caching allocator (record_stream).  (Streams that PERSIST over the plane loop
        -- encoder running ahead, one recurrent chain per level, decoder on the caller's stream: the inference pipeline's data flow --
        give identical results in eager mode, but hipStreamEndCapture of this image faults on that capture; fork / join per plane it is.)"""
        dev = neg.device
        main = torch.cuda.current_stream(dev)
        side = _side_streams(dev, 3)

        def cell_on(stream, cell, x, state):
            stream.wait_stream(main)                             # x (and the first plane's state) were produced on the main stream
            with torch.cuda.stream(stream):
                r, st = cell(x, state)
            x.record_stream(stream)
            state.record_stream(stream)
            return r, st

        r1, s1 = cell_on(side[0], self.conv_gru1, neg, s1)
        e1 = self.conv1(neg)
        r2, s2 = cell_on(side[1], self.conv_gru2, e1, s2)
        e2 = self.conv2(e1)
        r3, s3 = cell_on(side[2], self.conv_gru3, e2, s3)
        e3 = self.conv3(e2)
        r4, s4 = self.conv_gru4(e3, s4)
        u3 = self.upconv3(r4)
        main.wait_stream(side[2]); r3.record_stream(main)
        u2 = self.upconv2(u3 + r3)
        main.wait_stream(side[1]); r2.record_stream(main)
        u1 = self.upconv1(u2 + r2)
        main.wait_stream(side[0]); r1.record_stream(main)
        return _conv3x3_cat(self.upconv2d, u1 + r1), s1, s2, s3, s4


class RED_Regularization(_REDCore):
    """Whole-volume (train) variant: (B,C,D,H,W) -> (B,D,H,W).  reference: module.py:595-649."""

    def forward(self, volume_variance):
        b, _, d_num, h, w = volume_variance.shape
        s = self.initial_states(b, h, w, volume_variance.device)
        outs = []
        planes = self._per_plane_parameters(d_num) if torch.is_grad_enabled() else None
        if planes is not None and volume_variance.is_cuda and SW.train_streams:
            self._prepack_on_callers_stream()
        if planes is None:
            for d in range(d_num):
                reg, *s = self.step(volume_variance[:, :, d], *s)
                outs.append(reg)
            return torch.stack(outs, dim=1).squeeze(2)
        per_plane = sum(((m.weight.numel() + (m.weight.shape[0] if m.bias is not None else 0)) + 3) & ~3
                        for m in self.modules() if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and m.kernel_size == (3, 3))
        sink = planes.pop("", None)
        if sink is not None:                                     # weight gradients deferred to one launch per layer (_WgradSink)
            with sink:
                if SW.train_streams and SW.train_loop_pipeline and volume_variance.is_cuda:
                    outs = self._planes_software_pipelined(volume_variance, s, planes, d_num)
                else:
                    for d in range(d_num):
                        with _reparametrize(self, {n: t[d] for n, t in planes.items()}):
                            reg, *s = self.step(volume_variance[:, :, d], *s)
                        outs.append(reg)
            return torch.stack(outs, dim=1).squeeze(2)
        with _WgradArena(per_plane * d_num if volume_variance.is_cuda else 0, volume_variance.device):
            for d in range(d_num):
                with _reparametrize(self, {n: t[d] for n, t in planes.items()}):
                    reg, *s = self.step(volume_variance[:, :, d], *s)
                outs.append(reg)
        return torch.stack(outs, dim=1).squeeze(2)

    def _prepack_on_callers_stream(self):
        """Kernel-layout copies of every 3x3 weight the native training path will ask for (forward AND input-gradient layouts), packed on
        the caller's stream BEFORE the plane loop forks its side streams: the pack kernels then precede every consumer on every stream
        -- and, inside a captured training step, every branch of the graph -- through the fork itself, whichever stream asks
        train_fns._conv_packed first (the cache has no stream bookkeeping of its own).  A hit costs a dictionary lookup."""
        for m in self.modules():
            kind = _native_kind(m) if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) else None
            w = getattr(m, "weight", None)
            if kind is None or w is None or not w.is_cuda or w.dtype is not torch.float32 or not w.is_contiguous() or not w.requires_grad:
                continue
            _, flay, _, blay, _ = _NATIVE_KINDS[kind]
            cin, cout = (w.shape[0], w.shape[1]) if kind[0] == "t" else (w.shape[1], w.shape[0])
            _conv_packed(w, flay, cin, cout)
            _conv_packed(w, blay, cout, cin)

    def _planes_software_pipelined(self, volume, s, planes, d_num):
        """The training loop software-pipelined over the planes: in iteration d the ConvGRU cells of plane d (one side stream per level),
        the encoder of plane d+1 (a fifth side stream) and the decoder of plane d-1 (the caller's stream) run side by side; every side
        stream is forked from the caller's stream and joined back inside the iteration (streams that persist over the loop fault in
        hipStreamEndCapture on this image).  Same autograd graph as the plain loop -- the nodes only run on other streams, and
        autograd's backward, which runs a node on the stream of its forward, overlaps the same way.  The chain of an iteration is
        max(cells, decoder, encoder) = ~10 dependent launches instead of ~20."""
        dev = volume.device
        main = torch.cuda.current_stream(dev)
        side = _side_streams(dev, 5)
        enc_stream, cell_streams = side[0], side[1:5]
        cells = (self.conv_gru1, self.conv_gru2, self.conv_gru3, self.conv_gru4)
        s = list(s)

        def params(d):
            return _reparametrize(self, {n: t[d] for n, t in planes.items()})

        def encoder(cost):
            neg = -cost
            e1 = self.conv1(neg)
            e2 = self.conv2(e1)
            return neg, e1, e2, self.conv3(e2)

        def decoder(r1, r2, r3, r4):
            u3 = self.upconv3(r4)
            u2 = self.upconv2(u3 + r3)
            u1 = self.upconv1(u2 + r2)
            return _conv3x3_cat(self.upconv2d, u1 + r1)

        outs, pending = [], None
        with params(0):
            xs = encoder(volume[:, :, 0])
        for d in range(d_num):
            rs = []
            with params(d):
                for k in range(4):
                    cell_streams[k].wait_stream(main)
                    with torch.cuda.stream(cell_streams[k]):
                        r, s[k] = cells[k](xs[k], s[k])
                    xs[k].record_stream(cell_streams[k])
                    if d == 0:
                        s[k].record_stream(main)                 # (the final states go back to the caller)
                    rs.append(r)
            nxt = None
            if d + 1 < d_num:
                enc_stream.wait_stream(main)
                with torch.cuda.stream(enc_stream), params(d + 1):
                    nxt = encoder(volume[:, :, d + 1])
                volume.record_stream(enc_stream)
            if pending is not None:
                with params(d - 1):
                    outs.append(decoder(*pending))
            for k in range(4):
                main.wait_stream(cell_streams[k])
                rs[k].record_stream(main)
            if nxt is not None:
                main.wait_stream(enc_stream)
                for t in nxt:
                    t.record_stream(main)
            pending, xs = rs, nxt
        with params(d_num - 1):
            outs.append(decoder(*pending))
        return outs

    def _per_plane_parameters(self, d_num):
        """Training: every plane of the loop uses its own VIEW of each parameter (`p.expand(D, ...)` unbound along D: same
        memory, same values).  Autograd then delivers the D per-plane gradients of a parameter to ONE stack + ONE sum over D
        (UnbindBackward, ExpandBackward) instead of D-1 `AccumulateGrad` additions per parameter: ~6 000 add launches per
        training step of the 48/32/8 cascade become ~140 (round 4: graphed step 126 -> see profiles/r04_train_step.txt).
        Forward values are untouched; the gradient is the same sum in another association.  SMVS_TRAIN_PLANE_VIEWS=0 keeps the
        plain loop."""
        if _reparametrize is None or d_num < 2 or not SW.train_plane_views:
            return None
        named = [(n, p) for n, p in self.named_parameters() if p.requires_grad]
        if not named:
            return None
        if not SW.train_defer_wgrad or not named[0][1].is_cuda or (SW.train_composite_mask & 8):
            return {n: torch.unbind(p.unsqueeze(0).expand(d_num, *p.shape), 0) for n, p in named}
        sink = _WgradSink()
        views = {n: _PlaneViewsFn.apply(p, d_num, sink) for n, p in named}
        views[""] = sink                                         # (popped by forward)
        return views


class slice_RED_Regularization(_REDCore):
    """Plane-at-a-time (pred) variant with explicit state.  reference: module.py:653-693."""

    def forward(self, cost, state1, state2, state3, state4):
        return self.step(cost, state1, state2, state3, state4)


# ---- 3-D conv regulariser (casmvs / ucs) ----------------------------------------------------------------
class CostRegNet(nn.Module):
    """reference: module.py:546-577."""

    def __init__(self, in_channels, base_channels):
        super().__init__()
        self.conv0 = Conv3d(in_channels, base_channels, padding=1)
        self.conv1 = Conv3d(base_channels, base_channels * 2, stride=2, padding=1)
        self.conv2 = Conv3d(base_channels * 2, base_channels * 2, padding=1)
        self.conv3 = Conv3d(base_channels * 2, base_channels * 4, stride=2, padding=1)
        self.conv4 = Conv3d(base_channels * 4, base_channels * 4, padding=1)
        self.conv5 = Conv3d(base_channels * 4, base_channels * 8, stride=2, padding=1)
        self.conv6 = Conv3d(base_channels * 8, base_channels * 8, padding=1)
        self.conv7 = Deconv3d(base_channels * 8, base_channels * 4, stride=2, padding=1, output_padding=1)
        self.conv9 = Deconv3d(base_channels * 4, base_channels * 2, stride=2, padding=1, output_padding=1)
        self.conv11 = Deconv3d(base_channels * 2, base_channels * 1, stride=2, padding=1, output_padding=1)
        self.prob = nn.Conv3d(base_channels, 1, 3, stride=1, padding=1, bias=False)

    _LAYERS = ["conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11"]

    def _names(self):
        return [l + s for l in self._LAYERS for s in (".conv.weight", ".bn.weight", ".bn.bias", ".bn.running_mean",
                                                      ".bn.running_var")] + ["prob.weight"]

    def _packed_weights(self, device):
        tensors = _tensors_by_path(self, self._names())
        in_ch = tensors[0].shape[1]

        def build():
            lib = _lib.load()
            packed = torch.empty(lib.smvs_costreg_packed_floats(in_ch), dtype=torch.float32, device=device)
            src = [t.detach().to(device=device, dtype=torch.float32).contiguous() for t in tensors]
            with torch.cuda.device(device):
                _lib.call("smvs_costreg_pack_weights", _lib.ptr_array(src), in_ch, _lib.ptr(packed),
                          _lib.current_stream(device))
            return packed
        return _packed("costreg", device, tensors, build), in_ch

    def _use_native(self, x):
        """Native kernels: GPU, eval mode (BatchNorm3d folded from its running statistics), no gradient wanted,
        and a volume the kernels support; anything else takes the PyTorch composite below."""
        if SW.force_composite("COSTREG"):     # A/B switch: force the stock PyTorch composite
            return False
        if not x.is_cuda or self.training or x.dim() != 5 or self.conv0.conv.out_channels != 8:
            return False
        if _autograd_needed(x, _tensors_by_path(self, self._names())):
            return False
        b, c, d, h, w = x.shape
        return _lib.load().smvs_costreg_workspace_bytes(b, c, d, h, w) != 0

    def native_forward(self, x):
        """(B,C,D,H,W) -> (B,1,D,H,W) through smvs_costreg_fwd (HIP, inference-form BatchNorm)."""
        dev = _lib.require_device(x)
        packed, in_ch = self._packed_weights(dev)
        x = x.detach().to(torch.float32).contiguous()
        b, c, d, h, w = x.shape
        if c != in_ch:
            raise ValueError("volume has %d channels, regulariser expects %d" % (c, in_ch))
        lib = _lib.load()
        nbytes = lib.smvs_costreg_workspace_bytes(b, c, d, h, w)
        if nbytes == 0:
            raise ValueError("volume %s is not a positive multiple of 8 in D, H, W" % (tuple(x.shape),))
        ws = _workspace(self, "_ws", nbytes, dev)
        out = torch.empty((b, 1, d, h, w), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("smvs_costreg_fwd", _lib.ptr(packed), _lib.ptr(x), _lib.ptr(out), _lib.ptr(ws), nbytes,
                      b, c, d, h, w, _lib.current_stream(dev))
        return out

    def forward(self, x):
        if self._use_native(x):
            return self.native_forward(x)
        if x.is_cuda and self.training and torch.is_grad_enabled() and not (SW.train_composite_mask & 64):
            # one zero fill for everything the native backward accumulates into: the 11 weight gradients and the BatchNorm sums
            n = sum(((p.numel() + 3) & ~3) for k, p in self.named_parameters() if k.endswith("conv.weight") or k == "prob.weight")
            n += sum(8 * m.num_features + 8 for m in self.modules() if isinstance(m, nn.BatchNorm3d))
            with _WgradArena(n, x.device):
                return self._composite(x)
        return self._composite(x)

    def _composite(self, x):
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        c4 = self.conv4(self.conv3(c2))
        x = self.conv6(self.conv5(c4))
        x = c4 + self.conv7(x)
        x = c2 + self.conv9(x)
        x = c0 + self.conv11(x)
        return _conv3d(self.prob, x)


# ---- feature extractor ------------------------------------------------------------------------------------
class FeatureNet(nn.Module):
    """2-D U-Net / FPN producing stage1..3 features with 32/16/8 channels.  reference: module.py:442-543."""

    def __init__(self, base_channels, num_stage=3, stride=4, arch_mode="unet"):
        super().__init__()
        assert arch_mode in ("unet", "fpn"), "mode must be in 'unet' or 'fpn', but get:{}".format(arch_mode)
        self.arch_mode, self.stride, self.base_channels, self.num_stage = arch_mode, stride, base_channels, num_stage
        c = base_channels
        self.conv0 = nn.Sequential(Conv2d(3, c, 3, 1, padding=1), Conv2d(c, c, 3, 1, padding=1))
        self.conv1 = nn.Sequential(Conv2d(c, c * 2, 5, stride=2, padding=2), Conv2d(c * 2, c * 2, 3, 1, padding=1),
                                   Conv2d(c * 2, c * 2, 3, 1, padding=1))
        self.conv2 = nn.Sequential(Conv2d(c * 2, c * 4, 5, stride=2, padding=2), Conv2d(c * 4, c * 4, 3, 1, padding=1),
                                   Conv2d(c * 4, c * 4, 3, 1, padding=1))
        self.out1 = nn.Conv2d(c * 4, c * 4, 1, bias=False)
        self.out_channels = [4 * c]
        if arch_mode == "unet":
            if num_stage == 3:
                self.deconv1 = DeConv2dFuse(c * 4, c * 2, 3)
                self.deconv2 = DeConv2dFuse(c * 2, c, 3)
                self.out2 = nn.Conv2d(c * 2, c * 2, 1, bias=False)
                self.out3 = nn.Conv2d(c, c, 1, bias=False)
                self.out_channels += [2 * c, c]
            elif num_stage == 2:
                self.deconv1 = DeConv2dFuse(c * 4, c * 2, 3)
                self.out2 = nn.Conv2d(c * 2, c * 2, 1, bias=False)
                self.out_channels.append(2 * c)
        else:
            final = c * 4
            if num_stage == 3:
                self.inner1 = nn.Conv2d(c * 2, final, 1, bias=True)
                self.inner2 = nn.Conv2d(c, final, 1, bias=True)
                self.out2 = nn.Conv2d(final, c * 2, 3, padding=1, bias=False)
                self.out3 = nn.Conv2d(final, c, 3, padding=1, bias=False)
                self.out_channels += [2 * c, c]
            elif num_stage == 2:
                self.inner1 = nn.Conv2d(c * 2, final, 1, bias=True)
                self.out2 = nn.Conv2d(final, c, 3, padding=1, bias=False)
                self.out_channels.append(c)

    # ---- native inference path (smvs_featnet_fwd) ---------------------------------------------------------
    _TRUNK = ("conv0.0", "conv0.1", "conv1.0", "conv1.1", "conv1.2", "conv2.0", "conv2.1", "conv2.2")
    _UNET_BLOCKS = ("deconv1.deconv", "deconv1.conv", "deconv2.deconv", "deconv2.conv")
    _BN5 = (".conv.weight", ".bn.weight", ".bn.bias", ".bn.running_mean", ".bn.running_var")

    def _arch(self):
        return 0 if self.arch_mode == "unet" else 1

    def _names(self):
        names = [b + s for b in self._TRUNK for s in self._BN5]
        if self.arch_mode == "unet":
            return names + [b + s for b in self._UNET_BLOCKS for s in self._BN5] + ["out1.weight", "out2.weight", "out3.weight"]
        return names + ["out1.weight", "inner1.weight", "inner1.bias", "out2.weight", "inner2.weight", "inner2.bias",
                        "out3.weight"]

    def _packed_weights(self, device):
        """Parameters + BatchNorm running statistics repacked for the HIP kernels; cached until any of them
        changes (load_state_dict / an optimiser step / a training-mode forward bump the tensor versions)."""
        tensors = _tensors_by_path(self, self._names())

        def build():
            lib = _lib.load()
            packed = torch.empty(lib.smvs_featnet_packed_floats(self.base_channels, self._arch()), dtype=torch.float32,
                                 device=device)
            src = [t.detach().to(device=device, dtype=torch.float32).contiguous() for t in tensors]
            with torch.cuda.device(device):
                _lib.call("smvs_featnet_pack_weights", _lib.ptr_array(src), self.base_channels, self._arch(),
                          _lib.ptr(packed), _lib.current_stream(device))
            return packed
        return _packed("featnet%d" % self._arch(), device, tensors, build)

    def _use_native(self, x):
        """Native kernels: GPU, eval mode (BatchNorm folded from running statistics), no gradient wanted, and
        an image size the kernels support; anything else takes the PyTorch composite."""
        if SW.force_composite("FEATNET"):     # A/B switch: force the stock PyTorch composite
            return False
        if not x.is_cuda or self.training or self.num_stage != 3 or self.base_channels > 16 or x.dim() < 4:
            return False
        if _autograd_needed(x, _tensors_by_path(self, self._names())):
            return False
        n = 1
        for d in x.shape[:-3]:
            n *= d
        return _lib.load().smvs_featnet_workspace_bytes(n, x.shape[-2], x.shape[-1], self.base_channels, self._arch()) != 0

    def native_forward(self, x):
        """(N,3,H,W) -> {"stage1": (N,4c,H/4,W/4), "stage2": (N,2c,H/2,W/2), "stage3": (N,c,H,W)} in one
        native call (15 launches for all N images)."""
        dev = _lib.require_device(x)
        packed = self._packed_weights(dev)
        x = x.detach().to(torch.float32).contiguous()
        n, ch, h, w = x.shape
        if ch != 3:
            raise ValueError("FeatureNet expects 3-channel images, got %d" % ch)
        c = self.base_channels
        lib = _lib.load()
        nbytes = lib.smvs_featnet_workspace_bytes(n, h, w, c, self._arch())
        if nbytes == 0:
            raise ValueError("image %dx%d is not a positive multiple of 4 in both dimensions" % (h, w))
        ws = _workspace(self, "_ws", nbytes, dev)
        s1 = torch.empty((n, 4 * c, h // 4, w // 4), dtype=torch.float32, device=dev)
        s2 = torch.empty((n, 2 * c, h // 2, w // 2), dtype=torch.float32, device=dev)
        s3 = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("smvs_featnet_fwd", _lib.ptr(packed), _lib.ptr(x), _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(s3),
                      _lib.ptr(ws), nbytes, n, h, w, c, self._arch(), _lib.current_stream(dev))
        return {"stage1": s1, "stage2": s2, "stage3": s3}

    def forward_views(self, imgs):
        """imgs (B,V,3,H,W) -> list over views of {"stageK": (B,C,h,w)} (what the networks' per-view loop
        `[self.feature(imgs[:, v]) ...]` produces, casred.py:116-121).  Native path: all B*V images in one call."""
        b, v = imgs.shape[:2]
        if self._use_native(imgs):
            out = self.native_forward(imgs.transpose(0, 1).reshape(v * b, *imgs.shape[2:]))
            return [{k: t[i * b:(i + 1) * b] for k, t in out.items()} for i in range(v)]
        return [self(imgs[:, i]) for i in range(v)]

    def forward(self, x):
        if self._use_native(x):
            return self.native_forward(x)
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        feat = c2
        outs = {"stage1": self.out1(feat)}
        if self.arch_mode == "unet":
            if self.num_stage >= 2:
                feat = self.deconv1(c1, feat)
                outs["stage2"] = self.out2(feat)
            if self.num_stage == 3:
                feat = self.deconv2(c0, feat)
                outs["stage3"] = self.out3(feat)
        else:
            if self.num_stage >= 2:
                feat = F.interpolate(feat, scale_factor=2, mode="nearest") + self.inner1(c1)
                outs["stage2"] = self.out2(feat)
            if self.num_stage == 3:
                feat = F.interpolate(feat, scale_factor=2, mode="nearest") + self.inner2(c0)
                outs["stage3"] = self.out3(feat)
        return outs


# ---- regression ---------------------------------------------------------------------------------------------
def depth_regression(p, depth_values):
    """sum_D p * depth_values; differentiable torch composite.  reference: module.py:433-439."""
    if depth_values.dim() <= 2:
        depth_values = depth_values.view(*depth_values.shape, 1, 1)
    else:
        depth_values = F.interpolate(depth_values, [p.shape[2], p.shape[3]], mode="bilinear", align_corners=False)
    return torch.sum(p * depth_values, 1)


def softmax_depth_regression(reg, depth_values):
    """softmax over D + expected height + max probability in one HIP kernel (no_grad paths).

    reg (B,D,H,W) float32; depth_values (B,D) or (B,D,H,W).  Returns (depth, confidence), each
    (B,H,W).  Equals F.softmax(reg,1) -> depth_regression -> max(1) of networks/casred.py:58-62.
    With autograd enabled on `reg` the torch composite is used instead (it is differentiable).
    """
    from .depth_range import GeneratedHeights
    gen = depth_values if isinstance(depth_values, GeneratedHeights) else None
    if torch.is_grad_enabled() and reg.requires_grad:
        p = F.softmax(reg, dim=1)
        return depth_regression(p, gen.materialize() if gen is not None else depth_values), p.max(1)[0]
    if gen is not None:
        import ctypes
        dev = _lib.require_device(reg, gen.prev)
        r = reg.detach().to(torch.float32).contiguous()
        B, D, H, W = r.shape
        depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        conf = torch.empty_like(depth)
        gs = gen.c_struct()
        with torch.cuda.device(dev):
            _lib.call("smvs_softmax_regress_fwd_gen", _lib.ptr(r), ctypes.addressof(gs), _lib.ptr(depth), _lib.ptr(conf),
                      B, D, H, W, _lib.current_stream(dev))
        return depth, conf
    dev = _lib.require_device(reg, depth_values)
    r = reg.detach().to(torch.float32).contiguous()
    B, D, H, W = r.shape
    dv = depth_values.detach().to(torch.float32).contiguous()
    is4d = 1 if dv.dim() == 4 else 0
    if is4d and tuple(dv.shape) != (B, D, H, W):
        dv = F.interpolate(dv, [H, W], mode="bilinear", align_corners=False).contiguous()
    depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    conf = torch.empty_like(depth)
    with torch.cuda.device(dev):
        _lib.call("smvs_softmax_regress_fwd", _lib.ptr(r), _lib.ptr(dv), is4d, _lib.ptr(depth), _lib.ptr(conf),
                  B, D, H, W, _lib.current_stream(dev))
    return depth, conf


def window_depth_regression(reg, depth_values, lamb=None):
    """CascadeMVSNet / UCSNet regression in one HIP kernel (no_grad paths): softmax over D, expected height,
    photometric confidence = probability mass of the 4 hypotheses around the expected index
    (networks/casmvs.py:66-74) and, with `lamb`, UCSNet's lamb * sqrt(sum p (h - depth)^2) (networks/ucs.py:73-74).

    reg (B,D,H,W); depth_values (B,D) or (B,D,H,W).  Returns (depth, confidence) or (depth, confidence, variance).
    With autograd enabled on `reg` the torch composite runs instead (differentiable, same operations)."""
    from .depth_range import GeneratedHeights
    gen = depth_values if isinstance(depth_values, GeneratedHeights) else None
    if (torch.is_grad_enabled() and reg.requires_grad) or not reg.is_cuda:
        p = F.softmax(reg, dim=1)
        num_depth = reg.shape[1]
        dv = gen.materialize() if gen is not None else depth_values
        depth = depth_regression(p, depth_values=dv)
        with torch.no_grad():
            sum4 = 4 * F.avg_pool3d(F.pad(p.unsqueeze(1), pad=(0, 0, 0, 0, 1, 2)), (4, 1, 1), stride=1, padding=0).squeeze(1)
            idx = depth_regression(p, depth_values=torch.arange(num_depth, device=p.device, dtype=torch.float)).long()
            conf = torch.gather(sum4, 1, idx.clamp(min=0, max=num_depth - 1).unsqueeze(1)).squeeze(1)
        if lamb is None:
            return depth, conf
        if dv.dim() == 2:
            dv = dv.view(*dv.shape, 1, 1)
        return depth, conf, lamb * torch.sum((dv - depth.unsqueeze(1)) ** 2 * p, dim=1) ** 0.5
    if gen is not None:
        import ctypes
        dev = _lib.require_device(reg, gen.prev)
        r = reg.detach().to(torch.float32).contiguous()
        B, D, H, W = r.shape
        depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        conf = torch.empty_like(depth)
        var = torch.empty_like(depth) if lamb is not None else None
        gs = gen.c_struct()
        with torch.cuda.device(dev):
            _lib.call("smvs_window_regress_fwd_gen", _lib.ptr(r), ctypes.addressof(gs), _lib.ptr(depth), _lib.ptr(conf),
                      _lib.ptr(var) if var is not None else None, float(lamb or 0.0), B, D, H, W, _lib.current_stream(dev))
        return (depth, conf) if lamb is None else (depth, conf, var)
    dev = _lib.require_device(reg, depth_values)
    r = reg.detach().to(torch.float32).contiguous()
    B, D, H, W = r.shape
    dv = depth_values.detach().to(torch.float32).contiguous()
    is4d = 1 if dv.dim() == 4 else 0
    if is4d and tuple(dv.shape) != (B, D, H, W):
        dv = F.interpolate(dv, [H, W], mode="bilinear", align_corners=False).contiguous()
    depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    conf = torch.empty_like(depth)
    var = torch.empty_like(depth) if lamb is not None else None
    with torch.cuda.device(dev):
        _lib.call("smvs_window_regress_fwd", _lib.ptr(r), _lib.ptr(dv), is4d, _lib.ptr(depth), _lib.ptr(conf),
                  _lib.ptr(var) if var is not None else None, float(lamb or 0.0), B, D, H, W, _lib.current_stream(dev))
    return (depth, conf) if lamb is None else (depth, conf, var)


class StreamingRegression:
    """The float64 accumulators of the pred loop (networks/casred.py:182-184, 218-236) on the device.

    step(reg_plane (B,1,H,W)|(B,H,W), depth_values, d) folds plane d in; result() -> (depth, conf).
    `state` (3,B,H,W) float64 = [exp_sum, depth_img, max_prob]: additive (sum, sum, max) over
    planes, which is what a depth-sharded run all-reduces (satmvs_amd/shard.py).
    """

    def __init__(self, B, H, W, device):
        self.B, self.H, self.W = B, H, W
        self.state = torch.zeros((3, B, H, W), dtype=torch.float64, device=device)

    def step(self, reg_plane, depth_values, d):
        dev = _lib.require_device(reg_plane, depth_values, self.state)
        r = reg_plane.detach().to(torch.float32).contiguous()
        dv = depth_values.detach().to(torch.float32).contiguous()
        is4d = 1 if dv.dim() == 4 else 0
        D = dv.shape[1]
        with torch.cuda.device(dev):
            _lib.call("smvs_stream_regress_step", _lib.ptr(r), _lib.ptr(dv), is4d, _lib.ptr(self.state[0]),
                      _lib.ptr(self.state[1]), _lib.ptr(self.state[2]), self.B, D, self.H, self.W, d,
                      _lib.current_stream(dev))

    def result(self):
        dev = self.state.device
        depth = torch.empty((self.B, self.H, self.W), dtype=torch.float32, device=dev)
        conf = torch.empty_like(depth)
        with torch.cuda.device(dev):
            _lib.call("smvs_stream_regress_final", _lib.ptr(self.state[0]), _lib.ptr(self.state[1]),
                      _lib.ptr(self.state[2]), _lib.ptr(depth), _lib.ptr(conf), self.B * self.H * self.W,
                      _lib.current_stream(dev))
        return depth, conf
