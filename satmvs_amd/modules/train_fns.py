"""Training path of the RED regulariser: the autograd functions behind satmvs_amd/modules/module.py's ConvGRUCell2 / RED_Regularization.

Native under autograd (C entry points in include/satmvs.h; kernels in csrc/groupnorm.hip, conv_wgrad.hip, red.hip, mfma_conv.h):
GroupNorm(1, C) + gate activation (_GroupNorm1Fn, _GroupNormPairFn, GroupNorm1), the cell's element-wise steps (_GruMulCatFn,
_GruBlendFn), every 3x3 layer of the regulariser -- forward, input gradient, weight / bias gradient (_Conv3x3NativeFn,
_Conv3x3WgradFn, _conv3x3_cat, _conv3x3), weight gradients deferred to one launch per layer over all planes (_WgradArena,
_WgradSink, _PlaneViewsFn) and a whole ConvGRU cell as one autograd node (_ConvGRUCellFn).  Reference: /root/reference/modules/
module.py:6-58 (ConvGRUCell2), :595-649 (RED_Regularization) under /root/reference/train.py:279-285 (loss.backward()).
DESIGN.md section 8 tells what each of them bought; satmvs_amd/modules/switches.py holds the A/B switches they honour.
"""
from __future__ import annotations

import threading

import torch
from torch.autograd.function import once_differentiable
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .switches import SW


_PARAM_EPOCH = [0]      # bumped by whoever changes parameters behind autograd's version counters (train_graph: a graph replay runs the
                        # optimizer step on the device without touching Python), so that no kernel-layout copy outlives the weights


def bump_param_epoch():
    _PARAM_EPOCH[0] += 1


_SIDE_STREAMS = {}


def _side_streams(device, n):
    key = (device.index, threading.get_ident())
    ss = _SIDE_STREAMS.get(key)
    if ss is None or len(ss) < n:
        ss = _SIDE_STREAMS[key] = [torch.cuda.Stream(device) for _ in range(n)]
    return ss


def _gn_scratch(dev, doubles):
    """Scratch of the GroupNorm kernels (float64 partial sums, written by one kernel and folded by the next on the same stream):
    a fresh torch allocation per call.  The caching allocator makes that cheap in eager mode, keeps a block that queued kernels
    still use from being handed out on another stream, and inside a HIP-graph capture takes it from that graph's own pool -- no
    buffer outlives the graph it was captured in."""
    return torch.empty((max(64, doubles),), dtype=torch.float64, device=dev)


def _f32c_fast(t):
    """float32 contiguous tensor with the cheapest possible host path (these run ~10^4 times per eager training step)."""
    return t if (t.dtype is torch.float32 and t.is_contiguous()) else t.float().contiguous()


class _GroupNorm1Fn(torch.autograd.Function):
    """act(GroupNorm(1, C)(x)) through smvs_groupnorm1_fwd / _bwd (csrc/groupnorm.hip).  x may be a channel slice of a
    wider tensor (the gate halves of the 2C-channel gate convolution): only its batch stride has to be regular."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, act):
        dev = _lib.require_device(x, weight, bias)
        B, C, H, W = x.shape
        if x.dtype != torch.float32 or x.stride(1) != H * W or x.stride(2) != W or x.stride(3) != 1 or (B > 1 and x.stride(0) < C * H * W):
            x = x.float().contiguous()
        xbs = x.stride(0) if B > 1 else C * H * W
        w, b = _f32c_fast(weight.detach()), _f32c_fast(bias.detach())
        y = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
        stats = torch.empty((B, 2), dtype=torch.float32, device=dev)
        ws = _gn_scratch(dev, 2 * B * ((C * H * W + 4095) // 4096))
        with torch.cuda.device(dev):
            _lib.call("smvs_groupnorm1_fwd", _lib.ptr(x), xbs, _lib.ptr(w), _lib.ptr(b), float(eps), int(act), _lib.ptr(y),
                      _lib.ptr(stats), _lib.ptr(ws), B, C, H * W, _lib.current_stream(dev))
        ctx.save_for_backward(x, w, y, stats)
        ctx.meta = (xbs, int(act))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y, stats = ctx.saved_tensors
        xbs, act = ctx.meta
        B, C, H, W = y.shape
        dev = y.device
        dy = _f32c_fast(dy)
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
        dg = torch.empty((C,), dtype=torch.float32, device=dev)
        db = torch.empty((C,), dtype=torch.float32, device=dev)
        ws = _gn_scratch(dev, 2 * B * C * ((H * W + 4095) // 4096))
        with torch.cuda.device(dev):
            _lib.call("smvs_groupnorm1_bwd", _lib.ptr(dy), _lib.ptr(x), xbs, _lib.ptr(y), _lib.ptr(w), _lib.ptr(stats), act,
                      _lib.ptr(dx), C * H * W, _lib.ptr(dg), _lib.ptr(db), _lib.ptr(ws), B, C, H * W, _lib.current_stream(dev))
        return dx, dg, db, None, None


class _GroupNormPairFn(torch.autograd.Function):
    """act(GroupNorm(1, C)) of BOTH halves of the gate convolution's output (B, 2C, H, W), each half with its own affine
    parameters, in one native call each way (smvs_groupnorm1_pair_fwd / _bwd; reference: module.py:34-40)."""

    @staticmethod
    def forward(ctx, gates, w1, b1, w2, b2, eps, act):
        dev = _lib.require_device(gates, w1, b1, w2, b2)
        x = _f32c_fast(gates)
        B, C2, H, W = x.shape
        C = C2 // 2
        ws_ = [_f32c_fast(t.detach()) for t in (w1, b1, w2, b2)]
        y = torch.empty_like(x)
        stats = torch.empty((2 * B, 2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("smvs_groupnorm1_pair_fwd", _lib.ptr(x), _lib.ptr(ws_[0]), _lib.ptr(ws_[1]), _lib.ptr(ws_[2]), _lib.ptr(ws_[3]), float(eps),
                      int(act), _lib.ptr(y), _lib.ptr(stats), _lib.ptr(_gn_scratch(dev, 4 * B * ((C * H * W + 4095) // 4096))), B, C, H * W, _lib.current_stream(dev))
        ctx.save_for_backward(x, ws_[0], ws_[2], y, stats)
        ctx.act = int(act)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w1, w2, y, stats = ctx.saved_tensors
        B, C2, H, W = x.shape
        C = C2 // 2
        dev = x.device
        dy = _f32c_fast(dy)
        dx = torch.empty_like(x)
        g = torch.empty((4, C), dtype=torch.float32, device=dev)              # dgamma, dbeta, dgamma2, dbeta2
        with torch.cuda.device(dev):
            _lib.call("smvs_groupnorm1_pair_bwd", _lib.ptr(dy), _lib.ptr(x), _lib.ptr(y), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(stats), ctx.act,
                      _lib.ptr(dx), _lib.ptr(g[0]), _lib.ptr(g[1]), _lib.ptr(g[2]), _lib.ptr(g[3]), _lib.ptr(_gn_scratch(dev, 4 * B * C * ((H * W + 4095) // 4096))),
                      B, C, H * W, _lib.current_stream(dev))
        return dx, g[0], g[1], g[2], g[3], None, None


class _GruMulCatFn(torch.autograd.Function):
    """cat((x, r * h), 1) in one launch each way (reference: module.py:43-44)."""

    @staticmethod
    def forward(ctx, x, r, h):
        dev = _lib.require_device(x, r, h)
        x, r, h = _f32c_fast(x), _f32c_fast(r), _f32c_fast(h)
        B, Cx, H, W = x.shape
        Ch = h.shape[1]
        out = torch.empty((B, Cx + Ch, H, W), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("smvs_gru_mul_cat_fwd", _lib.ptr(x), _lib.ptr(r), _lib.ptr(h), _lib.ptr(out), B, Cx, Ch, H * W, _lib.current_stream(dev))
        ctx.save_for_backward(r, h)
        ctx.cx = Cx
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dcat):
        r, h = ctx.saved_tensors
        B, Ch, H, W = h.shape
        dcat = _f32c_fast(dcat)
        dr, dh = torch.empty_like(r), torch.empty_like(h)
        with torch.cuda.device(h.device):
            _lib.call("smvs_gru_mul_cat_bwd", _lib.ptr(dcat), _lib.ptr(r), _lib.ptr(h), _lib.ptr(dr), _lib.ptr(dh), B, ctx.cx, Ch, H * W,
                      _lib.current_stream(h.device))
        return dcat[:, :ctx.cx], dr, dh


class _GruBlendFn(torch.autograd.Function):
    """u * h + (1 - u) * y in one launch each way (reference: module.py:57)."""

    @staticmethod
    def forward(ctx, u, h, y):
        dev = _lib.require_device(u, h, y)
        u, h, y = _f32c_fast(u), _f32c_fast(h), _f32c_fast(y)
        out = torch.empty_like(h)
        with torch.cuda.device(dev):
            _lib.call("smvs_gru_blend_fwd", _lib.ptr(u), _lib.ptr(h), _lib.ptr(y), _lib.ptr(out), h.numel(), _lib.current_stream(dev))
        ctx.save_for_backward(u, h, y)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        u, h, y = ctx.saved_tensors
        dy = _f32c_fast(dy)
        du, dh, dc = torch.empty_like(u), torch.empty_like(h), torch.empty_like(y)
        with torch.cuda.device(h.device):
            _lib.call("smvs_gru_blend_bwd", _lib.ptr(dy), _lib.ptr(u), _lib.ptr(h), _lib.ptr(y), _lib.ptr(du), _lib.ptr(dh), _lib.ptr(dc), h.numel(),
                      _lib.current_stream(h.device))
        return du, dh, dc


try:                                                     # torch's own functional-module context manager (private, stable since 2.0)
    from torch.nn.utils.stateless import _reparametrize_module as _reparametrize
except Exception:                                        # pragma: no cover
    _reparametrize = None


class GroupNorm1(nn.GroupNorm):
    """nn.GroupNorm(1, C, eps) -- same parameters, same state_dict keys (reference: module.py:15-20) -- whose forward can take
    the gate's activation along ("sigmoid" / "tanh").  On the GPU with gradients enabled it runs the native kernels: with one
    group a sample is a single row for torch's RowwiseMoments / ComputeInternalGradients kernels (one workgroup each: 45 % of
    the training step's kernel time, profiles/r03_train_step.txt).  CPU tensors take torch's composite."""

    _ACT = {None: 0, "sigmoid": 1, "tanh": 2}

    def forward(self, x, act=None):
        if (x.is_cuda and x.dtype is torch.float32 and x.dim() == 4 and self.num_groups == 1 and self.affine
                and x.shape[0] * x.shape[1] <= 65535 and not (SW.train_composite_mask & 1)):
            return _GroupNorm1Fn.apply(x, self.weight, self.bias, self.eps, self._ACT[act])
        y = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        return torch.sigmoid(y) if act == "sigmoid" else torch.tanh(y) if act == "tanh" else y


_TLS = threading.local()     # the active _WgradArena / _WgradSink of THIS thread (nn.DataParallel runs its replicas on one thread per device)


class _WgradArena:
    """Zero-initialised gradient buffers for one training forward of a regulariser: ONE fill per forward instead of one per
    convolution and plane (the native weight-gradient kernel accumulates with atomics into zeroed memory; ~800 fills per training
    step of the 48/32/8 cascade).  A fresh tensor per forward: nothing aliases across steps; each convolution call takes its slice at
    forward time and its backward writes there once."""
    def __init__(self, floats, device):
        self.buf = torch.zeros((floats,), dtype=torch.float32, device=device)
        self.off = 0

    def take(self, n):
        n4 = (n + 3) & ~3
        if self.off + n4 > self.buf.numel():
            return None
        v = self.buf[self.off:self.off + n]
        self.off += n4
        return v

    def __enter__(self):
        self.prev, _TLS.arena = getattr(_TLS, "arena", None), self
        return self

    def __exit__(self, *exc):
        _TLS.arena = self.prev
        return False


_ZERO_SCALAR = {}       # device -> 0-dim float32 zero: the storage every deferred-gradient placeholder expands


def _placeholder(like):
    """Stand-in for a weight gradient that _WgradSink will deliver later: a stride-0 expansion of one shared zero (no kernel, no memory)."""
    z = _ZERO_SCALAR.get(like.device)
    if z is None:
        z = _ZERO_SCALAR[like.device] = torch.zeros((), dtype=torch.float32, device=like.device)
    return z.expand(like.shape)


def _is_placeholder(g):
    z = _ZERO_SCALAR.get(g.device)
    return z is not None and g.data_ptr() == z.data_ptr() and g.numel() > 1 and all(st == 0 for st in g.stride())


class _WgradSink:
    """Weight gradients of one training forward of a regulariser, DEFERRED: the plane loop calls every 3x3 layer once per plane, and a
    weight-gradient launch per call is a 13 us latency floor on these shapes (~1 300 per training step of the 48/32/8 cascade, the
    largest item of the step after the ConvGRU convolutions went native).  With the sink active a layer's backward only records
    (window tensor(s), grid tensor) of its plane and returns a placeholder; when autograd reaches the parameter (_PlaneViewsFn: after
    the last plane) ONE smvs_conv3x3_wgrad_list launch per layer sums over all planes."""
    def __init__(self):
        self.layers = {}        # weight address -> {"entries": [(win, win2, grid)], "meta": ..., "weight_shape": ...}
        self.bias_of = {}       # bias address -> weight address (layers whose bias gradient is the grid tensor's sum)
        self.results = {}       # weight address -> (dw, db or None)

    def __enter__(self):
        self.prev, _TLS.sink = getattr(_TLS, "sink", None), self
        return self

    def __exit__(self, *exc):
        _TLS.sink = self.prev
        return False

    def add(self, weight, bias, win, win2, grid, stride):
        key = weight.data_ptr()
        self.results.pop(key, None)                              # (a second backward through the same graph starts over)
        lay = self.layers.get(key)
        if lay is None:
            lay = self.layers[key] = {"entries": [], "shape": tuple(weight.shape), "stride": stride, "sums": bias is not None}
            if bias is not None:
                self.bias_of[bias.data_ptr()] = key
        lay["entries"].append((win, win2, grid))
        st = torch.cuda.current_stream(win.device)                # the backward of a side-stream cell runs on that stream
        ev = torch.cuda.Event()
        ev.record(st)
        lay.setdefault("events", {})[st.cuda_stream] = ev

    def result(self, key):
        res = self.results.get(key)
        if res is None:
            lay = self.layers.pop(key)
            ent = lay["entries"]
            win0, win20, grid0 = ent[0]
            dev = win0.device
            nw = 1
            for n in lay["shape"]:
                nw *= n
            Bper, CA = win0.shape[0], win0.shape[1]
            CB = win20.shape[1] if win20 is not None else 0
            Cg, H, W = grid0.shape[1], grid0.shape[2], grid0.shape[3]
            buf = torch.zeros((nw + (Cg if lay["sums"] else 0),), dtype=torch.float32, device=dev)
            dw = buf[:nw].view(lay["shape"])
            here = torch.cuda.current_stream(dev)
            for sid, ev in lay.get("events", {}).items():           # the planes' tensors come from the streams their cells ran on
                if sid != here.cuda_stream:
                    here.wait_event(ev)
                    for e_ in ent:
                        for t in e_:
                            if t is not None:
                                t.record_stream(here)
            with torch.cuda.device(dev):
                _lib.call("smvs_conv3x3_wgrad_list", _lib.ptr_array([e[0] for e in ent]),
                          _lib.ptr_array([e[1] for e in ent]) if win20 is not None else None, _lib.ptr_array([e[2] for e in ent]), len(ent),
                          _lib.ptr(dw), _lib.ptr(buf[nw:]) if lay["sums"] else None, Bper, CA, CB, Cg, H, W, lay["stride"],
                          _lib.current_stream(dev))
            res = self.results[key] = (dw, buf[nw:] if lay["sums"] else None)
        return res


class _PlaneViewsFn(torch.autograd.Function):
    """The D per-plane views of a parameter (RED_Regularization._per_plane_parameters) with the gradients summed here: whatever the
    planes delivered as real tensors (one stack + one sum) plus what the layer's backward deferred to the sink."""

    @staticmethod
    def forward(ctx, p, d_num, sink):
        ctx.sink, ctx.ptr = sink, p.data_ptr()
        return p.unsqueeze(0).expand(d_num, *p.shape).unbind(0)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        real = [g for g in grads if g is not None and not _is_placeholder(g)]
        total = None
        sink = ctx.sink
        if sink is not None:
            if ctx.ptr in sink.layers or ctx.ptr in sink.results:
                total = sink.result(ctx.ptr)[0]
            elif ctx.ptr in sink.bias_of and (sink.bias_of[ctx.ptr] in sink.layers or sink.bias_of[ctx.ptr] in sink.results):
                total = sink.result(sink.bias_of[ctx.ptr])[1]
        if real:
            s_ = real[0] if len(real) == 1 else torch.stack(real).sum(0)
            total = s_ if total is None else total + s_
        return total, None, None


class _Conv3x3WgradFn(torch.autograd.Function):
    """A 3x3 / pad 1 nn.Conv2d (stride 1 or 2) or nn.ConvTranspose2d (stride 2 with output_padding 1, or stride 1) whose WEIGHT and BIAS
    gradients come from smvs_conv3x3_wgrad_strided (csrc/conv_wgrad.hip); the forward and the input gradient stay torch's (MIOpen's
    direct kernels are fine there).  On this image MIOpen computes the weight gradient of these small-channel layers as im2col + layout
    transposes + implicit GEMM + col2im: ~5 launches per call, ~1 300 calls per training step of the 48/32/8 cascade
    (profiles/r04_train_step.txt)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, transposed):
        ctx.save_for_backward(x, weight)
        ctx.meta = (bias is not None, int(stride), bool(transposed))
        ctx.sink = getattr(_TLS, "sink", None)
        ctx.bias = bias if (ctx.sink is not None and bias is not None and not transposed) else None
        arena = getattr(_TLS, "arena", None) if ctx.sink is None else None
        ctx.zeroed = arena.take(weight.numel() + (weight.shape[0] if bias is not None and not transposed else 0)) if arena is not None else None
        if transposed:
            return F.conv_transpose2d(x, weight, bias, stride=stride, padding=1, output_padding=stride - 1)
        return F.conv2d(x, weight, bias, stride=stride, padding=1)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        has_bias, stride, transposed = ctx.meta
        dy = _f32c_fast(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(dy, x, weight, None, [stride, stride], [1, 1], [1, 1], transposed, [stride - 1, stride - 1] if transposed else [0, 0],
                                                     1, [True, False, False])[0]
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            xc = _f32c_fast(x)
            window, grid = (dy, xc) if transposed else (xc, dy)             # the tensor read through the taps / the one on the output grid
            B, Cg, H, W = grid.shape
            Cw = window.shape[1]
            nw = weight.numel()
            sums = has_bias and not transposed                              # a convolution's bias gradient = the grid tensor's sums
            if ctx.sink is not None:                                        # deferred: one launch per layer after the last plane
                ctx.sink.add(weight, ctx.bias, window, None, grid, stride)
                db = (_placeholder(ctx.bias) if sums else dy.sum((0, 2, 3))) if has_bias else None
                return dx, _placeholder(weight), db, None, None
            buf, ctx.zeroed = ctx.zeroed, None                              # (a second backward through the same graph gets fresh memory)
            if buf is None or buf.device != xc.device:
                buf = torch.zeros((nw + (Cg if sums else 0),), dtype=torch.float32, device=xc.device)    # one fill for both gradients
            dw = buf[:nw].view(weight.shape)
            with torch.cuda.device(xc.device):
                _lib.call("smvs_conv3x3_wgrad_strided", _lib.ptr(window), _lib.ptr(grid), _lib.ptr(dw), _lib.ptr(buf[nw:]) if sums else None,
                          B, Cw, Cg, H, W, stride, _lib.current_stream(xc.device))
            if has_bias:
                db = buf[nw:] if sums else dy.sum((0, 2, 3))
        return dx, dw, db, None, None


def _conv3d_wgrad(window, grid, weight_shape, stride, zeroed=None):
    """dw (zero-filled -- `zeroed`: a slice of the forward's arena, else a fresh fill -- then smvs_conv3d_wgrad in its two-stage,
    deterministic form: per-wave partial sums in a scratch buffer from torch's allocator, folded by a second kernel)."""
    dev = window.device
    B, Cg, D, H, W = grid.shape
    Cw = window.shape[1]
    dw = zeroed.view(weight_shape) if zeroed is not None and zeroed.device == dev else torch.zeros(weight_shape, dtype=torch.float32, device=dev)
    nws = _lib.load().smvs_conv3d_wgrad_workspace_floats(B, Cw, Cg, D, H, W)
    ws = torch.empty((max(nws, 1),), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("smvs_conv3d_wgrad", _lib.ptr(window), _lib.ptr(grid), _lib.ptr(dw), _lib.ptr(ws), nws, B, Cw, Cg, D, H, W, stride,
                  _lib.current_stream(dev))
    return dw


class _Conv3dWgradFn(torch.autograd.Function):
    """A 3x3x3 / pad 1 nn.Conv3d (stride 1 or 2) or nn.ConvTranspose3d (stride 2, output_padding 1) without bias -- every layer of
    CostRegNet (reference modules/module.py:324-410, 546-577) -- whose WEIGHT gradient comes from smvs_conv3d_wgrad
    (csrc/conv_wgrad.hip); forward and input gradient stay torch's (MIOpen / composable-kernel implicit GEMMs: 10 ms of a step).
    On this image MIOpen computes these weight gradients with naive_conv_ab_nonpacked_wrw_ncdhw (one thread per weight) and a
    batched-GEMM fallback: 1.98 s of the 1.98 s training step of CascadeMVSNet at the 768x384 tile (profiles/r05_train_step_casmvs.txt)."""

    @staticmethod
    def forward(ctx, x, weight, stride, transposed):
        ctx.save_for_backward(x, weight)
        ctx.meta = (int(stride), bool(transposed))
        if transposed:
            return F.conv_transpose3d(x, weight, None, stride=stride, padding=1, output_padding=stride - 1)
        return F.conv3d(x, weight, None, stride=stride, padding=1)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, transposed = ctx.meta
        dy = _f32c_fast(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(dy, x, weight, None, [stride] * 3, [1] * 3, [1] * 3, transposed,
                                                     [stride - 1] * 3 if transposed else [0] * 3, 1, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            xc = _f32c_fast(x)
            window, grid = (dy, xc) if transposed else (xc, dy)             # the tensor read through the taps / the one on the output grid
            dw = _conv3d_wgrad(window, grid, weight.shape, stride)
        return dx, dw, None, None


def _conv3d_packed(weight, layout, cin, cout):
    """Kernel-layout copy of a 3x3x3 weight for smvs_conv3d_fwd (layouts: include/satmvs.h), packed once per parameter version (the cache
    of _conv_packed: keyed by address, layout and direction; a graph replay bumps the epoch)."""
    from torch.multiprocessing.reductions import StorageWeakRef
    key = (weight.data_ptr(), "3d%d" % layout, cin, weight.device.index)
    hit = _CONV_PACK.get(key)
    if hit is not None and hit[0] == weight._version and hit[1] == _PARAM_EPOCH[0] and not hit[2].expired():
        return hit[3]
    packed = torch.empty((_lib.load().smvs_conv3d_packed_floats(cin, cout),), dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        _lib.call("smvs_conv3d_pack", _lib.ptr(weight), _lib.ptr(packed), cin, cout, layout, _lib.current_stream(weight.device))
    if sum(1 for k in _CONV_PACK if k[3] == key[3]) > 256:          # per device (nn.DataParallel: ~120 entries per replica device);
        for k in [k for k in _CONV_PACK if k[3] == key[3]][:128]:     # the oldest half of THIS device's entries goes (dicts keep insertion order)
            del _CONV_PACK[k]
    _CONV_PACK[key] = (weight._version, _PARAM_EPOCH[0], StorageWeakRef(weight.untyped_storage()), packed)
    return packed


# layer kinds of _Conv3dNativeFn: (forward kernel kind, forward weight layout, input-gradient kernel kind, its weight layout, window stride)
_NATIVE3D_KINDS = {"c1": (0, 0, 0, 2, 1),    # nn.Conv3d stride 1:           correlation / correlation with the transposed, flipped weights
                   "c2": (1, 0, 2, 1, 2),    # nn.Conv3d stride 2:           strided correlation / stride-2 transposed convolution
                   "t2": (2, 1, 1, 0, 2)}    # nn.ConvTranspose3d stride 2:  transposed convolution / strided correlation


def _conv3d_out_dims(kind, dims):
    return tuple(d // 2 for d in dims) if kind == "c2" else tuple(2 * d for d in dims) if kind == "t2" else tuple(dims)


def _conv3d_launchable(kind_id, B, cin, cout, dims):
    """Limits of smvs_conv3d_fwd (one launch grid, 32-bit byte offsets per batch item): callers outside keep torch's operator."""
    do = tuple(d // 2 for d in dims) if kind_id == 1 else tuple(2 * d for d in dims) if kind_id == 2 else tuple(dims)
    vi, vo = dims[0] * dims[1] * dims[2], do[0] * do[1] * do[2]
    rows = dims[0] * dims[1] if kind_id == 2 else do[0] * do[1]
    return (cin * vi * 4 < 2 ** 32 and cout * vo * 4 < 2 ** 32 and (rows + 3) // 4 <= 65535 and B * ((cout + 7) // 8) <= 65535
            and vo // 32 * B < 2 ** 31)


class _Conv3dNativeFn(torch.autograd.Function):
    """A 3x3x3 / pad 1 layer of CostRegNet (nn.Conv3d stride 1 / 2, nn.ConvTranspose3d stride 2 with output_padding 1, no bias) on the
    kernels of the inference regulariser without the folded BatchNorm (smvs_conv3d_fwd: direct or MFMA by channel count): forward,
    input gradient (the adjoint layer on the same kernels) and weight gradient (smvs_conv3d_wgrad).  Replaces, per layer and step, a
    composable-kernel implicit GEMM (forward), an im2col / GEMM / col2im chain or a grouped backward-data GEMM with its layout
    transposes (input gradient) and MIOpen's naive weight-gradient kernel.  kind: _NATIVE3D_KINDS."""

    @staticmethod
    def forward(ctx, x, weight, kind):
        fk, flay, _, _, _ = _NATIVE3D_KINDS[kind]
        x = _f32c_fast(x)
        B, Cin = x.shape[0], x.shape[1]
        dims = tuple(x.shape[2:])
        Cout = weight.shape[1] if kind == "t2" else weight.shape[0]
        out = torch.empty((B, Cout) + _conv3d_out_dims(kind, dims), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("smvs_conv3d_fwd", fk, _lib.ptr(x), _lib.ptr(_conv3d_packed(weight, flay, Cin, Cout)), None, _lib.ptr(out), B, Cin, Cout,
                      dims[0], dims[1], dims[2], 0, _lib.current_stream(x.device))
        ctx.save_for_backward(x, weight)
        ctx.kind = kind
        arena = getattr(_TLS, "arena", None)
        ctx.zeroed = arena.take(weight.numel()) if arena is not None else None      # the weight gradient's memory, zeroed with the forward's one fill
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        kind = ctx.kind
        _, _, bk, blay, stride = _NATIVE3D_KINDS[kind]
        dy = _f32c_fast(dy)
        B, Cin = x.shape[0], x.shape[1]
        Cout = dy.shape[1]
        dev = x.device
        dx = dw = None
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _lib.call("smvs_conv3d_fwd", bk, _lib.ptr(dy), _lib.ptr(_conv3d_packed(weight, blay, Cout, Cin)), None, _lib.ptr(dx), B, Cout, Cin,
                          dy.shape[2], dy.shape[3], dy.shape[4], 0, _lib.current_stream(dev))
            if ctx.needs_input_grad[1]:
                window, grid = (dy, x) if kind == "t2" else (x, dy)         # the tensor read through the taps / the one on the output grid
                buf, ctx.zeroed = ctx.zeroed, None                          # (a second backward through the same graph gets fresh memory)
                dw = _conv3d_wgrad(window, grid, weight.shape, stride, buf)
        return dx, dw, None


def _conv3d(conv, x):
    """conv(x) for the 3-D regulariser's layers (nn.Conv3d stride 1 / 2, nn.ConvTranspose3d stride 2; 3x3x3, pad 1, no bias) where a
    gradient is wanted on the GPU: fully native (forward, input gradient, weight gradient); SMVS_TRAIN_COMPOSITE_MASK bit 128 keeps
    torch's forward and input gradient with the native weight gradient, bit 64 torch's operator alone."""
    transposed = isinstance(conv, nn.ConvTranspose3d)
    s = conv.stride[0]
    if (x.is_cuda and x.dim() == 5 and x.dtype is torch.float32 and torch.is_grad_enabled() and conv.weight.requires_grad
            and conv.weight.dtype is torch.float32 and conv.weight.is_contiguous() and conv.bias is None
            and conv.kernel_size == (3, 3, 3) and conv.stride in ((1, 1, 1), (2, 2, 2)) and conv.padding == (1, 1, 1)
            and conv.dilation == (1, 1, 1) and conv.groups == 1
            and ((transposed and s == 2 and conv.output_padding == (1, 1, 1)) or
                 (not transposed and all(d % s == 0 for d in x.shape[2:])))
            and not (SW.train_composite_mask & 64)
            and x.shape[2] * x.shape[3] * x.shape[4] * (s ** 3 if transposed else 1) * 4 * 8 < 2 ** 31):
        kind = "t2" if transposed else "c%d" % s
        fk, _, bk, _, _ = _NATIVE3D_KINDS[kind]
        B, cin = x.shape[0], x.shape[1]
        cout = conv.weight.shape[1] if transposed else conv.weight.shape[0]
        dims = tuple(x.shape[2:])
        if (not (SW.train_composite_mask & 128) and _conv3d_launchable(fk, B, cin, cout, dims)
                and _conv3d_launchable(bk, B, cout, cin, _conv3d_out_dims(kind, dims))):
            return _Conv3dNativeFn.apply(x, conv.weight, kind)
        return _Conv3dWgradFn.apply(x, conv.weight, s, transposed)
    return conv(x)


class _BatchNormReluFn(torch.autograd.Function):
    """[relu](BatchNorm3d(x)) / [relu](BatchNorm2d(x)) in training form (batch statistics, running statistics updated) on
    smvs_batchnorm_train_fwd / _bwd (csrc/batchnorm.hip): the conv -> bn -> relu blocks of CostRegNet (reference modules/module.py:324-410)
    and of FeatureNet (:78-118).  Saves the block's input
    and (mean, rstd) only: the ReLU mask is recomputed from x in the backward.  `bn` (the nn.BatchNorm3d) rides along as a non-tensor
    argument: its running statistics are updated in place like F.batch_norm(training=True) does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, relu):
        x = _f32c_fast(x)
        B, C = x.shape[0], x.shape[1]
        N = x.numel() // (B * C)
        y = torch.empty_like(x)
        saved = torch.empty((C, 2), dtype=torch.float32, device=x.device)
        arena = getattr(_TLS, "arena", None)
        zf = arena.take(4 * C) if arena is not None and arena.buf.device == x.device else None      # 2 C doubles, zeroed by the forward's one fill
        zb = arena.take(4 * C) if zf is not None else None                                          # ... and the backward's
        ws = zf.view(torch.float64) if zf is not None else torch.empty((2 * C,), dtype=torch.float64, device=x.device)
        track = bn.track_running_stats and bn.running_mean is not None
        with torch.cuda.device(x.device):
            _lib.call("smvs_batchnorm_train_fwd", _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(bn.running_mean) if track else None,
                      _lib.ptr(bn.running_var) if track else None,
                      _lib.ptr(bn.num_batches_tracked) if track and bn.num_batches_tracked is not None and bn.num_batches_tracked.is_cuda else None,
                      float(bn.momentum), float(bn.eps), (1 if relu else 0) | (2 if zf is not None else 0),
                      _lib.ptr(y), _lib.ptr(saved), _lib.ptr(ws), B, C, N, _lib.current_stream(x.device))
        ctx.save_for_backward(x, gamma, beta, saved)
        ctx.relu = bool(relu)
        ctx.zeroed = zb
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, gamma, beta, saved = ctx.saved_tensors
        dy = _f32c_fast(dy)
        B, C = x.shape[0], x.shape[1]
        N = x.numel() // (B * C)
        dx = torch.empty_like(x)
        dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)
        zb, ctx.zeroed = ctx.zeroed, None                                   # (a second backward through the same graph clears its own)
        ws = zb.view(torch.float64) if zb is not None else torch.empty((2 * C,), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("smvs_batchnorm_train_bwd", _lib.ptr(dy), _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(saved),
                      (1 if ctx.relu else 0) | (2 if zb is not None else 0),
                      _lib.ptr(dx), _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), _lib.ptr(ws), B, C, N, _lib.current_stream(x.device))
        return dx, dgb[0], dgb[1], None, None


def _bn3d_relu(bn, x, relu):
    """[relu](bn(x)) for a training nn.BatchNorm3d / nn.BatchNorm2d on the native kernels, or None where they do not apply (evaluation
    mode, CPU, no affine parameters, cumulative-average momentum, SMVS_TRAIN_COMPOSITE_MASK bit 256): the caller keeps torch's operators."""
    if not (bn.training and x.is_cuda and x.dim() in (4, 5) and x.dtype is torch.float32 and torch.is_grad_enabled() and bn.affine
            and bn.momentum is not None and bn.weight.dtype is torch.float32 and bn.bias.dtype is torch.float32
            and bn.weight.is_contiguous() and bn.bias.is_contiguous() and x.shape[1] == bn.num_features      # (a mismatch: torch's operator raises its own error)
            and x.shape[1] <= 65535 and not (SW.train_composite_mask & 256)):
        return None
    return _BatchNormReluFn.apply(x, bn.weight, bn.bias, bn, relu)          # (num_batches_tracked += 1 happens in the forward kernel)


_CONV_PACK = {}         # (weight address, layout, cin) -> (version, epoch, storage weak reference, packed tensor): one entry per layer and direction


def _conv_packed(weight, layout, cin, cout):
    """Kernel-layout copy of a 3x3 weight for smvs_conv3x3_fwd (layouts: include/satmvs.h), packed once per parameter version --
    every plane of a training step uses a view of the same parameter."""
    from torch.multiprocessing.reductions import StorageWeakRef
    key = (weight.data_ptr(), layout, cin, weight.device.index)
    hit = _CONV_PACK.get(key)
    if hit is not None and hit[0] == weight._version and hit[1] == _PARAM_EPOCH[0] and not hit[2].expired():
        return hit[3]
    packed = torch.empty((_lib.load().smvs_conv3x3_packed_floats(cin, cout),), dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        _lib.call("smvs_conv3x3_pack", _lib.ptr(weight), _lib.ptr(packed), cin, cout, layout, _lib.current_stream(weight.device))
    if sum(1 for k in _CONV_PACK if k[3] == key[3]) > 256:          # per device (nn.DataParallel: ~120 entries per replica device);
        for k in [k for k in _CONV_PACK if k[3] == key[3]][:128]:     # the oldest half of THIS device's entries goes (dicts keep insertion order)
            del _CONV_PACK[k]
    _CONV_PACK[key] = (weight._version, _PARAM_EPOCH[0], StorageWeakRef(weight.untyped_storage()), packed)
    return packed


# layer kinds of _Conv3x3NativeFn: (forward kernel kind, forward weight layout, input-gradient kernel kind, its weight layout, window stride)
_NATIVE_KINDS = {"c1": (0, 0, 0, 2, 1),      # nn.Conv2d stride 1:           correlation / correlation with the transposed, flipped weights
                 "c2": (1, 0, 2, 1, 2),      # nn.Conv2d stride 2:           strided correlation / stride-2 transposed convolution
                 "t2": (2, 1, 1, 0, 2),      # nn.ConvTranspose2d stride 2:  transposed convolution / strided correlation
                 "t1": (0, 2, 0, 0, 1)}      # nn.ConvTranspose2d stride 1:  correlation with flipped taps / correlation


class _Conv3x3NativeFn(torch.autograd.Function):
    """[relu](layer(cat(xa, xb))) for the regulariser's 3x3 / pad 1 layers on the kernels of the RED plane loop (smvs_conv3x3_fwd: direct
    or MFMA by channel count): forward, input gradient (the adjoint layer on the same kernels) and weight / bias gradient
    (smvs_conv3x3_wgrad_list through _WgradSink, or smvs_conv3x3_wgrad_cat / _strided per call), without the concatenated tensor.
    xb may be None; kind: _NATIVE_KINDS."""

    @staticmethod
    def forward(ctx, xa, xb, weight, bias, kind, relu):
        fk, flay, _, _, _ = _NATIVE_KINDS[kind]
        xa = _f32c_fast(xa)
        xb = _f32c_fast(xb) if xb is not None else None
        B, CA, H, W = xa.shape
        CB = xb.shape[1] if xb is not None else 0
        transposed = kind[0] == "t"
        Cout = weight.shape[1] if transposed else weight.shape[0]
        Ho, Wo = (H // 2, W // 2) if kind == "c2" else (2 * H, 2 * W) if kind == "t2" else (H, W)
        out = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=xa.device)
        packed = _conv_packed(weight, flay, CA + CB, Cout)
        fused_bias = bias if kind != "t2" else None              # (the stride-2 transposed layers of the regulariser have no bias)
        with torch.cuda.device(xa.device):
            _lib.call("smvs_conv3x3_fwd", fk, _lib.ptr(xa), CA, _lib.ptr(xb) if xb is not None else None, CB, _lib.ptr(packed),
                      _lib.ptr(fused_bias) if fused_bias is not None else None, None, _lib.ptr(out), B, Cout, H, W, 1 if relu else 0,
                      _lib.current_stream(xa.device))
        ctx.save_for_backward(xa, xb, weight, out if relu else None)
        ctx.kind, ctx.relu, ctx.has_bias = kind, bool(relu), bias is not None
        ctx.sink = getattr(_TLS, "sink", None)
        sums = bias is not None and not transposed               # a convolution's bias gradient = the sums of its output gradient
        ctx.bias = bias if (ctx.sink is not None and sums) else None
        arena = getattr(_TLS, "arena", None) if ctx.sink is None else None
        ctx.zeroed = arena.take(weight.numel() + (Cout if sums else 0)) if arena is not None else None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        xa, xb, weight, out = ctx.saved_tensors
        kind = ctx.kind
        _, _, bk, blay, stride = _NATIVE_KINDS[kind]
        dy = _f32c_fast(dy)
        if ctx.relu:
            dy = torch.ops.aten.threshold_backward(dy, out, 0.0)
        B, CA, H, W = xa.shape
        CB = xb.shape[1] if xb is not None else 0
        transposed = kind[0] == "t"
        Cout = dy.shape[1]
        dev = xa.device
        dxa = dxb = dw = db = None
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[0] or (xb is not None and ctx.needs_input_grad[1]):
                dx = torch.empty((B, CA + CB, H, W), dtype=torch.float32, device=dev)
                _lib.call("smvs_conv3x3_fwd", bk, _lib.ptr(dy), Cout, None, 0, _lib.ptr(_conv_packed(weight, blay, Cout, CA + CB)), None, None,
                          _lib.ptr(dx), B, CA + CB, dy.shape[2], dy.shape[3], 0, _lib.current_stream(dev))
                dxa = dx[:, :CA] if ctx.needs_input_grad[0] else None
                dxb = dx[:, CA:] if xb is not None and ctx.needs_input_grad[1] else None
            if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
                sums = ctx.has_bias and not transposed
                # the tensor read through the taps / the one on the output grid (they swap for the transposed layers)
                window, win2, grid = (dy, None, xa) if transposed else (xa, xb, dy)
                if ctx.sink is not None:                         # deferred: one launch per layer after the last plane
                    ctx.sink.add(weight, ctx.bias, window, win2, grid, stride)
                    dw = _placeholder(weight)
                    db = (_placeholder(ctx.bias) if sums else dy.sum((0, 2, 3))) if ctx.has_bias else None
                else:
                    nw = weight.numel()
                    buf, ctx.zeroed = ctx.zeroed, None           # (a second backward through the same graph gets fresh memory)
                    if buf is None or buf.device != dev:
                        buf = torch.zeros((nw + (Cout if sums else 0),), dtype=torch.float32, device=dev)
                    dw = buf[:nw].view(weight.shape)
                    _lib.call("smvs_conv3x3_wgrad_list", _lib.ptr_array([window]), _lib.ptr_array([win2]) if win2 is not None else None,
                              _lib.ptr_array([grid]), 1, _lib.ptr(dw), _lib.ptr(buf[nw:]) if sums else None, B, window.shape[1],
                              win2.shape[1] if win2 is not None else 0, grid.shape[1], grid.shape[2], grid.shape[3], stride,
                              _lib.current_stream(dev))
                    if ctx.has_bias:
                        db = buf[nw:] if sums else dy.sum((0, 2, 3))
        return dxa, dxb, dw, db, None, None


def _native_kind(conv):
    """_NATIVE_KINDS key of a 3x3 / pad 1 layer, or None."""
    if conv.kernel_size != (3, 3) or conv.padding != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1 or conv.stride not in ((1, 1), (2, 2)):
        return None
    s = conv.stride[0]
    if isinstance(conv, nn.ConvTranspose2d):
        return ("t%d" % s) if conv.output_padding == (s - 1, s - 1) and (s == 1 or conv.bias is None) else None
    return ("c%d" % s) if isinstance(conv, nn.Conv2d) else None


def _conv3x3_cat(conv, xa, xb=None, relu=False):
    """[relu](conv(cat(xa, xb))) for the regulariser's 3x3 layers: fully native under autograd on the GPU (SMVSSW.train_composite_mask
    bit 16 keeps torch's convolution with the native weight gradient, bit 8 torch's convolution alone)."""
    x_all = (xa,) if xb is None else (xa, xb)
    kind = _native_kind(conv)
    h, w = xa.shape[2], xa.shape[3]
    cin = sum(t.shape[1] for t in x_all)
    cout = conv.weight.shape[1] if isinstance(conv, nn.ConvTranspose2d) else conv.weight.shape[0]
    if (kind is not None and xa.is_cuda and all(t.dtype is torch.float32 for t in x_all) and torch.is_grad_enabled() and conv.weight.requires_grad
            and conv.weight.dtype is torch.float32 and conv.weight.is_contiguous()
            and not (SW.train_composite_mask & 24) and (xb is None or (xa.shape[1] % 2 == 0 and kind == "c1"))
            and (kind != "c2" or (h % 2 == 0 and w % 2 == 0))
            and max(cin, cout) * h * w * 4 * (4 if kind == "t2" else 1) < 2 ** 31
            and xa.shape[0] * ((max(cin, cout) + 7) // 8) <= 65535
            and (conv.bias is None or conv.bias.data_ptr() % 16 == 0)):
        return _Conv3x3NativeFn.apply(xa, xb, conv.weight, conv.bias, kind, relu)
    y = _conv3x3(conv, xa if xb is None else torch.cat((xa, xb), dim=1))
    return F.relu(y, inplace=True) if relu else y


def _conv3x3(conv, x):
    """conv(x) for the regulariser's 3x3 layers (nn.Conv2d stride 1 / 2, nn.ConvTranspose2d stride 2 / 1, pad 1): with the native weight
    gradient where a gradient is wanted on the GPU."""
    transposed = isinstance(conv, nn.ConvTranspose2d)
    s = conv.stride[0]
    if (x.is_cuda and x.dtype is torch.float32 and torch.is_grad_enabled() and conv.weight.requires_grad and conv.weight.dtype is torch.float32
            and conv.kernel_size == (3, 3) and conv.stride in ((1, 1), (2, 2)) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and (not transposed or conv.output_padding == (s - 1, s - 1)) and (transposed or (x.shape[2] % s == 0 and x.shape[3] % s == 0))
            and not (SW.train_composite_mask & 8) and x.shape[2] * x.shape[3] * 8 * 4 * (s * s if transposed else 1) < 2 ** 31):
        return _Conv3x3WgradFn.apply(x, conv.weight, conv.bias, s, transposed)
    return conv(x)


def _wgrad_now_or_later(sink, weight, bias, window, win2, grid, stride):
    """(dw, db) of a 3x3 convolution (bias gradient = the grid tensor's sums): placeholders when a _WgradSink collects the planes,
    else one smvs_conv3x3_wgrad_list launch now."""
    if sink is not None:
        sink.add(weight, bias, window, win2, grid, stride)
        return _placeholder(weight), (_placeholder(bias) if bias is not None else None)
    dev = window.device
    nw = weight.numel()
    Cg = grid.shape[1]
    buf = torch.zeros((nw + (Cg if bias is not None else 0),), dtype=torch.float32, device=dev)
    dw = buf[:nw].view(weight.shape)
    _lib.call("smvs_conv3x3_wgrad_list", _lib.ptr_array([window]), _lib.ptr_array([win2]) if win2 is not None else None, _lib.ptr_array([grid]), 1,
              _lib.ptr(dw), _lib.ptr(buf[nw:]) if bias is not None else None, window.shape[0], window.shape[1],
              win2.shape[1] if win2 is not None else 0, Cg, grid.shape[2], grid.shape[3], stride, _lib.current_stream(dev))
    return dw, (buf[nw:] if bias is not None else None)


class _ConvGRUCellFn(torch.autograd.Function):
    """A whole ConvGRUCell2 step (reference: module.py:24-57) as ONE autograd node, batch 1: gate convolution over (x, h), both gate norms
    + sigmoids (+ r*h), candidate convolution over (x, r*h), output norm + tanh (+ blend) -- four native calls, and a
    hand-chained backward in which the three gradient contributions to h and the two to x meet inside the kernels instead of in
    autograd's accumulation adds: the blend's state gradient rides into smvs_gru_mul_cat_bwd_acc, which leaves [dx | dh] in place of the
    candidate convolution's input gradient, and that buffer seeds the gate convolution's input gradient (smvs_conv3x3_fwd's `init`).
    Per cell and plane 3 adds + 1 cat (the gradient of torch.split) fewer: ~1 400 launches per training step of the 48/32/8 cascade."""

    @staticmethod
    def forward(ctx, x, h, gw, gb, rw, rb, uw, ub, ow, ob, nw_, nb_, eps):
        dev = _lib.require_device(x, h, gw)
        x, h = _f32c_fast(x), _f32c_fast(h)
        B, Cx, H, W = x.shape
        C = h.shape[1]
        HW = H * W
        st = _lib.current_stream(dev)
        rw, rb, uw, ub, nw_, nb_ = [_f32c_fast(t.detach()) for t in (rw, rb, uw, ub, nw_, nb_)]
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        gates, ru, rh, craw, cand, out = e(B, 2 * C, H, W), e(B, 2 * C, H, W), e(B, C, H, W), e(B, C, H, W), e(B, C, H, W), e(B, C, H, W)
        stats_g, stats_o = e(2 * B, 2), e(B, 2)
        nblk = (C * HW + 4095) // 4096
        with torch.cuda.device(dev):
            _lib.call("smvs_conv3x3_fwd", 0, _lib.ptr(x), Cx, _lib.ptr(h), C, _lib.ptr(_conv_packed(gw, 0, Cx + C, 2 * C)), _lib.ptr(gb), None,
                      _lib.ptr(gates), B, 2 * C, H, W, 0, st)
            # (round 6: r * h leaves the gate norms' apply pass and the candidate convolution reads (x, r*h) as two tensors -- no cat launch;
            # the blend leaves the output norm's apply pass -- no blend launch)
            _lib.call("smvs_groupnorm1_pair_fwd_mul", _lib.ptr(gates), _lib.ptr(rw), _lib.ptr(rb), _lib.ptr(uw), _lib.ptr(ub), float(eps), 1, _lib.ptr(ru),
                      _lib.ptr(stats_g), _lib.ptr(_gn_scratch(dev, 4 * B * nblk)), _lib.ptr(h), _lib.ptr(rh), B, C, HW, st)
            _lib.call("smvs_conv3x3_fwd", 0, _lib.ptr(x), Cx, _lib.ptr(rh), C, _lib.ptr(_conv_packed(ow, 0, Cx + C, C)), _lib.ptr(ob), None,
                      _lib.ptr(craw), B, C, H, W, 0, st)
            _lib.call("smvs_groupnorm1_fwd_blend", _lib.ptr(craw), C * HW, _lib.ptr(nw_), _lib.ptr(nb_), float(eps), 2, _lib.ptr(cand), _lib.ptr(stats_o),
                      _lib.ptr(_gn_scratch(dev, 2 * B * nblk)), _lib.ptr(ru[:, C:]), 2 * C * HW, _lib.ptr(h), _lib.ptr(out), B, C, HW, st)
        ctx.save_for_backward(x, h, gw, gb, ow, ob, rw, uw, nw_, gates, ru, rh, craw, cand, stats_g, stats_o)
        ctx.sink = getattr(_TLS, "sink", None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dnew):
        x, h, gw, gb, ow, ob, rw, uw, nw_, gates, ru, rh, craw, cand, stats_g, stats_o = ctx.saved_tensors
        dev = x.device
        B, Cx, H, W = x.shape
        C = h.shape[1]
        HW = H * W
        st = _lib.current_stream(dev)
        dnew = _f32c_fast(dnew)
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        dru, dh_b, dcand, dcraw, dxc, dgates = e(B, 2 * C, H, W), e(B, C, H, W), e(B, C, H, W), e(B, C, H, W), e(B, Cx + C, H, W), e(B, 2 * C, H, W)
        g4, dn = e(4, C), e(2, C)
        r, u = ru[:, :C], ru[:, C:]
        nseg = (HW + 4095) // 4096
        with torch.cuda.device(dev):
            _lib.call("smvs_gru_blend_bwd", _lib.ptr(dnew), _lib.ptr(u), _lib.ptr(h), _lib.ptr(cand), _lib.ptr(dru[:, C:]), _lib.ptr(dh_b), _lib.ptr(dcand),
                      h.numel(), st)
            _lib.call("smvs_groupnorm1_bwd", _lib.ptr(dcand), _lib.ptr(craw), C * HW, _lib.ptr(cand), _lib.ptr(nw_), _lib.ptr(stats_o), 2, _lib.ptr(dcraw),
                      C * HW, _lib.ptr(dn[0]), _lib.ptr(dn[1]), _lib.ptr(_gn_scratch(dev, 2 * B * C * nseg)), B, C, HW, st)
            _lib.call("smvs_conv3x3_fwd", 0, _lib.ptr(dcraw), C, None, 0, _lib.ptr(_conv_packed(ow, 2, C, Cx + C)), None, None, _lib.ptr(dxc),
                      B, Cx + C, H, W, 0, st)
            dow, dob = _wgrad_now_or_later(ctx.sink, ow, ob, x, rh, dcraw, 1)
            _lib.call("smvs_gru_mul_cat_bwd_acc", _lib.ptr(dxc), _lib.ptr(r), _lib.ptr(h), _lib.ptr(dh_b), _lib.ptr(dru), B, Cx, C, HW, st)
            _lib.call("smvs_groupnorm1_pair_bwd", _lib.ptr(dru), _lib.ptr(gates), _lib.ptr(ru), _lib.ptr(rw), _lib.ptr(uw), _lib.ptr(stats_g), 1,
                      _lib.ptr(dgates), _lib.ptr(g4[0]), _lib.ptr(g4[1]), _lib.ptr(g4[2]), _lib.ptr(g4[3]), _lib.ptr(_gn_scratch(dev, 4 * B * C * nseg)),
                      B, C, HW, st)
            _lib.call("smvs_conv3x3_fwd", 0, _lib.ptr(dgates), 2 * C, None, 0, _lib.ptr(_conv_packed(gw, 2, 2 * C, Cx + C)), None, _lib.ptr(dxc),
                      _lib.ptr(dxc), B, Cx + C, H, W, 0, st)
            dgw, dgb = _wgrad_now_or_later(ctx.sink, gw, gb, x, h, dgates, 1)
        return dxc[:, :Cx], dxc[:, Cx:], dgw, dgb, g4[0], g4[1], g4[2], g4[3], dow, dob, dn[0], dn[1], None
