"""ctypes binding of libsatmvs_hip.so (the C ABI declared in include/satmvs.h).

There is no CPU fallback: if the library is missing, or a tensor is not on a HIP device, the
call raises.  PyTorch is used only for device memory and streams; every pointer handed to the
library is `tensor.data_ptr()` and the stream is torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMVS_LIB_PATH: load another build of the same library (A/B timing of kernel variants in one process-level run)
LIB_PATH = os.environ.get("SMVS_LIB_PATH") or os.path.join(_HERE, "lib", "libsatmvs_hip.so")

_vp, _i, _sz, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_float

# name -> argtypes; mirrors include/satmvs.h one to one
_SIGNATURES = {
    "smvs_rpc_costvol_fwd": [_vp, _vp, _i, _vp, _vp, _i, _vp] + [_i] * 9 + [_vp],
    "smvs_homo_costvol_fwd": [_vp, _vp, _i, _vp, _vp, _i, _vp] + [_i] * 9 + [_vp],
    "smvs_rpc_plane_coef": [_vp, _vp, _i, _vp] + [_i] * 7 + [_vp],
    "smvs_rpc_costvol_fwd_pc": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp] + [_i] * 9 + [_vp],
    "smvs_rpc_warp_fwd": [_vp, _vp, _vp, _vp, _i, _vp] + [_i] * 5 + [_vp],
    "smvs_rpc_warp_bwd": [_vp, _vp, _vp, _vp, _i, _vp] + [_i] * 5 + [_vp],
    "smvs_homo_warp_fwd": [_vp, _vp, _vp, _i, _vp] + [_i] * 5 + [_vp],
    "smvs_homo_warp_bwd": [_vp, _vp, _vp, _i, _vp] + [_i] * 5 + [_vp],
    "smvs_homo_compose": [_vp, _vp, _vp, _i, _vp],
    "smvs_costvol_bwd": [_i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp] + [_i] * 5 + [_vp],
    "smvs_rpc_project": [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp],
    "smvs_rpc_geo_consistency": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "smvs_pinhole_geo_consistency": [_vp, _vp, _vp, _i, _i, _i, _i, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "smvs_softmax_regress_fwd": [_vp, _vp, _i, _vp, _vp] + [_i] * 4 + [_vp],
    "smvs_window_regress_fwd": [_vp, _vp, _i, _vp, _vp, _vp, _f] + [_i] * 4 + [_vp],
    "smvs_height_hypotheses": [_vp, _vp, _i, _i, _i, _vp],
    "smvs_gru_mul_cat_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "smvs_gru_mul_cat_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "smvs_gru_blend_fwd": [_vp, _vp, _vp, _vp, C.c_longlong, _vp],
    "smvs_gru_blend_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_longlong, _vp],
    "smvs_groupnorm1_pair_fwd": [_vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "smvs_groupnorm1_fwd_blend": [_vp, C.c_longlong, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, C.c_longlong, _vp, _vp, _i, _i, _i, _vp],
    "smvs_groupnorm1_pair_fwd_mul": [_vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "smvs_groupnorm1_pair_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "smvs_groupnorm1_fwd": [_vp, C.c_longlong, _vp, _vp, _f, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "smvs_groupnorm1_bwd": [_vp, _vp, C.c_longlong, _vp, _vp, _vp, _i, _vp, C.c_longlong, _vp, _vp, _vp, _i, _i, _i, _vp],
    "smvs_rpc_costvol_fwd_gen": [_vp, _vp, _i, _vp, _vp, _vp] + [_i] * 9 + [_vp],
    "smvs_homo_costvol_fwd_gen": [_vp, _vp, _i, _vp, _vp, _vp] + [_i] * 9 + [_vp],
    "smvs_softmax_regress_fwd_gen": [_vp, _vp, _vp, _vp] + [_i] * 4 + [_vp],
    "smvs_window_regress_fwd_gen": [_vp, _vp, _vp, _vp, _vp, _f] + [_i] * 4 + [_vp],
    "smvs_red_pred_planes_gen": [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz] + [_i] * 7 + [_vp],
    "smvs_red_volume_planes_gen": [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz] + [_i] * 7 + [_vp],
    "smvs_stream_regress_step": [_vp, _vp, _i, _vp, _vp, _vp] + [_i] * 5 + [_vp],
    "smvs_stream_regress_final": [_vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "smvs_regress_fold": [_vp, _vp, _i, _sz, _sz, _sz, _vp],
    "smvs_conv3x3_wgrad": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "smvs_conv3x3_wgrad_strided": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "smvs_conv3x3_wgrad_cat": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "smvs_conv3x3_wgrad_list": [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "smvs_conv3d_wgrad": [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp],
    "smvs_conv3d_pack": [_vp, _vp, _i, _i, _i, _vp],
    "smvs_batchnorm_train_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _i, _i, C.c_longlong, _vp],
    "smvs_batchnorm_train_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, C.c_longlong, _vp],
    "smvs_conv3d_fwd": [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "smvs_conv3x3_pack": [_vp, _vp, _i, _i, _i, _vp],
    "smvs_conv3x3_fwd": [_i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "smvs_gru_mul_cat_bwd_acc": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "smvs_red_pack_weights": [_vp, _i, _vp, _vp],
    "smvs_red_step_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz] + [_i] * 4 + [_vp],
    "smvs_red_pred_planes": [_i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz] + [_i] * 7 + [_vp],
    "smvs_red_volume_planes": [_i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz] + [_i] * 7 + [_vp],
    "smvs_costreg_pack_weights": [_vp, _i, _vp, _vp],
    "smvs_costreg_fwd": [_vp, _vp, _vp, _vp, _sz] + [_i] * 5 + [_vp],
    "smvs_featnet_pack_weights": [_vp, _i, _i, _vp, _vp],
    "smvs_featnet_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _sz] + [_i] * 5 + [_vp],
}
_SIZE_FUNCS = {"smvs_rpc_plane_coef_bytes": [_i] * 3, "smvs_red_packed_floats": [_i], "smvs_red_workspace_bytes": [_i] * 4,
               "smvs_red_pred_workspace_bytes": [_i] * 4, "smvs_costreg_packed_floats": [_i],
               "smvs_costreg_workspace_bytes": [_i] * 5, "smvs_featnet_packed_floats": [_i] * 2,
               "smvs_featnet_workspace_bytes": [_i] * 5, "smvs_conv3x3_packed_floats": [_i] * 2,
               "smvs_conv3d_packed_floats": [_i] * 2, "smvs_conv3d_wgrad_workspace_floats": [_i] * 6}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + list(_SIZE_FUNCS) + ["smvs_version", "smvs_last_error", "smvs_red_set_streams", "smvs_shutdown",
                                                                    "smvs_set_arith", "smvs_get_arith"])
ARITH_MODES = {"exact": 0, "fused": 1}      # SMVS_ARITH_EXACT / SMVS_ARITH_FUSED of include/satmvs.h

_lib = None


class SatMVSNativeError(RuntimeError):
    pass


def load():
    """dlopen the library (once).  Raises ImportError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "satmvs_amd: %s not found. Build it with `python -m satmvs_amd.build` (hipcc, gfx950); "
            "there is no CPU fallback for the product path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name, argtypes in _SIZE_FUNCS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_size_t
    lib.smvs_version.restype = C.c_char_p
    lib.smvs_last_error.restype = C.c_char_p
    if hasattr(lib, "smvs_red_set_streams"):               # (older A/B builds loaded through SMVS_LIB_PATH predate these two)
        lib.smvs_red_set_streams.argtypes = [_i]
        lib.smvs_red_set_streams.restype = C.c_int
        lib.smvs_shutdown.argtypes = []
        lib.smvs_shutdown.restype = C.c_int
    if hasattr(lib, "smvs_set_arith"):
        lib.smvs_set_arith.argtypes = [_i]
        lib.smvs_set_arith.restype = C.c_int
        lib.smvs_get_arith.argtypes = []
        lib.smvs_get_arith.restype = C.c_int
        mode = os.environ.get("SMVS_ARITH")                 # the library itself never reads the environment
        if mode:
            if mode not in ARITH_MODES:
                raise ValueError("SMVS_ARITH must be one of %s, got %r" % (sorted(ARITH_MODES), mode))
            lib.smvs_set_arith(ARITH_MODES[mode])
    _lib = lib
    return lib


def set_arith(mode):
    """Arithmetic of the variance build: "exact" = the reference's float32 rounding sequence (bit-identical to the oracle),
    "fused" = the library default (contract tolerance, include/satmvs.h).  Process-wide; returns the previous mode's name."""
    if mode not in ARITH_MODES:
        raise ValueError("arith mode must be one of %s, got %r" % (sorted(ARITH_MODES), mode))
    prev = load().smvs_set_arith(ARITH_MODES[mode])
    return "exact" if prev == 0 else "fused"


def get_arith():
    return "exact" if load().smvs_get_arith() == 0 else "fused"


# ---- arithmetic of ONE call (include/satmvs.h: SMVS_CALL_ARITH_*) --------------------------------------------------
# set_arith() above moves the process default.  A model (or a test) that wants its own arithmetic does not touch it: inside
# `with arith_scope("exact"):` every cost-volume call of THIS thread carries the mode in its arguments (the bits OR-ed into
# depth_is_4d / smvs_height_gen.arith), so two models -- or nn.DataParallel replicas on their threads -- in one process
# can run different arithmetics side by side.
CALL_ARITH_BITS = {"exact": 0x100, "fused": 0x200}
_scope = threading.local()


class arith_scope:
    """Context manager: cost-volume calls made by this thread inside the block use `mode` ("exact" / "fused"; None = no-op)."""

    def __init__(self, mode):
        if mode is not None and mode not in ARITH_MODES:
            raise ValueError("arith mode must be one of %s, got %r" % (sorted(ARITH_MODES), mode))
        self.mode = mode

    def __enter__(self):
        self.prev = getattr(_scope, "mode", None)
        if self.mode is not None:
            _scope.mode = self.mode
        return self

    def __exit__(self, *exc):
        _scope.mode = self.prev
        return False


def scoped_arith():
    """The mode of the innermost arith_scope of this thread, or None."""
    return getattr(_scope, "mode", None)


def pipeline_arith_scope(own=None):
    """arith_scope of a plane pipeline / cascade: `own` (a model's arith= argument) if given, else the enclosing arith_scope,
    else "exact".  The pipelines do NOT follow the process default of the stand-alone builds (set_arith / SMVS_ARITH): behind a
    peaky softmax the fused volume's 1e-5 can move a regressed height by more than 1e-3 m, and the build is a few per cent of a
    pipeline's time -- the C entry points (smvs_red_pred_planes, smvs_red_volume_planes) default the same way."""
    return arith_scope(own or scoped_arith() or "exact")


def call_arith_bits():
    """0, or the SMVS_CALL_ARITH_* bit of the innermost arith_scope of this thread."""
    mode = getattr(_scope, "mode", None)
    return CALL_ARITH_BITS[mode] if mode else 0


def call_arith():
    """Name of the arithmetic the next cost-volume call of this thread will run in."""
    return getattr(_scope, "mode", None) or get_arith()


def version():
    return load().smvs_version().decode()


_TRACE_CALLS = os.environ.get("SMVS_TRACE_CALLS", "0") == "1"
_SYNC_CALLS = os.environ.get("SMVS_SYNC_CALLS", "0") == "1"


def call(name, *args):
    """Invoke an entry point; non-zero return -> SatMVSNativeError(smvs_last_error())."""
    lib = load()
    if _TRACE_CALLS:                                         # SMVS_TRACE_CALLS=1: every native call with its pointer arguments, to stderr
        import sys
        sys.stderr.write("smvs-call %s %s\n" % (name, " ".join(hex(a.value or 0) if isinstance(a, C.c_void_p) else str(a) if isinstance(a, (int, float)) else "arr" for a in args)))
        sys.stderr.flush()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise SatMVSNativeError("%s failed (code %d): %s" % (name, rc, lib.smvs_last_error().decode()))
    if _SYNC_CALLS:                                          # SMVS_SYNC_CALLS=1: wait for every native call (debugging: nothing native is in flight afterwards)
        torch.cuda.synchronize()


def current_stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_device(*tensors):
    """All tensors on one HIP device; returns it.  CPU tensors are an error, not a fallback."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise SatMVSNativeError(
                "satmvs_amd operators run on an MI355X only: got a %s tensor (no CPU fallback)" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise SatMVSNativeError("tensors on different devices: %s vs %s" % (dev, t.device))
    return dev


def ptr(t):
    return C.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """Host array of device pointers (for the `const float* const*` arguments)."""
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
