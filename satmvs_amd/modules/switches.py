"""Every A/B and debugging switch the Python layer takes from the environment, read in ONE place.

The C library never reads the environment (tuning switches exist only in -DSMVS_TUNING builds); the package around it has a few
switches for bisecting the training path against torch's own operators and for forcing the stock PyTorch composites.  They live on
the `SW` object below: the SMVS_TRAIN_* family is read once, when the package is imported (tests flip the attributes); the three
SMVS_{RED,COSTREG,FEATNET}_TORCH switches are looked up per forward (tests set them around single forwards).  The arithmetic of the
variance build is not a switch of this file: satmvs_amd.set_arith / arith_scope / the networks' `arith=` argument (satmvs_amd/_lib.py).
"""
from __future__ import annotations

import os

import torch


class _Switches:
    def __init__(self):
        self.reload()
        self.find_warned = False             # guard_miopen_find() has warned once
        self.find_switched_off = False       # ... and holds the caller's cudnn.benchmark = True for restore_miopen_find()

    def reload(self, env=None):
        e = (os.environ if env is None else env).get
        # SMVS_TRAIN_COMPOSITE=1 keeps the training path on torch's own operators (A/B against the native ones; the cost-volume
        # operators are native either way); SMVS_TRAIN_COMPOSITE_MASK bisects: 1 GroupNorm, 2 cat(x, r*h), 4 u-blend, 8 conv weight
        # gradient, 16 ConvGRU convolutions (forward + input gradient), 32 the cell as one autograd node, 64 CostRegNet's 3-D weight gradient, 128 its forward + input gradient, 256 its BatchNorm3d + ReLU
        self.train_composite_mask = 511 if e("SMVS_TRAIN_COMPOSITE", "0") == "1" else int(e("SMVS_TRAIN_COMPOSITE_MASK", "0"))
        self.train_streams = e("SMVS_TRAIN_STREAMS", "1") != "0"              # ConvGRU levels 1-3 of a plane on side streams
        self.train_loop_pipeline = e("SMVS_TRAIN_LOOP_PIPELINE", "1") != "0"  # ... and the plane loop software-pipelined (cells d | encoder d+1 | decoder d-1)
        self.train_plane_views = e("SMVS_TRAIN_PLANE_VIEWS", "1") != "0"      # per-plane parameter views (one gradient sum per parameter)
        self.train_defer_wgrad = e("SMVS_TRAIN_DEFER_WGRAD", "1") != "0"      # weight gradients of a layer in one launch over all planes
        self.train_featnet_native = e("SMVS_TRAIN_FEATNET_NATIVE", "1") != "0"  # FeatureNet's 3x3 layers on the native layer kernels under autograd
        self.allow_miopen_find = e("SMVS_ALLOW_MIOPEN_FIND", "0") == "1"      # leave torch.backends.cudnn.benchmark alone while training

    @staticmethod
    def force_composite(which):
        """which in "RED", "COSTREG", "FEATNET": SMVS_<which>_TORCH=1 forces the stock PyTorch composite of that module (A/B)."""
        return os.environ.get("SMVS_%s_TORCH" % which) == "1"


SW = _Switches()


def guard_miopen_find():
    """Training forwards of the RED cascades switch `torch.backends.cudnn.benchmark` off (the reference's train.py:21 turns it on)
    and restore_miopen_find() puts it back at the next inference forward.

    On this image (ROCm 7.0 / PyTorch 2.10, MI355X) MIOpen's exhaustive search ends the training forward of this network at the
    768x384 tile in a GPU memory access fault -- in a process that never maps this library: tools/miopen_find_repro.py builds the
    same forward from torch operators only (torch convolutions, F.group_norm, F.grid_sample), checks /proc/self/maps, and faults in
    stage 2 after ~70 s (profiles/r04_miopen_find_repro.txt; round 3 saw the same with the native operators serialised and
    synchronised, 6 of 6 runs).  The fault is inside the search (its candidate kernels / workspaces), and where it does not fault
    the search costs ~7 minutes per process.  MIOpen's default (immediate-mode) choices are what every test, fixture and timing
    of this repository uses.  Only the RED cascades call this (the fault was reproduced for them); SMVS_ALLOW_MIOPEN_FIND=1
    leaves the flag alone."""
    if torch.backends.cudnn.benchmark and not SW.allow_miopen_find:
        torch.backends.cudnn.benchmark = False
        SW.find_switched_off = True
        if not SW.find_warned:
            SW.find_warned = True
            import warnings
            warnings.warn("satmvs_amd: torch.backends.cudnn.benchmark switched off while the RED cascade trains (MIOpen's search faults in "
                          "this network's training forward on this ROCm build, with or without this library in the process -- see "
                          "satmvs_amd.modules.switches.guard_miopen_find; it is restored at the next inference forward; "
                          "SMVS_ALLOW_MIOPEN_FIND=1 leaves the flag alone)")


def restore_miopen_find():
    """Inference forward after a guarded training forward: give the caller's cudnn.benchmark = True back."""
    if SW.find_switched_off:
        SW.find_switched_off = False
        torch.backends.cudnn.benchmark = True
