"""satmvs_amd -- MI355X-native RPC plane-sweep cost-volume engine.

Drop-in for the hot path of WHU-GPCV/SatMVS (modules/warping.py, networks/casred.py and twins):
RPC / homography plane-sweep warping -> per-channel variance cost volume -> regularise ->
soft-argmin height, as hand-written HIP kernels for gfx950 behind a C ABI (include/satmvs.h).
See DESIGN.md for the path, data layout and roofline, INTEGRATION.md for the reference-side binding.
"""
__version__ = "0.1.0"


def set_arith(mode):
    """"fused" (default) or "exact": arithmetic of the variance build -- see include/satmvs.h, smvs_set_arith."""
    from . import _lib
    return _lib.set_arith(mode)


def get_arith():
    from . import _lib
    return _lib.get_arith()


def arith_scope(mode):
    """`with satmvs_amd.arith_scope("exact"): ...` -- the cost-volume calls this thread makes inside the block carry their own
    arithmetic (include/satmvs.h, SMVS_CALL_ARITH_*); the process default of set_arith is neither read nor changed.  The network
    classes take the same thing as a constructor argument (`arith=`)."""
    from . import _lib
    return _lib.arith_scope(mode)
