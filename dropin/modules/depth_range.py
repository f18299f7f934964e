"""Reference module name `modules.depth_range` -> satmvs_amd.modules.depth_range (dropin/README.md)."""
from satmvs_amd.modules.depth_range import *  # noqa: F401,F403
from satmvs_amd.modules import depth_range as _impl

globals().update({n: getattr(_impl, n) for n in dir(_impl) if not n.startswith("_") and n != "annotations"})
__all__ = [n for n in dir(_impl) if not n.startswith("_") and n != "annotations"]
