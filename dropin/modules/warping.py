"""Reference module name `modules.warping` -> satmvs_amd.modules.warping (dropin/README.md)."""
from satmvs_amd.modules.warping import *  # noqa: F401,F403
from satmvs_amd.modules import warping as _impl

globals().update({n: getattr(_impl, n) for n in dir(_impl) if not n.startswith("_") and n != "annotations"})
__all__ = [n for n in dir(_impl) if not n.startswith("_") and n != "annotations"]
