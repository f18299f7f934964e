"""Reference module name `networks.ucs` -> satmvs_amd.networks.ucs (dropin/README.md)."""
from satmvs_amd.networks.ucs import *  # noqa: F401,F403
from satmvs_amd.networks import ucs as _impl

globals().update({n: getattr(_impl, n) for n in dir(_impl) if not n.startswith("_") and n != "annotations"})
__all__ = [n for n in dir(_impl) if not n.startswith("_") and n != "annotations"]
