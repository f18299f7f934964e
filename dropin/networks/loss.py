"""`networks.loss` is not on the hot path: forward to the reference checkout's own file (found on sys.path after this
package), so that `from networks.loss import *` keeps working when dropin/ shadows the `networks` package."""
import importlib.util
import os
import sys


def _load():
    here = os.path.dirname(os.path.abspath(__file__))
    for p in sys.path:
        cand = os.path.join(p, "networks", "loss.py")
        if os.path.isfile(cand) and os.path.dirname(os.path.abspath(cand)) != here:
            spec = importlib.util.spec_from_file_location("_reference_networks_loss", cand)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod
    raise ImportError("networks/loss.py of the reference checkout is not on sys.path (put it after dropin/)")


_m = _load()
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
