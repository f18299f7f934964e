#!/usr/bin/env python
"""Height parity of the well-conditioned 768x384 RED inference cascade (tests/test_full_size_red_conditioned.py's case: photo-consistent
rendered feature pyramids, trained-like regulariser weights, softmax confidence >= 0.5) in BOTH arithmetic modes, as one JSON line:
per stage, the native pipeline against a float64 evaluation of the stage on the reference's (exact) variance volume -- the closest
stand-in for the reference path at a size the reference's own CPU code cannot be asked for on the GPU box.  bench.py runs this in a
subprocess and files the line under extra.height_parity_vs_reference.conditioned_768x384.
    python tools/conditioned_parity.py [redinf|red]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SMVS_ARITH", "exact")
import torch  # noqa: E402

import test_full_size_red_conditioned as C  # noqa: E402
from test_full_size_cascade import red_stages_against_float64  # noqa: E402
from satmvs_amd import _lib  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "redinf"
dev = torch.device("cuda:0")
net, imgs, pm, dv, truth, gains, confs = C.conditioned_case(tag, dev)
rec = {"network": {"redinf": "Infer_CascadeREDNet", "red": "CascadeREDNet"}[tag], "tile": "3-view 768x384, planes 48/32/8",
       "mean_confidence": {s: round(c, 3) for s, c in confs.items()}, "target_m": 1e-3,
       "what": "max |native - float64 evaluation of the stage on the EXACT variance volume| per stage, same stage inputs"}
for mode in ("exact", "fused"):
    with _lib.arith_scope(mode):
        det = {}
        red_stages_against_float64(net, imgs, pm, dv, "rpc", detail=det)
    err = {s: (d["native"].double() - d["float64"]).abs() for s, d in det.items()}
    rec[mode] = {"max_abs_m": {s: float("%.3g" % float(e.max())) for s, e in err.items()},
                 "fraction_beyond_1e-3_m": {s: float("%.2g" % float((e > 1e-3).double().mean())) for s, e in err.items()}}
rec["default_of_cascades_and_plane_pipelines"] = "exact (a model built with arith='fused', or an enclosing arith_scope('fused'), selects the fused build)"
print(json.dumps(rec))
