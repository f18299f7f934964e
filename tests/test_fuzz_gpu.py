"""Randomised parity runs (tests/fuzz/fuzz_*.py) as part of the GPU suite: random shapes, view counts, channel counts, geometries, height
spans (incl. NaN / far-away hypotheses and plane windows) through the kernels against the CPU oracle (bits) resp. the warp operators'
autograd.  Seeds are fixed: a failure is reproducible with the printed command."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(tool, *args, arith=None):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "fuzz", tool)] + [str(a) for a in args]
    env = dict(os.environ, SMVS_ARITH=arith) if arith else None          # the fuzzers inherit the suite's arithmetic ("exact") unless told otherwise
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, "%s\n%s" % (" ".join(cmd), p.stderr[-2000:])
    assert "MISMATCH" not in p.stdout, "%s\n%s" % (" ".join(cmd), p.stdout[-3000:])
    return p.stdout.strip().splitlines()[-1]


def test_fuzz_cost_volume_forward_vs_oracle():
    last = _run("fuzz_costvol_fwd.py", 60, 11)
    assert last.startswith("60 cases: 0 differing voxels"), last


def test_fuzz_plane_coefficient_entry_vs_oracle():
    """smvs_rpc_plane_coef + smvs_rpc_costvol_fwd_pc on random shapes / windows with plane-constant, partly jittered, NaN and fully
    jittered heights: the oracle's volume up to the voxels whose float32 tap coordinate the re-associated cubics flip (bounded per
    case by the fuzzer; in total a handful)."""
    last = _run("fuzz_costvol_fwd.py", 60, 15, "pc")
    assert last.startswith("60 cases:"), last
    assert int(last.split()[2]) <= 60, last


@pytest.mark.parametrize("mode", ["exact", "fused"])
def test_fuzz_cost_volume_backward_vs_warp_autograd(mode):
    """(both arithmetics of the forward: the gradient kernel is the same, the library default must run it too)"""
    last = _run("fuzz_costvol_bwd.py", 60, 12, arith=mode)
    assert last.startswith("60 cases, worst relative error"), last
    assert float(last.split()[-1]) <= 1e-4, last


def test_fuzz_small_operators_vs_oracle():
    last = _run("fuzz_ops.py", 30, 13)
    assert last == "30 rounds, 0 mismatching checks", last


def test_fuzz_training_operators_3d_vs_float64():
    """smvs_conv3d_fwd / its adjoint / smvs_conv3d_wgrad and smvs_batchnorm_train_* on random layers against float64 (round 5)."""
    last = _run("fuzz_train3d.py", 60, 14)
    assert last.startswith("60 cases, worst relative error"), last
