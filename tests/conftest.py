import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The bit-level suite runs the reference-rounding instance of the variance build (smvs_set_arith, include/satmvs.h); the
# library's default, the fused arithmetic, is covered at the contract tolerances by tests/test_fused_arith.py and by the
# reference-golden end-to-end tests that take the `arith` fixture (both modes).  Through the environment so that test
# subprocesses (fuzzers, shard ranks, DataParallel workers) start in the same mode.
os.environ.setdefault("SMVS_ARITH", "exact")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _finalise_between_tests(request):
    """GPU tests: objects that own HIP resources (graphs with private pools, streams, events, big tensors) are finalised BETWEEN
    tests, with the device idle -- not by a cyclic collection that happens to run in the middle of a later test's capture or
    side-stream work (seen as a segmentation fault "Garbage-collecting" inside test_graphed_training_step_matches_eager's warm-up,
    one run in four, round 6)."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        gc.collect()


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        return cache[name]

    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(params=["exact", "fused"])
def arith(request):
    """Runs the test once per arithmetic of the variance build -- the process default of the stand-alone builds AND an
    arith_scope, which the plane pipelines / cascades follow (without one they run "exact" whatever the default is);
    leaves the suite's mode ("exact") behind."""
    from satmvs_amd import _lib
    _lib.set_arith(request.param)
    with _lib.arith_scope(request.param):
        yield request.param
    _lib.set_arith(os.environ.get("SMVS_ARITH", "exact"))
