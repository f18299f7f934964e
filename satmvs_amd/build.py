"""Build recipe for libsatmvs_hip.so (hipcc, gfx950 only, in-tree so the .so ships with the repo snapshot).

    python -m satmvs_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this runs in the build container as well as on the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libsatmvs_hip.so")
SOURCES = ["costvol.hip", "costvol_fused.hip", "costvol_bwd.hip", "warp.hip", "regress.hip", "red.hip", "costreg.hip", "featnet.hip", "filter.hip", "groupnorm.hip", "conv_wgrad.hip", "batchnorm.hip"]
HEADERS = ["smvs_device.h", "smvs_host.h", "mfma_conv.h", "costvol_kernels.h", os.path.join("..", "..", "include", "satmvs.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fvisibility=hidden", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), jobs=None):
    """Compile every HIP source (one hipcc process each, in parallel) and link them into one shared library.  Returns its path."""
    if not force and not _stale():
        return LIB
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIB_DIR, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra_flags)
    with tempfile.TemporaryDirectory(prefix="obj_", dir=LIB_DIR) as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, src + ".o")
            cmd = [hipcc] + cflags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj
        with ThreadPoolExecutor(max_workers=jobs or min(len(SOURCES), os.cpu_count() or 1)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
