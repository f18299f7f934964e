#!/bin/bash
# Build the working tree's library into gpurun_ab/<name>.so (travels with gpurun, git-ignored) for same-box A/B runs:
#   tools/ab_build.sh cand [-Dflags...] && gpurun -- 'SMVS_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_ab/cand.so python bench.py ...'
# AB_SRC=<file.hip>[,<file.hip>...] picks the sources the flags apply to (default costvol.hip,costvol_fused.hip: the exact and the
# fused instances of the cost-volume kernels live in one source each).
# Objects of the other sources are cached in gpurun_ab/obj (rebuilt when a source or header is newer);
# AB_FAST=1 adds -DSMVS_ONLY_BENCH (only the instances the headline bench launches: compiles in seconds, bench.py
# --no-extra only).
set -e
name=${1:?name}; shift; export AB_FLAGS="$*"
cd "$(dirname "$0")/.."
mkdir -p gpurun_ab/obj
python - <<PY
import os, subprocess, sys
sys.path.insert(0, ".")
from satmvs_amd import build as b
flags = [f for f in b.FLAGS if f != "-shared"]
hdrs = [os.path.join(b.CSRC, h) for h in b.HEADERS]
var = os.environ.get("AB_SRC", "costvol.hip,costvol_fused.hip").split(",")
objs = []
for s in b.SOURCES:
    if s in var:
        continue
    src, obj = os.path.join(b.CSRC, s), "gpurun_ab/obj/%s.o" % s
    if not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in [src] + hdrs):
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", src, "-o", obj], stderr=subprocess.DEVNULL)
    objs.append(obj)
extra = os.environ.get("AB_FLAGS", "").split() + (["-DSMVS_ONLY_BENCH"] if os.environ.get("AB_FAST") == "1" else [])
cvs = []
for i, v in enumerate(var):
    cv = "gpurun_ab/obj/var%d_$name.o" % i
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + extra + ["-c", os.path.join(b.CSRC, v), "-o", cv], stderr=subprocess.DEVNULL)
    cvs.append(cv)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + cvs + ["-o", "gpurun_ab/$name.so"])
for cv in cvs:
    os.remove(cv)
print("built gpurun_ab/$name.so")
PY
