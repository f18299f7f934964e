#!/bin/bash
# Build the working tree's library into gpurun_ab/<name>.so (travels with gpurun, git-ignored) for same-box A/B runs:
#   tools/ab_build.sh cand && gpurun -- 'SMVS_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_ab/cand.so python bench.py ...'
set -e
name=${1:?name}; shift; export AB_FLAGS="$*"
cd "$(dirname "$0")/.."
mkdir -p gpurun_ab
python - <<PY
import os, subprocess, sys
sys.path.insert(0, ".")
from satmvs_amd import build as b
cmd = ["/opt/rocm/bin/hipcc"] + b.FLAGS + os.environ.get("AB_FLAGS", "").split() + [os.path.join(b.CSRC, s) for s in b.SOURCES] + ["-o", "gpurun_ab/$name.so"]
subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
print("built gpurun_ab/$name.so")
PY
