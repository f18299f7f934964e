#!/bin/bash
# Kernel trace of tools/bench_bwd.py for one or more library variants: tools/profile_bwd.sh <lib.so> [<lib.so> ...]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  OUT=$REPO/gpurun_out/prof_bwd_$tag
  mkdir -p "$OUT"
  SMVS_LIB_PATH=$lib rocprofv3 --kernel-trace --stats -d "$OUT" -o trace -- python $REPO/tools/bench_bwd.py > "$OUT/run.log" 2>&1
  echo "== $tag"; tail -3 "$OUT/run.log"
  python $REPO/tools/rocpd_summary.py "$OUT" 2>/dev/null | grep -v rocclr | cut -c1-200 | head -14
done
