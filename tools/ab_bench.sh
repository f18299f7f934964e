#!/bin/bash
# On the GPU box: alternate the default bench over the given gpurun_ab builds, 3 rounds.  tools/ab_bench.sh base cand1 ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do for v in "$@"; do
  echo -n "$v: "; SMVS_LIB_PATH=$REPO/gpurun_ab/$v.so python $REPO/bench.py --no-cpu-baseline 2>&1 | tail -1 | grep -o 'ms_per_step": [0-9.]*'
done; done
