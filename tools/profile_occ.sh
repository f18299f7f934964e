#!/bin/bash
# Occupancy / latency look: tools/profile_occ.sh <tag>   (SMVS_LIB_PATH selects the build)
set -u
TAG=${1:-occ}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --steps 3 --warmup 1"
for set in "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM" \
           "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64" \
           "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
