#!/bin/bash
# PMC look at an arbitrary command: tools/profile_cmd.sh <tag> <command...>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- "$@" > "$OUT/trace.log" 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc -- "$@" > "$OUT/pmc_$name.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
