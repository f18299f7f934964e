#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box.  Usage: tools/profile.sh r01 [workload]
# Writes raw output under gpurun_out/prof_<tag>/ and the summaries to copy into profiles/.
set -u
TAG=${1:-r01}
WL=${2:-cfg2_rpc_3view_768x384x64_c32}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --workload $WL --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH --steps 100 --warmup 10 > "$OUT/trace.log" 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-60)
  rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc -- $BENCH --steps 3 --warmup 1 > "$OUT/pmc_$name.log" 2>&1 || echo "pmc set failed: $set" >> "$OUT/errors.log"
done
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
ls -R "$OUT" | head -80
