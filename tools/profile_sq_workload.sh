#!/bin/bash
# SQ counters of one bench workload: tools/profile_sq_workload.sh <tag> <workload> [SMVS_ARITH]   (on the GPU box; each PMC set in its own pass)
set -u
TAG=${1:-sq}; WL=${2:-cfg2_rpc_3view_768x384x64_c32}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-extra --workload $WL --steps 3 --warmup 1 --prewarm-seconds 0.05"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extra --workload $WL > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
python $REPO/tools/rocpd_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -v "rocclr\|at::native\|elementwise" "$OUT/summary.txt" | cut -c1-200
