#!/bin/bash
# cfg4 shard (5-view 1536x768x8, C=32) counters on the round-6 kernels: kernel trace + TCP / fetch / write PMC passes, each in its own run.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_cfg4_r06
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
WL="--workload cfg4_rpc_5view_1536x768x8_c32"
rocprofv3 --kernel-trace --stats -d "$OUT/trace_cfg4" -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extra $WL > "$OUT/trace_cfg4_bench.json" 2> "$OUT/trace.err"
BENCH="python $REPO/bench.py --no-cpu-baseline --no-extra $WL --steps 3 --warmup 1 --prewarm-seconds 0.05"
for set in "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
python $REPO/tools/rocpd_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
