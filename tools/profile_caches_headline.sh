#!/bin/bash
# Instruction-cache and scalar-cache counters of the headline launch: tools/profile_caches_headline.sh <tag>   (SMVS_BENCH_TRIVARIATE=1: round 5's path)
set -u
TAG=${1:-caches}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-extra --prewarm-seconds 0.05"
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
python $REPO/tools/rocpd_summary.py "$OUT" | grep costvol_dma | awk '{for(i=1;i<=NF;i++) if ($i ~ /^mean=/) print $(i-2), $i}'
