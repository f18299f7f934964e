#!/bin/bash
# On the GPU box: memory-side PMC passes of the cfg4 8-plane shard for the given gpurun_ab builds (one rocprofv3 run per counter set,
# never combined with tracing).   tools/profile_cfg4_pmc.sh <outdir> <build> [<build> ...]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  BENCH="python $REPO/bench.py --no-cpu-baseline --no-extra --workload ${PMC_WORKLOAD:-cfg4_rpc_5view_1536x768x8_c32} --steps 3 --warmup 1 --prewarm-seconds 0.05"
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE" ${PMC_EXTRA:+"$PMC_EXTRA"}; do
    name=$(echo $set | tr ' ' '+' | cut -c1-40)
    SMVS_LIB_PATH=$REPO/gpurun_ab/$v.so rocprofv3 --pmc $set -d "$OUT/pmc_${v}_$name" -o pmc -- $BENCH > "$OUT/pmc_${v}_$name.log" 2>&1 || echo "failed: $v $set" >> "$OUT/errors.log"
  done
done
python $REPO/tools/rocpd_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -E "costvol_dma|^==" "$OUT/summary.txt" | cut -c1-170
cat "$OUT/errors.log" 2>/dev/null
