#!/bin/bash
# The passes of tools/profile_r05.sh whose databases its first run deleted before tools/rocpd_summary.py looked at them (it globbed pmc_* and
# pmcm_* only): SQ counters of the cfg4 shard, MFMA counters + kernel trace at 8 tiles per forward.  tools/profile_r05_extra.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r05x
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C4="python $REPO/bench.py --no-cpu-baseline --no-extra --workload cfg4_rpc_5view_1536x768x8_c32 --steps 3 --warmup 1 --prewarm-seconds 0.05"
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INST_LEVEL_VMEM" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set -d "$OUT/pmc4_$name" -o pmc -- $C4 > "$OUT/pmc4_$name.log" 2>&1 || echo "failed: cfg4 $set" >> "$OUT/errors.log"
done
rocprofv3 --kernel-trace --stats -d "$OUT/trace_c4" -o trace -- $C4 > "$OUT/trace_c4.log" 2>&1
SMVS_BENCH_BATCH=8 rocprofv3 --kernel-trace --stats -d "$OUT/trace_b8" -o trace -- python $REPO/tools/bench_pred.py > "$OUT/b8_profiled.txt" 2>&1
SMVS_BENCH_BATCH=8 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d "$OUT/pmcb8_SQ_INSTS_VALU_MFMA" -o pmc -- python $REPO/tools/bench_pred.py > "$OUT/pmcb8.log" 2>&1
python $REPO/tools/rocpd_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
python $REPO/tools/mfma_util.py "$OUT/summary.txt" trace_b8 pmcb8_SQ_INSTS_VALU_MFMA > "$OUT/mfma_b8.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -E "^==|costvol_dma" "$OUT/summary.txt" | cut -c1-180
cat "$OUT/mfma_b8.txt"
