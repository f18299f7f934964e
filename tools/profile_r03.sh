#!/bin/bash
# Round-3 evidence set (run on the GPU box): tools/profile_r03.sh
#   bench lines (driver flags, default), kernel trace of the default bench command, SQ / TA / traffic PMC passes (each in
#   its own rocprofv3 run, never combined with tracing), power / clock samples of the kernel and its ablation builds,
#   native-model kernel trace + MFMA counters, micro-benchmarks.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r03
mkdir -p "$OUT"
python $REPO/bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"
python $REPO/bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
# power: the kernel runs at the board power limit -- sustained runs of the round-2 library, the current one and its ablations
(cd $REPO && PS_STEPS=5000 PS_DELAY=3.2 tools/power_sweep.sh base f_cur f_a1 f_a2 f_a4 f_a8 f_a32 f_a47 f_cur base) > "$OUT/power.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extra > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
BENCH="python $REPO/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --prewarm-seconds 0.05"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
# native models: kernel trace, then MFMA counters (separate passes)
rocprofv3 --kernel-trace --stats -d "$OUT/trace_models" -o trace -- python $REPO/tools/run_native_models.py > "$OUT/models.txt" 2>&1
for set in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  SMVS_RUNS=2 rocprofv3 --pmc $set -d "$OUT/pmcm_$name" -o pmc -- python $REPO/tools/run_native_models.py > "$OUT/pmcm_$name.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
python $REPO/tools/rocpd_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
python $REPO/tools/make_traffic_json.py "$OUT" > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"
python $REPO/tools/host_bound_probe.py > "$OUT/host_bound_probe.txt" 2>&1
for t in bench_pred bench_casred_eval bench_casmvs_eval bench_costreg bench_featnet bench_bwd; do python $REPO/tools/$t.py >> "$OUT/models_timing.txt" 2>&1; done
SMVS_BENCH_BATCH=8 python $REPO/tools/bench_pred.py >> "$OUT/models_timing.txt" 2>&1
for u in gridbar store; do [ -x $REPO/gpurun_ab/ubench_$u ] && timeout 120 $REPO/gpurun_ab/ubench_$u > "$OUT/ubench_$u.txt" 2>&1; done
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
