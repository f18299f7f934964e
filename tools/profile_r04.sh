#!/bin/bash
# Round-4 evidence set (run on the GPU box): tools/profile_r04.sh
#   bench lines (driver flags, default), power / clock samples of both arithmetic instances, their 3-waves-per-SIMD builds and
#   the fused instance's ablation builds (gpurun_ab/*.so from tools/ab_build.sh), kernel trace of the default bench command,
#   SQ / TA / traffic PMC passes (each in its own rocprofv3 run, never combined with tracing), native-model timings.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r04
mkdir -p "$OUT"
python $REPO/bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"
python $REPO/bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
{
  echo "# sustained runs (6000 launches), rocm-smi power / sclk sampled 4 times from 3.2 s after process start; tools/power_sweep.sh"
  echo "# fused arithmetic (library default): shipped kernel, 3 waves per SIMD (-DSMVS_WPS_DP8_FUSED=3, VGPR budget 168), ablations (wrong results): a1 stores dropped, a2 no staging DMA, a4 no float64 chain"
  (cd $REPO && PS_STEPS=6000 PS_DELAY=3.2 tools/power_sweep.sh f_cur f_w3 f_a1 f_a2 f_a4 f_cur)
  echo "# exact arithmetic (SMVS_ARITH=exact): shipped kernel, 3 waves per SIMD (-DSMVS_WPS_DP8=3: the row round 3 reported without a file)"
  (cd $REPO && SMVS_ARITH=exact PS_STEPS=6000 PS_DELAY=3.2 tools/power_sweep.sh f_cur e_w3 f_cur)
} > "$OUT/power.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extra > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
BENCH="python $REPO/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --prewarm-seconds 0.05"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
# the exact instance: kernel trace only
SMVS_ARITH=exact rocprofv3 --kernel-trace --stats -d "$OUT/trace_exact" -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extra > "$OUT/trace_exact_bench.json" 2> "$OUT/trace_exact.err"
# native models: kernel trace (this file's timings are UNDER THE PROFILER), then un-profiled timings
rocprofv3 --kernel-trace --stats -d "$OUT/trace_models" -o trace -- python $REPO/tools/run_native_models.py > "$OUT/models_profiled.txt" 2>&1
python $REPO/tools/rocpd_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
python $REPO/tools/make_traffic_json.py "$OUT" > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"
python $REPO/tools/host_bound_probe.py > "$OUT/host_bound_probe.txt" 2>&1
echo "# NOT under the profiler" > "$OUT/models_timing.txt"
for t in bench_pred bench_casred_eval bench_casmvs_eval bench_costreg bench_featnet bench_bwd; do python $REPO/tools/$t.py >> "$OUT/models_timing.txt" 2>&1; done
SMVS_BENCH_BATCH=8 python $REPO/tools/bench_pred.py >> "$OUT/models_timing.txt" 2>&1
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
