#!/bin/bash
# Round-5 evidence set (run on the GPU box): tools/profile_r05.sh
#   bench lines (driver flags, default), power / clock samples of the headline instance with its ablation builds (gpurun_ab/f_*.so from
#   tools/ab_build.sh) and the board's idle reading, kernel trace of the default bench command, SQ / TA / traffic PMC passes of the
#   headline launch and of the cfg4 shard (each counter set in its own rocprofv3 run, never combined with tracing), backward PMC
#   passes, native-model timings and trace, MFMA counters at 1 and 8 tiles per forward, graphed training step.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r05
mkdir -p "$OUT"
python $REPO/bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"
python $REPO/bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
{
  echo "# idle board (no process on the GPU): rocm-smi --showpower --showclocks"
  sleep 3; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo
  echo "# sustained runs (6000 launches), rocm-smi power / sclk sampled 4 times from 3.2 s after process start; tools/power_sweep.sh"
  echo "# fused arithmetic (library default): shipped kernel and ablations (wrong results): a1 stores dropped, a2 no staging DMA, a4 no float64 chain, a64 DMA instructions with every lane out of range (no traffic), a128 DMA from the first 64 KB of a channel (cache hits), a32 no packed arithmetic (LDS reads kept)"
  (cd $REPO && PS_STEPS=6000 PS_DELAY=3.2 tools/power_sweep.sh f_cur f_a1 f_a2 f_a4 f_a64 f_a128 f_a32 f_cur)
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
# the cfg4 shard: SQ passes (the memory-side passes are in profiles/r05_cfg4_summary.txt)
C4="python $REPO/bench.py --no-cpu-baseline --no-extra --workload cfg4_rpc_5view_1536x768x8_c32 --steps 3 --warmup 1 --prewarm-seconds 0.05"
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set -d "$OUT/pmc4_$name" -o pmc -- $C4 > "$OUT/pmc4_$name.log" 2>&1 || echo "failed: cfg4 $set" >> "$OUT/errors.log"
done
# the exact instance: kernel trace only
SMVS_ARITH=exact rocprofv3 --kernel-trace --stats -d "$OUT/trace_exact" -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extra > "$OUT/trace_exact_bench.json" 2> "$OUT/trace_exact.err"
# native models: kernel trace + MFMA counters (these timings are UNDER THE PROFILER), then un-profiled timings
rocprofv3 --kernel-trace --stats -d "$OUT/trace_models" -o trace -- python $REPO/tools/run_native_models.py > "$OUT/models_profiled.txt" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d "$OUT/pmcm_SQ_INSTS_VALU_MFMA" -o pmc -- python $REPO/tools/run_native_models.py > "$OUT/pmcm.log" 2>&1
SMVS_BENCH_BATCH=8 rocprofv3 --kernel-trace --stats -d "$OUT/trace_b8" -o trace -- python $REPO/tools/bench_pred.py > "$OUT/b8_profiled.txt" 2>&1
SMVS_BENCH_BATCH=8 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d "$OUT/pmcb8_SQ_INSTS_VALU_MFMA" -o pmc -- python $REPO/tools/bench_pred.py > "$OUT/pmcb8.log" 2>&1
python $REPO/tools/rocpd_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
python $REPO/tools/make_traffic_json.py "$OUT" > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"
{ echo "== 1 tile per forward (tools/run_native_models.py)"; python $REPO/tools/mfma_util.py "$OUT/summary.txt"; echo; echo "== 8 tiles per forward (SMVS_BENCH_BATCH=8 tools/bench_pred.py)"; python $REPO/tools/mfma_util.py "$OUT/summary.txt" trace_b8 pmcb8_SQ_INSTS_VALU_MFMA; } > "$OUT/mfma_utilisation.txt" 2>&1
python $REPO/tools/host_bound_probe.py > "$OUT/host_bound_probe.txt" 2>&1
echo "# NOT under the profiler" > "$OUT/models_timing.txt"
for t in bench_pred bench_casred_eval bench_casmvs_eval bench_costreg bench_featnet bench_bwd bench_train_graph; do python $REPO/tools/$t.py >> "$OUT/models_timing.txt" 2>&1; done
SMVS_BENCH_BATCH=8 python $REPO/tools/bench_pred.py >> "$OUT/models_timing.txt" 2>&1
bash $REPO/tools/profile_bwd_pmc2.sh $REPO/satmvs_amd/lib/libsatmvs_hip.so > "$OUT/bwd_pmc.txt" 2>&1
find "$REPO/gpurun_out" -name "*.db" -delete
du -sh "$OUT"
