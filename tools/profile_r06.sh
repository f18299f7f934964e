#!/bin/bash
# Round-6 evidence set (run on the GPU box): tools/profile_r06.sh
#   bench lines (driver flags, default), power / clock samples of the headline launch (plane coefficients vs the trivariate chain, and two
#   ablation builds if present: gpurun_ab/f_a1.so stores dropped, f_a4.so no float64 chain), kernel trace of the default bench command,
#   SQ / cache / traffic PMC passes of the headline launch (each counter set in its own rocprofv3 run, never combined with tracing), the
#   exact instance's trace, the cfg4 shard's trace, un-profiled model / backward timings.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r06
mkdir -p "$OUT"
# ONLY_POWER=1: just the power / clock table (power.txt)
if [ -z "${ONLY_POWER:-}" ]; then
python $REPO/bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"
python $REPO/bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
fi
{
  echo "# idle board (no process on the GPU): rocm-smi --showpower --showclocks"
  sleep 3; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo
  echo "# sustained runs (6000 launches of bench.py --no-extra), rocm-smi power / sclk sampled 4 times from 3.2 s after process start"
  for v in pc tri pc tri; do
    if [ $v = tri ]; then export SMVS_BENCH_TRIVARIATE=1; else unset SMVS_BENCH_TRIVARIATE; fi
    python $REPO/bench.py --no-cpu-baseline --no-extra --steps 6000 --warmup 10 > /tmp/ps_$v.out 2>&1 &
    pid=$!; sleep 3.2; p=""; c=""
    for i in 1 2 3 4; do
      s=$(rocm-smi --showpower --showclocks 2>/dev/null)
      p="$p $(echo "$s" | grep -o 'Power (W): [0-9.]*' | grep -o '[0-9.]*$')"; c="$c $(echo "$s" | grep sclk | grep -o '([0-9]*Mhz' | tr -d '(Mhz')"; sleep 0.4
    done
    wait $pid
    echo "$v (plane coefficients = shipped; tri = smvs_rpc_costvol_fwd, the 20-coefficient chain): $(tail -1 /tmp/ps_$v.out | grep -o 'ms_per_step": [0-9.]*')  W:$p  sclk:$c"
  done
  unset SMVS_BENCH_TRIVARIATE
  ls $REPO/gpurun_ab/f_a1.so $REPO/gpurun_ab/f_a4.so > /dev/null 2>&1 && (cd $REPO && PS_STEPS=6000 PS_DELAY=3.2 tools/power_sweep.sh f_a1 f_a4)
} > "$OUT/power.txt" 2>&1
[ -n "${ONLY_POWER:-}" ] && exit 0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extra > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
BENCH="python $REPO/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --prewarm-seconds 0.05"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INST_LEVEL_VMEM" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" \
           "GRBM_GUI_ACTIVE" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
SMVS_ARITH=exact rocprofv3 --kernel-trace --stats -d "$OUT/trace_exact" -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extra > "$OUT/trace_exact_bench.json" 2> "$OUT/trace_exact.err"
rocprofv3 --kernel-trace --stats -d "$OUT/trace_cfg4" -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extra --workload cfg4_rpc_5view_1536x768x8_c32 > "$OUT/trace_cfg4_bench.json" 2> "$OUT/trace_cfg4.err"
rocprofv3 --kernel-trace --stats -d "$OUT/trace_models" -o trace -- python $REPO/tools/run_native_models.py > "$OUT/models_profiled.txt" 2>&1
python $REPO/tools/rocpd_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
python $REPO/tools/make_traffic_json.py "$OUT" > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"
echo "# NOT under the profiler" > "$OUT/models_timing.txt"
for t in bench_pred bench_casred_eval bench_casmvs_eval bench_costreg bench_featnet bench_bwd bench_train_graph; do python $REPO/tools/$t.py >> "$OUT/models_timing.txt" 2>&1; done
find "$REPO/gpurun_out" -name "*.db" -delete
du -sh "$OUT"
