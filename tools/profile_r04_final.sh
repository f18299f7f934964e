#!/bin/bash
# End-of-round refresh of the files under profiles/ that depend on the RED / training code (the cost-volume kernel's own evidence --
# power sweep, PMC passes, kernel traces -- comes from tools/profile_r04.sh and is unchanged since): tools/profile_r04_final.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r04_final
mkdir -p "$OUT"
python $REPO/bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"
python $REPO/bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python $REPO/tools/host_bound_probe.py > "$OUT/host_bound_probe.txt" 2>&1
echo "# NOT under the profiler" > "$OUT/models_timing.txt"
for t in bench_pred bench_casred_eval bench_casmvs_eval bench_costreg bench_featnet bench_bwd; do python $REPO/tools/$t.py >> "$OUT/models_timing.txt" 2>&1; done
SMVS_BENCH_BATCH=8 python $REPO/tools/bench_pred.py >> "$OUT/models_timing.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace_models" -o trace -- python $REPO/tools/run_native_models.py > "$OUT/models_profiled.txt" 2>&1
python $REPO/tools/rocpd_summary.py "$OUT" > "$OUT/summary_models.txt" 2>&1
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
