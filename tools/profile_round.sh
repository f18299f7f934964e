#!/bin/bash
# Round-end evidence set (run on the GPU box): tools/profile_round.sh <tag>
#   1. default bench line (un-profiled)                                  -> gpurun_out/prof_<tag>/bench.json
#   2. kernel trace + PMC passes of the bench command (tools/profile.sh) -> trace/, pmc_*/
#   3. kernel trace of the inference cascade (tools/bench_pred.py)       -> cascade/
#   4. MFMA counters of the regularisers (bench_costreg / bench_red)     -> mfma_*/
set -u
TAG=${1:-r01_final}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
python $REPO/bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
bash $REPO/tools/profile.sh $TAG > "$OUT/profile_sh.log" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace_cascade" -o trace -- python $REPO/tools/bench_pred.py > "$OUT/cascade.log" 2>&1
for cmd in bench_costreg bench_red; do
  rocprofv3 --kernel-trace --stats -d "$OUT/trace_$cmd" -o trace -- python $REPO/tools/$cmd.py > "$OUT/$cmd.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d "$OUT/pmc_mfma_$cmd" -o pmc -- python $REPO/tools/$cmd.py > "$OUT/pmc_mfma_$cmd.log" 2>&1
done
python $REPO/tools/rocpd_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
# the raw rocpd databases exceed what gpurun copies back: keep the text extracts only
find "$OUT" -name "*.db" -delete
rm -f "$OUT/counters_available.txt"
du -sh "$OUT"
tail -3 "$OUT/bench.json"
