#!/bin/bash
# Re-measure the headline kernel's HBM traffic after its sources changed (FETCH_SIZE and WRITE_SIZE in separate --pmc passes)
# and write gpurun_out/restamp/pmc_traffic.json for profiles/pmc_traffic.json:  tools/restamp_traffic.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/restamp
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --prewarm-seconds 0.05"
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $set -d "$OUT/pmc_$set" -o pmc -- $BENCH > "$OUT/pmc_$set.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
python $REPO/tools/make_traffic_json.py "$OUT" > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"
find "$OUT" -name "*.db" -delete
cat "$OUT/pmc_traffic.json"
