#!/bin/bash
# kernel trace of the default bench command on the final library (the agreement check between bench.py's live HIP-event
# timing and rocprofv3): tools/profile_final_trace.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_final
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extra > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
python - <<PY > "$OUT/summary.txt"
import sqlite3, glob, json
db = sqlite3.connect(glob.glob("$OUT/trace/**/*.db", recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 6 desc" % (kd, ks)).fetchall()
print("rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra   (final round-3 library)")
line = json.loads(open("$OUT/trace_bench.json").read().strip().splitlines()[-1])
print("bench.py inside this run: ms_per_step %.4f, roofline.kernel_ms %.4f, frac %.4f" % (line["ms_per_step"], line["roofline"]["kernel_ms"], line["roofline"]["frac"]))
for r in rows[:8]:
    print("  %-90s n=%5d avg %9.1f us  min %9.1f  max %9.1f" % (r[0][:90], r[1], r[2], r[3], r[4]))
PY
find "$OUT" -name "*.db" -delete
cat "$OUT/summary.txt"
