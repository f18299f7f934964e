#!/bin/bash
# kernel breakdown of Infer_CascadeREDNet forwards at SMVS_BENCH_BATCH tiles per forward: tools/profile_pred_batch.sh [batch]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_pred_b${1:-8}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SMVS_BENCH_BATCH=${1:-8} rocprofv3 --kernel-trace -d "$OUT" -o trace -- python $REPO/tools/bench_pred.py > "$OUT/run.log" 2>&1
tail -2 "$OUT/run.log"
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$OUT/**/*.db", recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 4 desc" % (kd, ks)).fetchall()
tot = sum(r[3] for r in rows)
print("total kernel time %.1f ms" % (tot / 1e3))
for r in rows[:14]:
    print("  %8.1f us avg  n=%5d  %5.1f %%  %s" % (r[2], r[1], 100 * r[3] / tot, r[0][:100]))
PY
find "$OUT" -name "*.db" -delete
