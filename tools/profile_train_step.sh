#!/bin/bash
# kernel trace of one training step: tools/profile_train_step.sh [casred|casmvs|ucs]
MODEL=${1:-casred}
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_train_$MODEL
mkdir -p "$OUT"
python $REPO/tools/bench_train_step.py 5 $MODEL > "$OUT/train_step.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python $REPO/tools/bench_train_step.py 3 $MODEL > "$OUT/trace.log" 2>&1
python - <<PY >> "$OUT/train_step.txt"
import sqlite3, glob
db = sqlite3.connect(glob.glob("$OUT/trace/**/*.db", recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, count(*), sum(d.end-d.start)/1e6 from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kd, ks)).fetchall()
tot = sum(r[2] for r in rows)
print("kernel time over 5 steps (2 warm-up + 3 timed): %.1f ms in %d launches; top kernels:" % (tot, sum(r[1] for r in rows)))
for r in rows[:40]:
    print("  %6.1f ms %5.1f %%  n=%6d  %s" % (r[2], 100 * r[2] / tot, r[1], r[0][:110]))
PY
find "$OUT" -name "*.db" -delete
cat "$OUT/train_step.txt"
