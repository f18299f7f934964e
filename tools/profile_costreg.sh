#!/bin/bash
# per-kernel times of one native CostRegNet forward per stage shape: tools/profile_costreg.sh [lib.so]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_costreg
rm -rf "$OUT"; mkdir -p "$OUT"
[ $# -ge 1 ] && export SMVS_LIB_PATH=$1
cd /tmp && export TMPDIR=/tmp
SMVS_COSTREG_NATIVE_ONLY=1 rocprofv3 --kernel-trace -d "$OUT" -o trace -- python $REPO/tools/run_costreg_once.py > "$OUT/run.log" 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$OUT/**/*.db", recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, d.grid_size_x*d.grid_size_y*d.grid_size_z, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from %s d join %s s on d.kernel_id=s.id where s.kernel_name like '%%smvs%%' group by s.kernel_name, 2 order by 5 desc" % (kd, ks)).fetchall()
tot = sum(r[4] for r in rows)
print("total smvs kernel time %.1f us" % tot)
for r in rows[:40]:
    print("  %8.1f us avg  n=%3d  %5.1f %%  threads %9d  %s" % (r[3], r[2], 100 * r[4] / tot, r[1], r[0][:100]))
PY
