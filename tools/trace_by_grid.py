#!/usr/bin/env python
"""Per-(kernel, grid size) duration statistics of a rocprofv3 kernel trace (rocpd sqlite): tools/trace_by_grid.py <dir> [substring]"""
import glob, os, sqlite3, sys
root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for db in glob.glob(os.path.join(root, "trace*", "*.db")):
    con = sqlite3.connect(db)
    rows = con.execute("select name, grid_x, count(*), avg(duration), min(duration), max(duration) from kernels "
                       "where name like ? group by name, grid_x order by name, grid_x", ("%" + sub + "%",)).fetchall()
    for r in rows:
        print("%-60s grid %8d  n=%5d  avg %8.1f us  min %8.1f  max %8.1f" % (r[0][:60], r[1], r[2], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
