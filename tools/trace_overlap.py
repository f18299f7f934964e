#!/usr/bin/env python
"""Summarise a rocprofv3 kernel trace (rocpd .db) per HW queue: busy time, span, and a timeline of one window.
usage: trace_overlap.py <dir-with-db> [n_timeline_rows]"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select start, end, queue_id, stream_id, name from kernels order by start").fetchall()
t0 = rows[0][0]
span = rows[-1][1] - t0
print("kernels %d, span %.2f ms" % (len(rows), span / 1e6))
byq = {}
for s, e, q, st, n in rows:
    byq.setdefault((q, st), []).append((s, e, n))
for k, v in sorted(byq.items()):
    print("queue %s stream %s: %5d kernels, busy %.2f ms" % (k[0], k[1], len(v), sum(e - s for s, e, _ in v) / 1e6))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mid = len(rows) // 2
for s, e, q, st, name in rows[mid:mid + n]:
    print("%10.1f %8.1f  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, name[:60]))
