#!/usr/bin/env python
"""Timeline of the RED plane loop from a rocprofv3 kernel trace (rocpd sqlite): for each stage of
tools/host_bound_probe.py print a window of consecutive dispatches (start offset, duration, queue, kernel) and the
per-kernel mean durations inside that stage.

    rocprofv3 --kernel-trace -d /tmp/prof -o t -- python tools/host_bound_probe.py ; python tools/red_timeline.py /tmp/prof
"""
import glob, os, sqlite3, sys
from collections import defaultdict

db = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = con.execute("select start, end, %s, name, grid_x from kernels order by start" % qcol).fetchall()
# stages are told apart by the grid of the chunked cost-volume kernel
stages = []
cur = None
for i, r in enumerate(rows):
    if "costvol" in r[3]:
        key = r[3]
        if cur is None or cur[0] != key:
            cur = [key, i, i]
            stages.append(cur)
    if cur is not None:
        cur[2] = i
for key, i0, i1 in stages:
    seg = rows[i0:i1 + 1]
    n = len(seg)
    print("== stage with %s: %d dispatches over %.2f ms" % (key[:60], n, (seg[-1][1] - seg[0][0]) / 1e6))
    agg = defaultdict(list)
    for s in seg[n // 2:]:
        agg[s[3]].append((s[1] - s[0]) / 1e3)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("   %-80s n=%4d mean %7.2f us  total %8.1f us" % (k[:80], len(v), sum(v) / len(v), sum(v)))
    w = seg[n // 2: n // 2 + int(sys.argv[2]) if len(sys.argv) > 2 else n // 2 + 40]
    t0 = w[0][0]
    for s in w:
        print("   +%8.2f us  %7.2f us  q=%-4s %s" % ((s[0] - t0) / 1e3, (s[1] - s[0]) / 1e3, s[2], s[3][:70]))
