#!/usr/bin/env python
"""Mean PMC values per (kernel, grid size) from a rocprofv3 --pmc run (rocpd sqlite): tells the jobs of the level-batched
launches apart when they are launched one by one (SMVS_RED_SPLIT_JOBS=1).   python tools/pmc_by_grid.py <dir> [name filter]"""
import glob, os, sqlite3, sys
from collections import defaultdict
for db in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    gcol = [c for c in cols if c in ("grid_size", "grid_size_x", "grid_x")]
    if not gcol:
        print("no grid column in", cols); continue
    q = "select kernel_name, %s, counter_name, count(*), avg(value) from counters_collection group by kernel_name, %s, counter_name" % (gcol[0], gcol[0])
    acc = defaultdict(dict)
    for name, grid, cn, n, v in con.execute(q):
        if len(sys.argv) > 2 and sys.argv[2] not in name: continue
        acc[(name[:48], grid, n)][cn] = v
    for k in sorted(acc):
        print("%-48s grid %8s n=%-5d " % k + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(acc[k].items())))
