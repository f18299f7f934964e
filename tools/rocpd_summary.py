#!/usr/bin/env python
"""Summarise rocprofv3 (rocpd sqlite) output: per-kernel time stats and per-kernel mean PMC values.

    python tools/rocpd_summary.py gpurun_out/prof_r01 > profiles/r01_summary.txt
"""
import glob
import os
import sqlite3
import sys


def kernel_stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                       "max(grid_x), max(workgroup_x*workgroup_y*workgroup_z) from kernels group by name "
                       "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-72s %6s %12s %10s %10s %10s %6s  %s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct",
                                                    "vgpr/agpr/sgpr lds scratch grid wg"))
    for r in rows:
        print("%-72s %6d %12.1f %10.1f %10.1f %10.1f %6.2f  %s/%s/%s %s %s %s %s" % (
            r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot,
            r[6], r[7], r[8], r[9], r[10], r[11], r[12]))


def pmc_stats(db):
    con = sqlite3.connect(db)
    try:
        rows = con.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                           "from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        print("   (unexpected schema: %s)" % cols)
        return
    for r in rows:
        print("   %-60s %-34s n=%-4d mean=%-16.6g min=%-14.6g max=%-14.6g" % (r[0][:60], r[1], r[2], r[3], r[4], r[5]))


def main(root):
    tr = glob.glob(os.path.join(root, "trace*", "*.db"))
    for db in tr:
        print("== kernel trace: %s" % os.path.relpath(db, root))
        kernel_stats(db)
    for db in sorted(glob.glob(os.path.join(root, "pmc*", "*.db"))):
        print("== pmc pass: %s" % os.path.basename(os.path.dirname(db)))
        pmc_stats(db)


if __name__ == "__main__":
    main(sys.argv[1])
