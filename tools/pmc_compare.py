#!/usr/bin/env python
"""Side-by-side of the per-launch PMC means of the cost-volume kernel from two or more tools/profile_sq.sh runs:
    python tools/pmc_compare.py base cand1 ...      (reads gpurun_out/prof_<tag>)"""
import re
import subprocess
import sys

tags = sys.argv[1:]
out = {}
for v in tags:
    t = subprocess.run([sys.executable, "tools/rocpd_summary.py", "gpurun_out/prof_%s" % v], capture_output=True, text=True).stdout
    d = {}
    for l in t.splitlines():
        if "costvol" in l and "mean=" in l:
            m = re.search(r"\)\s+(\S+)\s+n=\d+\s+mean=(\S+)", l)
            if m:
                d[m.group(1)] = float(m.group(2))
    out[v] = d
keys = sorted(set().union(*[set(d) for d in out.values()]))
print("%-30s" % "counter" + "".join("%14s" % t for t in tags) + "".join("%9s" % ("x" + t) for t in tags[1:]))
for k in keys:
    vals = [out[t].get(k, 0.0) for t in tags]
    print("%-30s" % k + "".join("%14.4g" % v for v in vals) + "".join("%9.3f" % (v / vals[0] if vals[0] else 0) for v in vals[1:]))
