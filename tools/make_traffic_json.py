#!/usr/bin/env python
"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_r02.sh (rocpd sqlite files still present):
    python tools/make_traffic_json.py <prof dir> > profiles/pmc_traffic.json
HBM bytes per launch of the dominant kernel = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B
request for coalesced streams, MI355X_MICROARCH.md HBM section); stamped with the hash of the kernel sources so that bench.py
only reports it for the code it was measured on."""
import glob, json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

root = sys.argv[1]
val = {}
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob(os.path.join(root, "pmc_%s" % cn, "**", "*.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
                       "group by kernel_name order by avg(value) desc", (cn,)).fetchall()
    name, n, v = [r for r in rows if "costvol_dma_kernel" in r[0]][0]
    val[cn] = (name, n, v)
wl = bench.DEFAULT_WORKLOAD if hasattr(bench, "DEFAULT_WORKLOAD") else "cfg2_rpc_3view_768x384x64_c32"
fetch_kb, write_kb = val["FETCH_SIZE"][2], val["WRITE_SIZE"][2]
out = {
    "_comment": "HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate --pmc runs, "
                "never combined with tracing; tools/profile_r02.sh + tools/make_traffic_json.py): WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 "
                "(gfx950 FETCH_SIZE counts 64 B per 128-B request for coalesced streams, MI355X_MICROARCH.md HBM section).  bench.py "
                "reports the figure only while source_sha256 matches the kernel sources it is run with (bench.kernel_source_hash()).",
    wl: {
        "bytes": int(round(write_kb * 1024 + 2 * fetch_kb * 1024)),
        "write_bytes": int(round(write_kb * 1024)),
        "fetch_size_kb": round(fetch_kb, 1),
        "write_size_kb": round(write_kb, 1),
        "launches_sampled": [val["FETCH_SIZE"][1], val["WRITE_SIZE"][1]],
        "source_sha256": bench.kernel_source_hash(),
        "kernel": val["WRITE_SIZE"][0][:60],
    },
}
print(json.dumps(out, indent=2))
