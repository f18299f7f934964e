#!/usr/bin/env python
"""Register / LDS / scratch footprint of the kernels inside a built library (no GPU needed):
   python tools/kernel_regs.py [lib.so] [name filter]
Reads the AMDGPU code objects embedded in .hip_fatbin and prints the metadata notes of every kernel whose demangled
name contains the filter."""
import os, re, subprocess, sys, tempfile
so = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "satmvs_amd/lib/libsatmvs_hip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
LLVM = "/opt/rocm/lib/llvm/bin/"
data = open(so, "rb").read()
starts = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01\x40", data)]
rows = []
for i, s in enumerate(starts):
    e = starts[i + 1] if i + 1 < len(starts) else len(data)
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(data[s:e]); path = f.name
    txt = subprocess.run([LLVM + "llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
    os.unlink(path)
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        rows.append((g("name"), g("vgpr_count"), g("agpr_count") if False else blk.split()[0], g("sgpr_count"), g("group_segment_fixed_size"),
                     g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("%-90s %5s %5s %5s %7s %7s %6s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "vspill", "sspill"))
for r, n in zip(rows, names):
    n = re.sub(r"^void ", "", n).replace("smvs::", "").split("(")[0]
    if flt in n:
        print("%-90s %5s %5s %5s %7s %7s %6s %6s" % ((n[:90],) + r[1:]))
