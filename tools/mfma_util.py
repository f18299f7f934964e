#!/usr/bin/env python
"""MFMA utilisation of the convolution kernels from a tools/profile_r03.sh summary (kernel trace + counter passes of
tools/run_native_models.py):   python tools/mfma_util.py gpurun_out/prof_r03/summary.txt [trace block] [pmc block]

  flops      = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512      (one v_mfma_f32_32x32x2_f32 = 4096 flop = 8 MOPS; the counter pair
                                                        MFMA_BUSY_CYCLES / MOPS = 8.0 cycles confirms the unit: 64 cycles per instruction)
  TFLOP/s    = flops / mean kernel duration (kernel trace of the same command)
  of peak    = TFLOP/s / 157.3 (dense fp32 MFMA peak of MI355X, MI355X_MICROARCH.md)
  pipe busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * SQ_BUSY_CYCLES / 32 shader engines)
"""
import re, sys

txt = open(sys.argv[1]).read()
TRACE = sys.argv[2] if len(sys.argv) > 2 else "trace_models"          # block names of tools/rocpd_summary.py: kernel trace ...
PMC = sys.argv[3] if len(sys.argv) > 3 else "pmcm_SQ_INSTS_VALU_MFMA"    # ... and the MFMA counter pass
blocks = re.split(r"^== ", txt, flags=re.M)
dur = {}
cnt = {}
for b in blocks:
    head = b.split("\n", 1)[0]
    if head.startswith("kernel trace: " + TRACE):
        for l in b.splitlines()[2:]:
            m = re.match(r"(.{72}) +(\d+) +([\d.]+) +([\d.]+)", l)
            if m:
                dur[m.group(1).strip()[:60]] = (int(m.group(2)), float(m.group(4)))
    if head.startswith("pmc pass: " + PMC):
        for l in b.splitlines()[1:]:
            m = re.match(r" +(.{60}) (\S+) +n=(\d+) +mean=(\S+)", l)
            if m:
                cnt.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(4))
print("%-62s %7s %9s %10s %9s %8s %9s" % ("kernel", "calls", "avg us", "GFLOP/call", "TFLOP/s", "of peak", "pipe busy"))
for k, c in sorted(cnt.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0)):
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0)
    if mops <= 0:
        continue
    d = [v for n, v in dur.items() if n.startswith(k[:50]) or k.startswith(n[:50])]
    if not d:
        continue
    calls, avg = d[0]
    fl = mops * 512
    tf = fl / (avg * 1e-6) / 1e12
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["SQ_BUSY_CYCLES"] / 32.0)
    print("%-62s %7d %9.1f %10.3f %9.1f %7.1f%% %8.1f%%   (busy cycles / MOPS = %.2f)" % (k, calls, avg, fl / 1e9, tf, 100 * tf / 157.3, 100 * busy,
                                                                                 c["SQ_VALU_MFMA_BUSY_CYCLES"] / mops))
