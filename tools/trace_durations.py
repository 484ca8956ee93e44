#!/usr/bin/env python3
"""Print per-dispatch durations of the wavefront kernels from a rocprofv3 --kernel-trace CSV directory."""
import csv, glob, sys, os
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sh = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "wf_shade" in r["Kernel_Name"]]
tr = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "wf_trace" in r["Kernel_Name"]]
gaps = []
wf = [r for r in rows if "wf_" in r["Kernel_Name"]]
for a, b in zip(wf, wf[1:]): gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
print("dispatches shade %d trace %d; total shade %.1f ms trace %.1f ms; gaps total %.1f ms (median %.1f us)" % (len(sh), len(tr), sum(sh) / 1e3, sum(tr) / 1e3, sum(g for g in gaps if g < 1000) / 1e3, sorted(gaps)[len(gaps) // 2] if gaps else 0))
print("last %d iterations: shade us / trace us" % n)
for i in range(max(0, len(sh) - n), len(sh)):
    print("%4d  %8.1f  %8.1f" % (i, sh[i], tr[i] if i < len(tr) else -1))
