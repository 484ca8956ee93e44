#!/usr/bin/env python3
"""Summarise the counter_collection CSVs of tools/pmc_passes.sh: counters summed over all dispatches, per kernel."""
import csv, glob, os, sys, collections
out = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "render_" in k: k = "render_mega"
        elif "wf_trace" in k: k = "wf_trace"
        elif "wf_shade" in k: k = "wf_shade"
        else: continue
        vals[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k].add((f, r["Dispatch_Id"]))
for k in vals:
    v = vals[k]; g = v.get
    print("==== %s  (dispatches in a pass: ~%d)" % (k, len(ndisp[k]) // max(1, len(set(f for f, _ in ndisp[k])))))
    for c in sorted(v): print("  %-32s %.6g" % (c, v[c]))
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        print("  --- derived")
        print("  wait_any / wave_cycles        %.3f" % (g("SQ_WAIT_ANY", 0) / wc))
        print("  wait_inst_any / wave_cycles   %.3f" % (g("SQ_WAIT_INST_ANY", 0) / wc))
        print("  active_inst_any / wave_cycles %.3f" % (g("SQ_ACTIVE_INST_ANY", 0) / wc))
        if g("SQ_ACTIVE_INST_VALU"): print("  VALU lane utilisation         %.3f" % (g("SQ_THREAD_CYCLES_VALU", 0) / (g("SQ_ACTIVE_INST_VALU") * 64)))
        if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"): print("  VALU busy (active_inst_valu*4 / (busy_cycles/ (8 XCD..)))  see raw")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
        print("  L2 hit rate                   %.4f" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
        print("  L1 miss ratio (TCC reads / TCP accesses) %.4f" % (g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")))
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"): print("  VALU insts per wave           %.1f" % (g("SQ_INSTS_VALU") / g("SQ_WAVES")))
