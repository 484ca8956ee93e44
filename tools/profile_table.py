#!/usr/bin/env python3
"""Developer tool: print the numbers of DESIGN.md's per-configuration table from profiles/r03_* (after tools/update_profiles_r03.py)."""
import json, csv, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = json.load(open(os.path.join(root, "profiles", "r03_traffic.json")))
for c in ("c2", "c3", "c4", "c5"):
    d = json.load(open(os.path.join(root, "profiles", "r03_%s_bench.json" % c))); r = d["roofline"]; t = T[c]
    ks = [row for row in csv.reader(open(os.path.join(root, "profiles", "r03_%s_kernel_stats.csv" % c))) if row and "wf_trace8_kernel<false>" in row[0]]
    tot = (t["wf_trace_kernel_hbm_bytes_per_ray"] + t["wf_shade_kernel_hbm_bytes_per_ray"]) * d["value"] * 1e6 / 1e12
    half = r["launches_per_step"] / 2
    print(c, "Mray/s %.0f  ms/step %.1f  Mpath/s %.0f | trace ms events %.3f rocprof %.3f  shade %.3f | frac %.2f | B/ray %.0f + %.0f -> %.2f TB/s = %.2f | valu %.2f lane %.2f / %.2f | salu %.2f wait %.2f l2 %.2f | box %.1f tri %.2f | per stream %.1f x %.3f = %.1f" % (
        d["value"], d["ms_per_step"], d.get("paths_per_s", d.get("mpath_per_s", 0)), r["kernel_ms"], float(ks[0][3]) / 1e6, r["second_kernel"]["kernel_ms"], r["frac"],
        t["wf_trace_kernel_hbm_bytes_per_ray"], t["wf_shade_kernel_hbm_bytes_per_ray"], tot, tot / 8, t["valu_issue_frac"], t["wf_trace_kernel_valu_lane_utilisation"],
        t["wf_shade_kernel_valu_lane_utilisation"], t["wf_trace_kernel_salu_to_valu"], t["wf_trace_kernel_wait_any"], t["wf_trace_kernel_l2_hit"],
        r.get("box_tests_per_ray", 0), r.get("tri_tests_per_ray", 0), half, r["kernel_ms"] + r["second_kernel"]["kernel_ms"], half * (r["kernel_ms"] + r["second_kernel"]["kernel_ms"])))
    print("   ", t["valu_issue_note"][:110])
