#!/usr/bin/env python3
"""Developer tool: copy the evidence bundle tools/final_profile.sh left under gpurun_out/<tag>/ into profiles/ (tracked).
usage: python tools/update_profiles.py <tag> <prefix>     e.g.  r01b r01_final"""
import json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
src = os.path.join(root, "gpurun_out", tag); dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, prefix + "_bench.json"))
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, prefix + "_kernel_stats.csv"))
summ = open(os.path.join(src, "pmc", "summary.txt")).read()
vals = {}; k = None; hbm = []; sq = []
for line in summ.splitlines():
    m = re.match(r"==== (\S+)", line)
    if m: k = m.group(1); vals[k] = {}
    m2 = re.match(r"\s+(\S+)\s+([\d.e+]+)$", line)
    if m2 and k: vals[k][m2.group(1)] = float(m2.group(2))
    (hbm if ("FETCH_SIZE" in line or "WRITE_SIZE" in line or line.startswith("====")) else sq).append(line)
    if line.startswith("===="): sq.append(line)
open(os.path.join(dst, prefix + "_pmc_hbm.txt"), "w").write("\n".join(hbm) + "\n")
open(os.path.join(dst, prefix + "_pmc_sq_tcc.txt"), "w").write("\n".join(l for l in sq if "FETCH_SIZE" not in l and "WRITE_SIZE" not in l) + "\n")
n = 880   # one 8-spp warm-up + one 1024-spp render of tools/perf_probe.py: launches of each kernel (see the summary's "dispatches" line)
m = re.search(r"wf_trace\s+\(dispatches in a pass: ~(\d+)\)", summ)
if m: n = int(m.group(1))
def per_launch(kern, corrected=True):
    v = vals[kern]; return int((v["FETCH_SIZE"] * (2 if corrected else 1) + v["WRITE_SIZE"]) * 1024 / n)
j = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/final_profile.sh %s -> tools/pmc_passes.sh ... 1024 '' '4 5'), S-cornell 800x800 depth 8, "
               "one 8-spp warm-up + one 1024-spp render = %d launches of each kernel; raw sums in profiles/%s_pmc_hbm.txt" % (tag, n, prefix),
     "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) streaming reads -> fetch doubled; WRITE_SIZE taken as is",
     "launches": n,
     "wf_trace_kernel_hbm_bytes_per_launch": per_launch("wf_trace"),
     "wf_shade_kernel_hbm_bytes_per_launch": per_launch("wf_shade"),
     "wf_trace_kernel_uncorrected_bytes_per_launch": per_launch("wf_trace", False)}
# per traced ray: the PMC run renders 8 + 1024 spp of the bench workload; rays per path from the bench line of the same bundle
bj = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
rays = (8 + 1024) * 800 * 800 * bj["rays_per_path"]
j["rays_in_pmc_run"] = int(rays)
j["wf_trace_kernel_hbm_bytes_per_ray"] = round((vals["wf_trace"]["FETCH_SIZE"] * 2 + vals["wf_trace"]["WRITE_SIZE"]) * 1024 / rays, 2)
j["wf_shade_kernel_hbm_bytes_per_ray"] = round((vals["wf_shade"]["FETCH_SIZE"] * 2 + vals["wf_shade"]["WRITE_SIZE"]) * 1024 / rays, 2)
json.dump(j, open(os.path.join(dst, "r01_traffic.json"), "w"), indent=1)
print(json.dumps(j, indent=1))
