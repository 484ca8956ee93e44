#!/usr/bin/env python3
"""Developer tool (round 4): copy the bundle tools/r04_profile.sh left under gpurun_out/r04_prof_<cfg>/ into profiles/ (tracked) and merge its
HBM traffic / VALU / lane utilisation figures into profiles/r04_traffic.json[<cfg>] (what bench.py reads for roofline.traffic, hbm_measured, bound).
usage: python tools/update_profiles_r04.py <cfg> [inner_steps_per_ray from tools/sched_stats.py]"""
import json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1]
src = os.path.join(root, "gpurun_out", "r04_prof_" + cfg); dst = os.path.join(root, "profiles"); pre = "r04_%s_" % cfg
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, pre + "bench.json"))
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, pre + "kernel_stats.csv"))
summ = open(os.path.join(src, "pmc", "summary.txt")).read()
vals = {}; k = None; hbm = []; sq = []
for line in summ.splitlines():
    m = re.match(r"==== (\S+)", line)
    if m: k = m.group(1); vals[k] = {}
    m2 = re.match(r"\s+(\S+)\s+([\d.e+]+)$", line)
    if m2 and k: vals[k][m2.group(1)] = float(m2.group(2))
    m3 = re.match(r"\s+VALU lane utilisation\s+([\d.]+)", line)
    if m3 and k: vals[k]["lane_util"] = float(m3.group(1))
    (hbm if ("FETCH_SIZE" in line or "WRITE_SIZE" in line or line.startswith("====")) else sq).append(line)
    if line.startswith("===="): sq.append(line)
w = json.load(open(os.path.join(src, "pmc", "workload4.json")))
open(os.path.join(dst, pre + "pmc_hbm.txt"), "w").write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (KB), separate passes, workload: %s\n" % json.dumps(w) + "\n".join(hbm) + "\n")
open(os.path.join(dst, pre + "pmc_sq_tcc.txt"), "w").write("# rocprofv3 --pmc SQ_* / TCC_* passes, workload: %s\n" % json.dumps(w) + "\n".join(l for l in sq if "FETCH_SIZE" not in l and "WRITE_SIZE" not in l) + "\n")
def bytes_of(kern, corrected=True): v = vals[kern]; return (v["FETCH_SIZE"] * (2 if corrected else 1) + v["WRITE_SIZE"]) * 1024
entry = {
    "source": "tools/r04_profile.sh %s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate counter-only passes over ONE render of %d spp (tools/pmc_workload.py): "
              "%d rays, %d launches of each kernel; raw sums in profiles/%spmc_hbm.txt" % (cfg, w["spp"], w["rays"], w["iterations"], pre),
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) reads -> fetch doubled; WRITE_SIZE as is",
    "fetch_correction_note": "x2 on ALL fetch: calibrated in round 4 for both access patterns of these kernels (profiles/r04_fetch_calibration.txt) -- a 1-GiB float4 stream reads 0.500 of its bytes "
                             "on the counter, and 52 M random 80-B record gathers beyond the Infinity Cache 0.495 of the 128-B LINES they touch (the L2 fetches whole lines for a 16-B "
                             "per-lane gather); FETCH_SIZE counts Infinity-Cache hits too, so for traversal data that fits the 256-MB MALL the figure is memory-side traffic, not all of it HBM",
    "rays_in_pmc_run": w["rays"], "launches_in_pmc_run": w["iterations"],
    "wf_trace_kernel_hbm_bytes_per_ray": round(bytes_of("wf_trace") / w["rays"], 2),
    "wf_shade_kernel_hbm_bytes_per_ray": round(bytes_of("wf_shade") / w["rays"], 2),
    "wf_trace_kernel_hbm_bytes_per_launch": int(bytes_of("wf_trace") / w["iterations"]),
    "wf_shade_kernel_hbm_bytes_per_launch": int(bytes_of("wf_shade") / w["iterations"]),
    "wf_trace_kernel_uncorrected_bytes_per_ray": round(bytes_of("wf_trace", False) / w["rays"], 2),
    "wf_trace_kernel_valu_lane_utilisation": vals["wf_trace"].get("lane_util"),
    "wf_shade_kernel_valu_lane_utilisation": vals["wf_shade"].get("lane_util"),
    "wf_trace_kernel_salu_to_valu": round(vals["wf_trace"]["SQ_INSTS_SALU"] / vals["wf_trace"]["SQ_INSTS_VALU"], 3),
    "wf_trace_kernel_wait_any": round(vals["wf_trace"]["SQ_WAIT_ANY"] / vals["wf_trace"]["SQ_WAVE_CYCLES"], 3),
    "wf_shade_kernel_wait_any": round(vals["wf_shade"]["SQ_WAIT_ANY"] / vals["wf_shade"]["SQ_WAVE_CYCLES"], 3),
    "wf_trace_kernel_l2_hit": round(vals["wf_trace"]["TCC_HIT_sum"] / (vals["wf_trace"]["TCC_HIT_sum"] + vals["wf_trace"]["TCC_MISS_sum"]), 3),
    "wf_shade_kernel_l2_hit": round(vals["wf_shade"]["TCC_HIT_sum"] / (vals["wf_shade"]["TCC_HIT_sum"] + vals["wf_shade"]["TCC_MISS_sum"]), 3),
}
# How busy the SIMDs' vector ALUs are: SQ_ACTIVE_INST_VALU (quad-cycles a wave spends executing VALU instructions -- the SQ's own accounting of what each
# instruction costs: 4.0 cycles on average in the trace kernel, consistent with tools/uarch_probe2's two price classes of ~2.5 and ~4 cycles at 2.4 GHz)
# of both kernels x 4 / (1024 SIMDs x 2.4 GHz x the UNPROFILED time this many rays take at the bench's rate)
bench = json.load(open(os.path.join(src, "bench.json")))
t_s = w["rays"] / (bench["value"] * 1e6)
valu = vals["wf_trace"]["SQ_INSTS_VALU"] + vals["wf_shade"]["SQ_INSTS_VALU"]
act = vals["wf_trace"]["SQ_ACTIVE_INST_VALU"] + vals["wf_shade"]["SQ_ACTIVE_INST_VALU"]
entry["valu_busy_frac"] = round(4.0 * act / (t_s * 2.4e9 * 1024), 3)
entry["valu_cycles_per_instruction"] = {"wf_trace": round(4.0 * vals["wf_trace"]["SQ_ACTIVE_INST_VALU"] / vals["wf_trace"]["SQ_INSTS_VALU"], 2),
                                        "wf_shade": round(4.0 * vals["wf_shade"]["SQ_ACTIVE_INST_VALU"] / vals["wf_shade"]["SQ_INSTS_VALU"], 2)}
entry["valu_issue_note"] = ("valu_busy_frac = SQ_ACTIVE_INST_VALU of both kernels (%.3g quad-cycles; %.3g wave-instructions, trace %.0f %%) x 4 / (1024 SIMDs x 2.4 GHz x %.1f ms, the time "
                            "the PMC run's %d rays take at this bundle's bench rate of %.0f Mray/s).  The price per instruction is the hardware's own (%.2f cycles on average in the trace "
                            "kernel); round 3 assumed 4 for all, the guide's 2 holds for v_mov / v_add / v_fma_f32 only (profiles/r04_uarch_probe.txt).  Regime measured directly: +16 v_nop "
                            "per inner step (+9 %% of its VALU instructions) = +3.9 %% step time, +48 = +9.7 %% (DESIGN 5.0)"
                            % (act, valu, 100 * vals["wf_trace"]["SQ_INSTS_VALU"] / valu, t_s * 1e3, w["rays"], bench["value"], 4.0 * vals["wf_trace"]["SQ_ACTIVE_INST_VALU"] / vals["wf_trace"]["SQ_INSTS_VALU"]))
if len(sys.argv) > 2: entry["wf_trace_kernel_inner_steps_per_ray"] = float(sys.argv[2])
tp = os.path.join(dst, "r04_traffic.json")
allj = json.load(open(tp)) if os.path.exists(tp) else {}
allj[cfg] = entry
json.dump(allj, open(tp, "w"), indent=1)
print(json.dumps(entry, indent=1))
