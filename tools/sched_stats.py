#!/usr/bin/env python3
"""Developer tool: scheduler statistics of the trace kernel (needs a library built with -DWF_SCHED_STATS; counters are re-purposed).
usage: MCPT_LIB_PATH=build/libmcpt_hip_stats.so python tools/sched_stats.py [spp] [c2|c3|c4|c5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
os.environ["MCPT_TIME_KERNELS"] = "1"
cfg_name = sys.argv[2] if len(sys.argv) > 2 else "c2"                   # a BASELINE.json configuration of bench.py
import bench
cfg = bench.CONFIGS[cfg_name]
r = pkg.Renderer(pkg.scenes.SCENES[cfg["scene"]](cfg["res"][0], cfg["res"][1], **cfg["kw"]), max_depth=cfg["depth"], flags=4)
r.render(spp, seed=1); r.sync(); c = r.counters()
rays = c.rays
print(cfg_name, "rays %.3g  inner: execs/ray %.4f lanes/exec %.1f (steps/ray %.2f) | leaf: execs/ray %.4f lanes/exec %.1f (visits/ray %.2f) | refill execs/ray %.4f free slots/exec %.1f" % (
    rays, c.shaded_hits / rays, c.box_tests / max(1, c.shaded_hits), c.box_tests / rays, c.texel_fetches / rays, c.tri_tests / max(1, c.texel_fetches),
    c.tri_tests / rays, c.self_shadow_tests / rays, c.self_shadow_hits / max(1, c.self_shadow_tests)), flush=True)
i = r.info()
grid = int(os.environ.get("MCPT_WF_GRID", "256"))
waves = grid * 16
print("trace launches %d, mean launch %.3f ms; mean wave lifetime %.3f ms = %.1f %% of the launch (grid %d blocks assumed)" % (
    c.iterations, c.trace_ms_total / c.iterations, c.paths * 1e-5 / (c.iterations * waves), 100.0 * c.paths * 1e-5 / (waves * c.trace_ms_total), grid), flush=True)
tot = float(sum(c.debug[:3])) or 1.0
print("wave time by scheduler block (shader cycles, s_memtime): inner %.1f %%  leaf %.1f %%  refill %.1f %%   | cycles per execution: inner %.0f  leaf %.0f  refill %.0f" % (
    100 * c.debug[0] / tot, 100 * c.debug[1] / tot, 100 * c.debug[2] / tot, c.debug[0] / max(1, c.shaded_hits), c.debug[1] / max(1, c.texel_fetches),
    c.debug[2] / max(1, c.self_shadow_tests)), flush=True)
