#!/usr/bin/env python3
"""Developer tool: time the render kernel of one library build on the bench workload (S-cornell 800x800 depth 8).
usage: MCPT_LIB_PATH=<.so> python tools/perf_probe.py [spp] [scene] [repeat]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
name = sys.argv[2] if len(sys.argv) > 2 else "cornell-box"
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if name == "cornell-box": scene = pkg.scenes.cornell_box(800, 800)
elif name == "veach-mis": scene = pkg.scenes.veach_mis(1280, 720)
elif name.startswith("bathroom"): scene = pkg.scenes.bathroom_stress(1920, 1080, detail=int(name.split(":")[1]) if ":" in name else 64)
else: raise SystemExit("unknown scene")
depth = int(os.environ.get("MCPT_DEPTH", "8"))
r = pkg.Renderer(scene, max_depth=depth, flags=int(os.environ.get("MCPT_FLAGS", "0")), samples_per_item=int(os.environ.get("MCPT_SPI", "0")))
i = r.info()
r.render(8, seed=1); r.sync()
best = None
for k in range(rep):
    r.reset_counters(); r.render(spp, seed=2 + k); r.sync(); c = r.counters()
    ms = c.kernel_ms; best = ms if best is None else min(best, ms)
print("%-40s %s tris=%d nodes=%d depth=%d bvh %.0f ms  spp=%d  best %.2f ms  %.1f Mray/s  %.1f Mpath/s  rays/path %.2f" % (
    os.path.basename(os.environ.get("MCPT_LIB_PATH", "default")), name, i.n_tris, i.n_nodes, i.bvh_depth, i.bvh_build_ms, spp, best,
    c.rays / best / 1e3, c.paths / best / 1e3, c.rays / c.paths), flush=True)
print("   iterations %d launches %d   trace %.3f ms/launch  shade %.3f ms/launch" % (c.iterations, c.launches, c.trace_ms_total / max(1, c.iterations), c.shade_ms_total / max(1, c.iterations)))
if c.box_tests: print("   box/ray %.2f tri/ray %.2f  stack spills/ray %.4f" % (c.box_tests / c.rays, c.tri_tests / c.rays, c.stack_spills / c.rays))
