#!/usr/bin/env python3
"""Developer tool: per-kernel time split of one wavefront render (needs MCPT_TIME_KERNELS=1). usage: kernel_split.py [spp] [scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MCPT_TIME_KERNELS", "1")
import __graft_entry__ as ge
pkg = ge.load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
name = sys.argv[2] if len(sys.argv) > 2 else "cornell-box"
if name == "cornell-box": scene = pkg.scenes.cornell_box(800, 800)
elif name == "veach-mis": scene = pkg.scenes.veach_mis(1280, 720)
else: scene = pkg.scenes.bathroom_stress(1920, 1080, detail=int(name.split(":")[1]) if ":" in name else 64)
r = pkg.Renderer(scene, max_depth=int(os.environ.get("MCPT_DEPTH", "8")), flags=int(os.environ.get("MCPT_FLAGS", "0")), samples_per_item=int(os.environ.get("MCPT_SPI", "0")))
r.render(8, seed=1); r.sync(); r.reset_counters(); r.render(spp, seed=2); r.sync(); c = r.counters()
print("%-28s %s spp=%d  total %.2f ms  trace %.2f  shade %.2f  iterations %d  %.1f Mray/s" % (
    os.path.basename(os.environ.get("MCPT_LIB_PATH", "default")), name, spp, c.kernel_ms, c.trace_ms_total, c.shade_ms_total, c.iterations, c.rays / c.kernel_ms / 1e3), flush=True)
