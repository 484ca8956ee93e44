#!/usr/bin/env python3
"""Developer tool (GPU box): many samples of a SMALL film -- how many pool slots may work on the same pixel at once before the film's float
atomics hurt?  usage: MCPT_WF_SLOTS_PER_PIXEL=<n> python tools/small_film_probe.py [size=64] [spp=4096]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
r = pkg.Renderer(pkg.scenes.cornell_box(size, size), max_depth=8)
r.render(64, seed=1); r.sync()
best = 1e9
for k in range(3):
    r.reset_counters(); r.render(spp, seed=2 + k); r.sync(); c = r.counters(); best = min(best, c.kernel_ms)
print("slots/pixel limit %s: %dx%d x %d spp  %.2f ms  %.0f Mray/s  iterations %d" % (os.environ.get("MCPT_WF_SLOTS_PER_PIXEL", "32"), size, size, spp, best, c.rays / best / 1e3, c.iterations))
