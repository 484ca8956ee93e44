#!/usr/bin/env python3
"""Developer tool: where a shade wave's time goes (needs a library built with -DWF_SHADE_STATS; counters are re-purposed).
usage: MCPT_LIB_PATH=build/libmcpt_hip_shstats.so python tools/shade_stats.py [spp] [cornell-box | bathroom:<detail>] [depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
os.environ["MCPT_TIME_KERNELS"] = "1"
name = sys.argv[2] if len(sys.argv) > 2 else "cornell-box"
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
scene = pkg.scenes.cornell_box(800, 800) if name == "cornell-box" else pkg.scenes.bathroom_stress(1920, 1080, detail=int(name.split(":")[1]))
r = pkg.Renderer(scene, max_depth=depth)
r.render(8, seed=1); r.sync(); r.reset_counters()
r.render(spp, seed=2); r.sync(); c = r.counters()
names = ["tables + slot records + class sort", "phase 1: hit record gather, emitter MIS", "phase 2: fp64 hit point + light sample",
         "phase 3: BSDF, NEE, BSDF sample", "film write, item pull, camera ray", "shadow queue + coalesced stores", "counters"]
t = [c.debug[0], c.debug[1], c.debug[2], c.debug[3], c.box_tests, c.tri_tests, c.stack_spills]
waves = max(1, c.texel_fetches)
tot = float(sum(t)) or 1.0
print("shade launches %d, mean launch %.3f ms; waves %d; mean wave lifetime %.0f shader-clock ticks" % (c.iterations, c.shade_ms_total / max(1, c.iterations), waves, tot / waves))
for n, v in zip(names, t):
    print("  %-44s %5.1f %%   %8.0f ticks per wave" % (n, 100.0 * v / tot, v / waves))
