#!/usr/bin/env python3
"""Developer tool (GPU box): leaf formation inside the 8-wide collapse (scene_build.cpp, LeafCosts) -- tree shape and render speed for a
list of (binary leaf size, leaf visit cost, per-triangle cost) settings.  usage: leaf_formation_probe.py [cornell-box|bathroom:D] [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "cornell-box"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
if name == "cornell-box": scene, depth = pkg.scenes.cornell_box(800, 800), 8
else:
    d = int(name.split(":")[1]); scene, depth = pkg.scenes.bathroom_stress(3840 if d >= 400 else 1920, 2160 if d >= 400 else 1080, detail=d), 16 if d >= 400 else 8
configs = [None, (1, 0.0, 0.3), (1, 0.1, 0.5), (1, 0.3, 0.3), (1, 0.0, 1.0), (1, 0.5, 0.2), (2, 0.1, 0.5), (3, 0.1, 0.5), None]
if len(sys.argv) > 3: configs = [None] + [tuple(float(x) for x in c.split(",")) for c in sys.argv[3:]] + [None]
for cfg in configs:
    for k in ("MCPT_BIN_LEAF", "MCPT_DP_LEAF_VISIT", "MCPT_DP_LEAF_TRI"): os.environ.pop(k, None)
    if cfg: os.environ["MCPT_BIN_LEAF"] = str(int(cfg[0])); os.environ["MCPT_DP_LEAF_VISIT"] = str(cfg[1]); os.environ["MCPT_DP_LEAF_TRI"] = str(cfg[2])
    r = pkg.Renderer(scene, max_depth=depth, flags=pkg.FLAG_COUNT_TRAVERSAL)
    i = r.info()
    r.render(4, seed=1); r.sync(); c0 = r.counters()
    r.close()
    r = pkg.Renderer(scene, max_depth=depth)
    r.render(8, seed=1); r.sync()
    best = None
    for k in range(3):
        r.reset_counters(); r.render(spp, seed=2 + k); r.sync(); c = r.counters(); best = c.kernel_ms if best is None else min(best, c.kernel_ms)
    print("%-18s %s  wide nodes %d depth %d  box/ray %.2f tri/ray %.2f  render %.2f ms  %.0f Mray/s" % (
        "default" if cfg is None else "leaf%d v%.2f t%.2f" % cfg, name, i.wide_nodes, i.wide_depth, c0.box_tests / c0.rays, c0.tri_tests / c0.rays, best, c.rays / best / 1e3), flush=True)
    r.close()
