#!/usr/bin/env python3
"""Developer tool (GPU box): build time and render speed of the three tree builders (host binned SAH, device LBVH, device PLOC)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "bathroom:160"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if name == "cornell-box": scene, depth = pkg.scenes.cornell_box(800, 800), 8
else:
    d = int(name.split(":")[1]); scene, depth = pkg.scenes.bathroom_stress(3840 if d >= 400 else 1920, 2160 if d >= 400 else 1080, detail=d), 16 if d >= 400 else 8
for kind in ("host", "lbvh", "ploc"):
    if kind != "host": os.environ["MCPT_GPU_BVH"] = kind
    t0 = time.time()
    r = pkg.Renderer(scene, max_depth=depth, flags=(pkg.FLAG_GPU_BVH_BUILD if kind != "host" else 0) | pkg.FLAG_COUNT_TRAVERSAL)
    t_create = time.time() - t0
    i = r.info()
    r.render(4, seed=1); r.sync(); c0 = r.counters()
    r.close()
    r = pkg.Renderer(scene, max_depth=depth, flags=(pkg.FLAG_GPU_BVH_BUILD if kind != "host" else 0))
    r.render(8, seed=1); r.sync()
    best = None
    for k in range(3):
        r.reset_counters(); r.render(spp, seed=2 + k); r.sync(); c = r.counters(); best = c.kernel_ms if best is None else min(best, c.kernel_ms)
    print("%-5s %s tris=%d  bvh build %.0f ms (create %.2f s)  depth %d  box/ray %.1f tri/ray %.2f  render %.1f ms  %.0f Mray/s" % (
        kind, name, i.n_tris, i.bvh_build_ms, t_create, i.bvh_depth, c0.box_tests / c0.rays, c0.tri_tests / c0.rays, best, c.rays / best / 1e3), flush=True)
    r.close()
