#!/usr/bin/env python3
"""Developer tool: configs[4] shape -- S-bath >= 4 M triangles, 3840x2160, depth 16 -- at a few spp."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
t0 = time.time(); scene = pkg.scenes.bathroom_stress(3840, 2160, detail=420); print("scene gen %.1f s, %d tris" % (time.time() - t0, scene.n_faces), flush=True)
t0 = time.time(); r = pkg.Renderer(scene, max_depth=16, flags=int(os.environ.get("MCPT_FLAGS", "0"))); i = r.info()
print("create %.1f s: nodes %d depth %d bvh %.0f ms, device %.2f GB" % (time.time() - t0, i.n_nodes, i.bvh_depth, i.bvh_build_ms, i.device_bytes / 1e9), flush=True)
r.render(2, seed=1); r.sync()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
    r.reset_counters(); r.clear()
    r.render(spp, seed=2); r.sync(); c = r.counters(); a = r.read_accum()
    print("spp %d: %.1f ms  %.1f Mray/s  %.1f Mpath/s  rays/path %.2f  self-shadow %.3f  iterations %d  mean %s finite %s" % (
        spp, c.kernel_ms, c.rays / c.kernel_ms / 1e3, c.paths / c.kernel_ms / 1e3, c.rays / c.paths, c.self_shadow_hits / max(1, c.self_shadow_tests), c.iterations,
        (a[..., :3] / a[..., 3:]).mean((0, 1)), bool(np.isfinite(a).all())), flush=True)
