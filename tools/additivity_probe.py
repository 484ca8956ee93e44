#!/usr/bin/env python3
"""Developer tool (GPU box): render(4)+render(4) against render(8) on S-bath 0.59 M, 1920x1080 (what test_full_size_properties_other_configs asserts)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
scene = pkg.scenes.bathroom_stress(1920, 1080, detail=160)
r = pkg.Renderer(scene, max_depth=8)
r.render(4, seed=3, first_sample=0); r.render(4, seed=3, first_sample=4); ab = r.read_accum()
r.clear(); r.render(8, seed=3); whole = r.read_accum()
r.clear(); r.render(8, seed=3); whole2 = r.read_accum(); r.close()
for name, x, y in (("4+4 vs 8", ab, whole), ("8 vs 8 again", whole, whole2)):
    bad = ~np.isclose(x, y, rtol=1e-4, atol=1e-4)
    px = np.any(bad, axis=-1)
    d = np.abs(x - y)[..., :3].max(-1)
    print(os.environ.get("MCPT_NO_RECENTRE", "-"), name, "pixels beyond tolerance:", int(px.sum()), " max abs diff %.4g" % d.max(), " at", np.unravel_index(d.argmax(), d.shape), x[np.unravel_index(d.argmax(), d.shape)], y[np.unravel_index(d.argmax(), d.shape)])
