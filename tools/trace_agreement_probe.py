#!/usr/bin/env python3
"""Developer tool (GPU box): does the production 8-wide trace kernel find what the exact-box cross-check traversal finds?  Two ray populations on
a bench scene: random rays from inside the scene, and second-generation rays that START ON SURFACES (the hit points of the first population,
cosine-ish directions off the surface) -- the population where thin quantised boxes, tmin and the fp16 plane arithmetic meet.
usage: [MCPT_LIB_PATH=...] python tools/trace_agreement_probe.py [c2|c3|c4] [n_rays]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
r = pkg.Renderer(pkg.scenes.SCENES[cfg["scene"]](cfg["res"][0], cfg["res"][1], **cfg["kw"]), max_depth=cfg["depth"], flags=4)      # flags=4: count box / triangle tests
info = r.info()
rng = np.random.default_rng(7)
lo = np.array(info.bbox_lo if hasattr(info, "bbox_lo") else [-1, -1, -1], float); hi = np.array(info.bbox_hi if hasattr(info, "bbox_hi") else [1, 1, 1], float)
def compare(o, d, label, any_hit=False, t2=None):
    ta, tria, ua, va = r.probe_trace(o, d, t2=t2, any_hit=any_hit)
    r.reset_counters()
    tb, trib, ub, vb = r.probe_trace4(o, d, t2=t2, any_hit=any_hit)
    c = r.counters()
    label = "%s [box %.2f tri %.2f per ray]" % (label, c.box_tests / len(ta), c.tri_tests / len(ta))
    if any_hit:
        hit_a, hit_b = tria != 0, trib != 0                           # any-hit probes return 1 / 0
        print("%-64s n=%d  occluded: cross-check %.4f production %.4f   differ %d (%.5f %%)  [production misses an occluder: %d]" % (
            label, len(ta), hit_a.mean(), hit_b.mean(), int((hit_a != hit_b).sum()), 100.0 * (hit_a != hit_b).mean(), int((hit_a & ~hit_b).sum())), flush=True)
        return ta, tria
    same_tri = tria == trib
    same_t = np.abs(ta - tb) <= 1e-5 * np.maximum(1.0, np.abs(ta))
    hit_a, hit_b = tria >= 0, trib >= 0
    lost = hit_a & (~hit_b | (tb > ta * (1 + 1e-4) + 1e-5))         # production missed, or found something strictly farther
    print("%-64s n=%d  hit rate %.4f / %.4f   different triangle %d (%.5f %%)   of those with a different distance %d   production farther or missing: %d" % (
        label, len(ta), hit_a.mean(), hit_b.mean(), int((~same_tri).sum()), 100.0 * (~same_tri).mean(), int((~same_tri & ~same_t).sum()), int(lost.sum())), flush=True)
    return ta, tria
# population 1: camera-like rays from around the eye
eye = np.array(cfg.get("eye", [0, 0, 0]), float) if isinstance(cfg.get("eye", None), (list, tuple)) else None
xy = np.stack([rng.integers(0, cfg["res"][0], n), rng.integers(0, cfg["res"][1], n)], 1).astype(np.int32)
cr = r.probe_cast_ray(xy, rng.random((n, 2)).astype(np.float32))
o1, d1 = cr[:, 0:3].astype(np.float64), cr[:, 3:6].astype(np.float64)
t1, tri1 = compare(o1, d1, "camera rays")
# population 2..4: rays starting on the surfaces the previous population hit, random directions (half of them leave the surface at grazing angles)
o, d, t, tri = o1, d1, t1, tri1
for gen in range(3):
    ok = tri >= 0
    p = (o[ok] + d[ok] * t[ok, None].astype(np.float64)).astype(np.float32).astype(np.float64)      # fp32 hit points, like the pool's ray origins
    nd = rng.normal(size=p.shape); nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    graze = rng.random(len(p)) < 0.3
    nd[graze] = nd[graze] * np.array([1.0, 1.0, 1.0]) ; k = rng.integers(0, 3, len(p)); nd[graze, k[graze]] *= 1e-3; nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    nd = nd.astype(np.float32).astype(np.float64)
    t, tri = compare(p, nd, "surface rays, generation %d" % (gen + 1))
    L = np.where(rng.random(len(p)) < 0.5, rng.random(len(p)) * 2.0 + 0.01, np.where(tri >= 0, t * (1.0 - 1e-4), 1e30))   # half random lengths, half ending just before the closest hit (what a light sample does)
    compare(p, nd, "  as shadow rays (t2 random)", any_hit=True, t2=L)
    o, d = p, nd
