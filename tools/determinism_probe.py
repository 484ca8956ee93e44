#!/usr/bin/env python3
"""Developer tool (GPU box): is a deterministic-mode render of a scene bit-reproducible run to run?  usage: determinism_probe.py [detail] [spp]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
detail = int(sys.argv[1]) if len(sys.argv) > 1 else 160
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
scene = pkg.scenes.bathroom_stress(960, 540, detail=detail)
imgs = []
for k in range(3):
    r = pkg.Renderer(scene, max_depth=8, flags=pkg.FLAG_DETERMINISTIC | pkg.FLAG_COUNT_TRAVERSAL)
    r.render(spp, seed=3); imgs.append(r.read_accum()); c = r.counters(); r.close()
    print("run", k, "spills/ray %.5f" % (c.stack_spills / c.rays), "box/ray %.2f" % (c.box_tests / c.rays))
for k in (1, 2):
    d = np.any(imgs[0] != imgs[k], axis=-1)
    rel = np.abs(imgs[0][..., :3] - imgs[k][..., :3]).max(-1) / np.maximum(1e-3, np.abs(imgs[0][..., :3]).max(-1))
    print("pend", os.environ.get("MCPT_WF_PEND"), "run 0 vs", k, "pixels differing: %d of %d" % (d.sum(), d.size), " rel > 1e-3: %d  max rel %.3g" % ((rel > 1e-3).sum(), rel.max()))
