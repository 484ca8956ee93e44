#!/usr/bin/env python3
"""Developer tool (GPU box): how far from the origin / at what scale does the fp32 traversal still follow the fp64 oracle?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
import oracle as orc
S = pkg.scenes


def transformed(scene, scale=1.0, offset=(0, 0, 0)):
    off = np.asarray(offset, float)
    q = np.vectorize(S._q)
    v = q(scene.vertex * scale + off)
    c = scene.camera
    cam = S._qcam(tuple(np.asarray(c.eye) * scale + off), tuple(np.asarray(c.lookat) * scale + off), c.up, c.fovy, c.width, c.height)
    return S.SceneData(scene.name + "-x", v, scene.normal, scene.texcoord, scene.face, scene.materials, cam, dict(scene.meta))


base = S.cornell_box_small(64, 64)
flags = pkg.FLAG_CORRECT_SHADOW_T2
for scale, off in [(1, (0, 0, 0)), (1, (100, -3, 0.25)), (1, (1000, -3, 0.25)), (1, (5e4, -3, 0.25)), (100, (0, 0, 0)), (0.01, (0, 0, 0))]:
    sc = transformed(base, scale, off)
    r = pkg.Renderer(sc, max_depth=6, flags=flags); r.render(16, seed=3); g = r.read_accum(); r.close()
    cpu, _, _ = orc.Oracle(sc, max_depth=6, flags=flags).render(16, seed=3)
    gm, cm = g[..., :3] / 16, cpu[..., :3] / 16
    tol = 1e-4 * np.maximum(1.0, cm)
    frac = float(np.mean(np.any(np.abs(gm - cm) > tol, axis=-1)))
    print("scale %-6g offset %-18s pixels beyond 1e-4: %6.2f%%  mean gpu %s cpu %s  finite %s" % (scale, off, 100 * frac, gm.mean((0, 1)), cm.mean((0, 1)), bool(np.isfinite(g).all())), flush=True)

# NaN provocation: zero vertex normals on the floor -> normalize(0) = NaN shading normal
nb = S.open_box(32, 32)
nrm = nb.normal.copy()
floor_corners = np.unique(nb.face[:2, :, 1])
nrm[floor_corners] = 0.0
sc = S.SceneData("nan-box", nb.vertex, nrm, nb.texcoord, nb.face, nb.materials, nb.camera, {})
for fl in (flags, 0):
    r = pkg.Renderer(sc, max_depth=4, flags=fl); r.render(16, seed=1); g = r.read_accum(); r.close()
    cpu, _, _ = orc.Oracle(sc, max_depth=4, flags=fl).render(16, seed=1)
    print("NaN scene flags", fl, "gpu finite", bool(np.isfinite(g).all()), "count ok", bool((g[..., 3] == 16).all()), "cpu finite", bool(np.isfinite(cpu).all()),
          "mean gpu", (g[..., :3] / 16).mean((0, 1)), "cpu", (cpu[..., :3] / 16).mean((0, 1)),
          "pixels differing", float(np.mean(np.any(np.abs(g[..., :3] - cpu[..., :3]) > 1e-3 * np.maximum(16.0, np.abs(cpu[..., :3])), axis=-1))))
