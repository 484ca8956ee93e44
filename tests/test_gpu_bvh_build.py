"""-m gpu: the device BVH builder (csrc/bvh_gpu.hip, MCPT_FLAG_GPU_BVH_BUILD) behind the same C ABI.

The traversal result does not depend on the tree (closest hit = min t, any hit = exists), so a device-built tree must
  (a) pass the host-side soundness walk of the quantised 8-wide tree (MCPT_VALIDATE_BVH=1: every triangle referenced once and
      inside every box on its root path),
  (b) return the reference's own hits on the reference's own random rays (tests/golden/ref_paths.npz), through both the binary
      tree (probe kernels) and the 8-wide tree (render),
  (c) render the image the host-built tree renders, sample for sample (deterministic mode), up to exact-tie pixels."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def paths():
    with np.load(os.path.join(G, "ref_paths.npz")) as z:
        return {k: z[k] for k in z.files}


def _gpu_tree_renderer(pkg, scene, **kw):
    os.environ["MCPT_VALIDATE_BVH"] = "1"
    try:
        return pkg.Renderer(scene, flags=kw.pop("flags", 0) | pkg.FLAG_GPU_BVH_BUILD, **kw)
    finally:
        os.environ.pop("MCPT_VALIDATE_BVH", None)


def test_device_built_tree_returns_the_reference_hits(pkg, paths):
    p = paths
    r = _gpu_tree_renderer(pkg, pkg.scenes.cornell_box_small(64, 64))
    i = r.info()
    assert i.n_tris == pkg.scenes.cornell_box_small(64, 64).n_faces and 1 <= i.bvh_depth <= 63 and i.max_leaf <= 2
    t, tri, u, v = r.probe_trace(p["cs_ray_o"], p["cs_ray_d"])
    anyh = r.probe_trace(p["cs_ray_o"], p["cs_ray_d"], t2=p["cs_ray_t2"], any_hit=True)[1]
    r.close()
    ref_tri = p["cs_ray_rec"][:, 11].astype(np.int32); ref_hit = p["cs_ray_hit"] == 1
    same = (tri == np.where(ref_hit, ref_tri, -1))
    assert same.mean() >= 0.999, same.mean()
    ok = same & ref_hit
    assert np.allclose(t[ok], p["cs_ray_rec"][ok, 0], rtol=2e-5, atol=2e-6)
    assert (anyh == p["cs_ray_any"]).mean() >= 0.999


@pytest.mark.parametrize("name,kw,res,depth", [("cornell-box-small", {}, (48, 48), 5), ("veach-mis", {"light_lon": 12, "light_lat": 6, "plate_cells": 4}, (64, 36), 0),
                                               ("bathroom2", {"detail": 24, "tex_size": 32}, (64, 36), 6)])
def test_device_and_host_trees_render_the_same_samples(pkg, name, kw, res, depth):
    scene = pkg.scenes.SCENES[name](*res, **kw)
    imgs = []
    for gpu_tree in (False, True):
        if gpu_tree: r = _gpu_tree_renderer(pkg, scene, max_depth=depth, flags=pkg.FLAG_DETERMINISTIC)
        else: r = pkg.Renderer(scene, max_depth=depth, flags=pkg.FLAG_DETERMINISTIC)
        r.render(16, seed=21); imgs.append(r.read_accum()); r.close()
    a, b = imgs
    assert np.array_equal(a[..., 3], b[..., 3])                                  # sample counts
    differ = np.any(a[..., :3] != b[..., :3], axis=-1)
    # identical arithmetic per ray => identical samples, except where two triangles tie exactly (shared edges) or an any-hit ray
    # has several occluders and the trees find different ones first (same verdict)
    assert differ.mean() <= 0.01, differ.mean()
    assert abs(a[..., :3].mean() - b[..., :3].mean()) <= 2e-3 * a[..., :3].mean()


def test_device_builder_on_awkward_geometry(pkg):
    """Huge coordinate offsets, tiny and huge triangles side by side, flat boxes, many identical centroids (equal Morton codes)."""
    rng = np.random.RandomState(11)
    base = pkg.scenes.open_box(8, 8)
    n = 3000
    centres = rng.uniform(-1, 1, (n, 3)) * np.array([1e3, 1.0, 1e-3]) + np.array([5e4, -3.0, 0.25])
    size = 10.0 ** rng.uniform(-5, 1, (n, 1, 1))
    tri = centres[:, None, :] + size * rng.normal(size=(n, 3, 3))
    tri[::7, :, 1] = tri[::7, :1, 1]
    tri[1000:1400] = tri[1000]                                                    # 400 copies of one triangle
    v = np.concatenate([base.vertex, tri.reshape(-1, 3)])
    nrm = np.concatenate([base.normal, np.tile([[0.0, 1.0, 0.0]], (3 * n, 1))])
    tc = np.concatenate([base.texcoord, np.zeros((3 * n, 2))])
    off = base.vertex.shape[0]
    f = np.zeros((n, 3, 4), np.int32)
    for k in range(3):
        f[:, k, 0] = f[:, k, 1] = f[:, k, 2] = off + 3 * np.arange(n) + k
    scene = pkg.scenes.SceneData("stress", v, nrm, tc, np.concatenate([base.face, f]), base.materials, base.camera)
    rg = _gpu_tree_renderer(pkg, scene); rh = pkg.Renderer(scene)
    assert rg.info().n_tris == base.n_faces + n and rg.info().bvh_depth <= 63
    m = 20000
    o = rng.uniform(-1, 1, (m, 3)) * np.array([1.2e3, 4.0, 4.0]) + np.array([5e4, -3.0, 0.25])
    tgt = centres[rng.randint(0, n, m)] + rng.normal(size=(m, 3)) * 0.5
    d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    tg, ig, _, _ = rg.probe_trace(o, d); th, ih, _, _ = rh.probe_trace(o, d)
    rg.close(); rh.close()
    assert (ih >= 0).mean() > 0.2                                                 # the rays do hit things
    assert ((ig >= 0) == (ih >= 0)).all()
    hit = ih >= 0
    assert np.array_equal(tg[hit], th[hit])                                       # same closest distance, bit for bit
    # the triangle may differ only between exact ties (the 400 coincident copies)
    assert ((ig == ih) | ((ig >= base.n_faces + 1000) & (ig < base.n_faces + 1400))).all()


def test_device_builder_is_faster_on_a_large_scene(pkg):
    scene = pkg.scenes.bathroom_stress(64, 36, detail=160, tex_size=16)          # 0.59 M triangles
    rh = pkg.Renderer(scene); ih = rh.info(); rh.close()
    rg = _gpu_tree_renderer(pkg, scene); ig = rg.info()
    rg.render(4, seed=3); a = rg.read_accum(); rg.close()
    assert np.isfinite(a).all() and (a[..., 3] == 4).all()
    print("\nBVH build, %d triangles: host SAH %.0f ms (depth %d, %d nodes) | device LBVH %.0f ms (depth %d, %d nodes)" % (
        ih.n_tris, ih.bvh_build_ms, ih.bvh_depth, ih.n_nodes, ig.bvh_build_ms, ig.bvh_depth, ig.n_nodes))
    assert ig.bvh_build_ms < ih.bvh_build_ms


@pytest.mark.parametrize("kind", ["lbvh", "ploc"])
def test_both_device_builders_are_sound_and_ploc_is_the_better_tree(pkg, paths, kind):
    """MCPT_FLAG_GPU_BVH_BUILD builds a SAH-costed tree by default (PLOC: every merge minimises the merged box's area within a +-16
    window of the Morton order); MCPT_GPU_BVH=lbvh keeps the plain Karras tree.  Both must pass the soundness walk and return the
    reference's hits through the production kernel; the PLOC tree must cost fewer box tests per ray than the LBVH and stay within
    15 % of the host's binned-SAH tree (measured on S-bath detail 24: host 1.00, PLOC ~1.0, LBVH ~1.3)."""
    os.environ["MCPT_GPU_BVH"] = kind
    try:
        r = _gpu_tree_renderer(pkg, pkg.scenes.cornell_box_small(64, 64))
        t, tri, u, v = r.probe_trace4(paths["cs_ray_o"], paths["cs_ray_d"]); r.close()
        ref_tri = paths["cs_ray_rec"][:, 11].astype(np.int32); ref_hit = paths["cs_ray_hit"] == 1
        assert (tri == np.where(ref_hit, ref_tri, -1)).mean() >= 0.999
        scene = pkg.scenes.bathroom_stress(96, 54, detail=24, tex_size=32)
        cost = {}
        for name, fl in (("host", 0), (kind, pkg.FLAG_GPU_BVH_BUILD)):
            rr = pkg.Renderer(scene, max_depth=6, flags=fl | pkg.FLAG_COUNT_TRAVERSAL | pkg.FLAG_CORRECT_SHADOW_T2)
            rr.render(8, seed=3); c = rr.counters(); i = rr.info(); rr.close()
            cost[name] = c.box_tests / c.rays
            assert i.bvh_depth <= 63
        print(kind, "box tests per ray: host %.1f device %.1f" % (cost["host"], cost[kind]))
        if kind == "ploc":
            assert cost["ploc"] <= 1.15 * cost["host"]
        else:
            assert cost["lbvh"] <= 2.0 * cost["host"]
    finally:
        os.environ.pop("MCPT_GPU_BVH", None)


@pytest.mark.parametrize("name,kw,res", [("cornell-box-small", {}, (48, 48)), ("veach-mis", {"light_lon": 12, "light_lat": 6, "plate_cells": 4}, (64, 36)),
                                         ("bathroom2", {"detail": 24, "tex_size": 32}, (64, 36)), ("bathroom2", {"detail": 100, "tex_size": 32}, (64, 36))])
def test_device_collapse_reproduces_the_host_collapse(pkg, name, kw, res):
    """gpu_collapse_bvh8 (bvh_gpu.hip) against build_bvh8 (scene_build.cpp) on the same device-built binary tree: the same dynamic programme in the
    same double arithmetic, the same octant slots, the same level-by-level numbering -- so the 8-wide records and the leaf order must be the
    host's bit for bit (mcpt_scene_info.wide_tree_hash covers both), and a deterministic render the same film."""
    scene = pkg.scenes.SCENES[name](res[0], res[1], **kw)
    out = []
    for host_collapse in (False, True):
        if host_collapse: os.environ["MCPT_HOST_COLLAPSE"] = "1"
        try:
            r = _gpu_tree_renderer(pkg, scene, max_depth=5, flags=pkg.FLAG_DETERMINISTIC)
        finally:
            os.environ.pop("MCPT_HOST_COLLAPSE", None)
        i = r.info(); r.render(4, seed=3); out.append(((i.wide_nodes, i.wide_depth, i.wide_tree_hash), r.read_accum())); r.close()
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
