"""-m gpu: the HIP path, always through the C ABI (csrc/libmcpt_hip.so), against
  (a) the CPU oracle on the same counter-based seed, (b) golden vectors produced by the REAL reference (tests/golden), and
  (c) size-independent properties at the BASELINE.json resolution.

Tolerances: same-seed GPU vs oracle  |d mean| <= 1e-4 * max(1, mean) per channel on >= 99.9 % of pixels for
S-cornell and S-veach (>= 99 % on S-bath small and >= 99.5 % on the 4 M-triangle scene through a small film: mirror / Ns-2000 path
divergence; fp32 traversal vs the oracle's fp64); statistical GPU vs reference: image mean within 1 %, <= 0.3 % of
pixels beyond 4 sigma; at the bench sizes 8x8-block statistics against full-size goldens of the real reference."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _npz(name):
    with np.load(os.path.join(G, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def kats():
    return _npz("ref_kats.npz")


@pytest.fixture(scope="module")
def paths():
    return _npz("ref_paths.npz")


@pytest.fixture(scope="module")
def images():
    return _npz("ref_images.npz")


def _frac_beyond(gpu_mean, cpu_mean, rel=1e-4):
    tol = rel * np.maximum(1.0, cpu_mean)
    return float(np.mean(np.any(np.abs(gpu_mean - cpu_mean) > tol, axis=-1)))


def _renderer(pkg, scene, pipeline="wave", **kw):
    os.environ["MCPT_PIPELINE"] = pipeline
    try:
        return pkg.Renderer(scene, **kw)
    finally:
        os.environ.pop("MCPT_PIPELINE", None)


# ------------------------------------------------------------------------------------------------ same-seed images
@pytest.mark.parametrize("pipeline", ["wave", "mega"])
@pytest.mark.parametrize("max_depth", [4, 0])
def test_same_seed_image_parity_stable_mode(pkg, orc, max_depth, pipeline):
    """MCPT_FLAG_CORRECT_SHADOW_T2 removes the reference's rounding-level coin flip (SURVEY A-9), so the fp32 GPU path and
    the fp64 oracle follow the same paths."""
    scene = pkg.scenes.cornell_box_small(64, 64)
    spp = 16
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    r = _renderer(pkg, scene, pipeline, max_depth=max_depth, flags=flags)
    r.render(spp, seed=1234)
    g = r.read_accum()
    c = r.counters()
    o = orc.Oracle(scene, max_depth=max_depth, flags=flags)
    cpu, oc, _ = o.render(spp, seed=1234)
    assert np.all(g[..., 3] == spp)
    gm, cm = g[..., :3] / spp, cpu[..., :3] / spp
    frac = _frac_beyond(gm, cm)
    print("pixels beyond tolerance: %.3f%%  image mean gpu %s cpu %s" % (100 * frac, gm.mean((0, 1)), cm.mean((0, 1))))
    assert frac <= 0.001                                  # >= 99.9 % of pixels (measured 99.976 %)
    assert np.allclose(gm.mean((0, 1)), cm.mean((0, 1)), rtol=1e-4)
    assert c.paths == oc["paths"] == 64 * 64 * spp
    assert c.rays_primary == oc["rays_primary"]
    assert abs(int(c.rays_continuation) - oc["rays_continuation"]) <= 0.002 * oc["rays_continuation"]
    assert abs(int(c.rays_shadow) - oc["rays_shadow"]) <= 0.002 * oc["rays_shadow"]
    r.close()


def test_wavefront_and_megakernel_agree(pkg):
    """Two independent kernel formulations of the same spec follow the same paths (stable mode: in reference-faithful mode
    the A-9 self-occlusion verdict hangs on the last bit of fp32 intermediates, which two separately compiled kernels
    contract differently -- there the two agree statistically, like everything else compared with the reference)."""
    scene = pkg.scenes.cornell_box_small(40, 24)
    out = []
    for pipe in ("wave", "mega"):
        r = _renderer(pkg, scene, pipe, max_depth=6, flags=pkg.FLAG_CORRECT_SHADOW_T2)
        r.render(32, seed=9); out.append(r.read_accum()); r.close()
    assert np.array_equal(out[0][..., 3], out[1][..., 3])
    assert _frac_beyond(out[0][..., :3] / 32, out[1][..., :3] / 32) <= 0.01
    assert np.allclose(out[0][..., :3].mean((0, 1)), out[1][..., :3].mean((0, 1)), rtol=1e-3)


@pytest.mark.parametrize("name,kw,res,max_frac", [("veach-mis", {"light_lon": 12, "light_lat": 6, "plate_cells": 4}, (64, 36), 0.001),
                                                  ("bathroom2", {"detail": 12, "tex_size": 32}, (64, 36), 0.01)])
def test_same_seed_other_scenes(pkg, orc, name, kw, res, max_frac):
    """S-veach (1440 light triangles, four Blinn-Phong exponents) and S-bath (image textures, mirror Ns=10000, glossy chrome)."""
    scene = pkg.scenes.SCENES[name](res[0], res[1], **kw)
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    r = pkg.Renderer(scene, max_depth=6, flags=flags); r.render(16, seed=77); g = r.read_accum(); r.close()
    cpu, _, _ = orc.Oracle(scene, max_depth=6, flags=flags).render(16, seed=77)
    gm, cm = g[..., :3] / 16, cpu[..., :3] / 16
    frac = _frac_beyond(gm, cm)
    print(name, "pixels beyond tolerance: %.3f%%" % (100 * frac), gm.mean((0, 1)), cm.mean((0, 1)))
    # measured: S-veach 0.000 %, S-bath 0.74 % (mirror + Ns = 2000 chrome: a 1e-7 difference in a reflected direction moves the next hit
    # across a texel or triangle edge -- path divergence, not bias: the image means agree to 3e-6)
    assert frac <= max_frac
    assert np.allclose(gm.mean((0, 1)), cm.mean((0, 1)), rtol=1e-4)


def test_recursive_nee_integrator(pkg, orc):
    """The reference's dead recursive integrator (Render.cpp:83-109 + sample_light :177-200) as an iterative kernel."""
    scene = pkg.scenes.cornell_box_small(48, 48)
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    r = pkg.Renderer(scene, integrator=pkg.INTEGRATOR_RECURSIVE_NEE, flags=flags); r.render(16, seed=5); g = r.read_accum(); r.close()
    cpu, _, _ = orc.Oracle(scene, integrator=pkg.INTEGRATOR_RECURSIVE_NEE, flags=flags).render(16, seed=5)
    gm, cm = g[..., :3] / 16, cpu[..., :3] / 16
    assert _frac_beyond(gm, cm) <= 0.01
    assert np.allclose(gm.mean((0, 1)), cm.mean((0, 1)), rtol=2e-3)


def _transformed(pkg, scene, scale=1.0, offset=(0.0, 0.0, 0.0)):
    """The same scene scaled about the origin and translated (vertices and camera; every number re-quantised to its file form)."""
    S = pkg.scenes
    off = np.asarray(offset, float)
    v = np.vectorize(S._q)(scene.vertex * scale + off)
    c = scene.camera
    cam = S._qcam(tuple(np.asarray(c.eye) * scale + off), tuple(np.asarray(c.lookat) * scale + off), c.up, c.fovy, c.width, c.height)
    return S.SceneData(scene.name + "-moved", v, scene.normal, scene.texcoord, scene.face, scene.materials, cam, dict(scene.meta))


@pytest.mark.parametrize("scale,offset,max_frac,mean_rtol", [(1.0, (100.0, -3.0, 0.25), 0.001, 1e-4), (1.0, (1000.0, -3.0, 0.25), 0.001, 1e-4),
                                                             (1.0, (5e4, -3.0, 0.25), 0.04, 1e-4),
                                                             (100.0, (0.0, 0.0, 0.0), 0.04, 5e-4), (0.01, (0.0, 0.0, 0.0), 0.002, 1e-4)])
def test_fp32_traversal_envelope_vs_fp64_oracle(pkg, orc, scale, offset, max_frac, mean_rtol):
    """The device intersects in fp32 with the reference's ABSOLUTE ray epsilon t1 = 1e-4 (Render.h:30); the reference (and the oracle)
    intersect in fp64, where that epsilon does not care where the scene sits.  Same seed, S-cornell-small moved away from the origin /
    rescaled.  Round 3: the host subtracts the fp64 centre of the scene's bounding box from every vertex and from the camera before
    anything is rounded to fp32 (DevScene::centre), so a TRANSLATED scene is the at-origin scene again: offsets of 100 and 1000 units
    meet the at-origin bound (<= 0.1 % of pixels beyond 1e-4, image mean to 1e-4; rounds 1-2 measured 2.9 % and 29 % of pixels there).
    At 5e4 units (rounds 1-2: 94 % of pixels, image 5 % darker) 2.7 % of pixels remain and the image means agree to 1e-5: that residue is
    the REFERENCE's own arithmetic -- Render::sample rounds the light point and the hit point to fp32 in world coordinates
    (Render.cpp:207-213; fp32 spacing 4e-3 out there), which both sides reproduce, so a 1e-9 difference in the hit point can land on the
    neighbouring fp32 and turn the shadow ray by a milliradian.  What remains beside it is SCALE: a box 100 units wide has fp32 spacing 4e-6 at its walls against the fixed
    t1 = 1e-4 (1.7 % of pixels | 3e-5 of the mean); scaled x0.01, t1 is 1 % of the box and both sides lose the same contact shadows."""
    scene = _transformed(pkg, pkg.scenes.cornell_box_small(64, 64), scale, offset)
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    r = pkg.Renderer(scene, max_depth=6, flags=flags); r.render(16, seed=3); g = r.read_accum(); r.close()
    cpu, _, _ = orc.Oracle(scene, max_depth=6, flags=flags).render(16, seed=3)
    gm, cm = g[..., :3] / 16, cpu[..., :3] / 16
    frac = _frac_beyond(gm, cm)
    print("scale %g offset %s: pixels beyond tolerance %.2f%%  mean gpu %s cpu %s" % (scale, offset, 100 * frac, gm.mean((0, 1)), cm.mean((0, 1))))
    assert np.isfinite(g).all() and np.all(g[..., 3] == 16)
    if max_frac is not None:
        assert frac <= max_frac
    assert np.allclose(gm.mean((0, 1)), cm.mean((0, 1)), rtol=mean_rtol)


def test_nan_samples_are_scrubbed_like_set_pixel(pkg, orc):
    """Scene::set_Pixel zeroes NaN components of a sample before adding it and still counts the sample (Scene.cpp:16-20).  Zero vertex
    normals on the floor make normalize(0) = NaN the shading normal of every floor hit: the film must stay finite, every pixel
    must have received all its samples, and in the stable mode the film equals the oracle's sample for sample."""
    S = pkg.scenes
    nb = S.open_box(32, 32)
    nrm = nb.normal.copy(); nrm[np.unique(nb.face[:2, :, 1])] = 0.0          # the floor's two triangles
    scene = S.SceneData("nan-box", nb.vertex, nrm, nb.texcoord, nb.face, nb.materials, nb.camera, {})
    for fl in (pkg.FLAG_CORRECT_SHADOW_T2, 0):
        r = pkg.Renderer(scene, max_depth=4, flags=fl); r.render(16, seed=1); g = r.read_accum(); r.close()
        cpu, _, _ = orc.Oracle(scene, max_depth=4, flags=fl).render(16, seed=1)
        assert np.isfinite(g).all() and np.all(g[..., 3] == 16) and np.isfinite(cpu).all()
        clean = S.open_box(32, 32)
        r = pkg.Renderer(clean, max_depth=4, flags=fl); r.render(16, seed=1); gc = r.read_accum(); r.close()
        assert g[..., :3].sum() < 0.9 * gc[..., :3].sum()                     # the NaN samples really were dropped to zero
        if fl:
            assert _frac_beyond(g[..., :3] / 16, cpu[..., :3] / 16) <= 0.002
        else:
            assert np.allclose((g[..., :3] / 16).mean((0, 1)), (cpu[..., :3] / 16).mean((0, 1)), rtol=0.05)


# ------------------------------------------------------------------------------------------------ statistics vs the real reference
@pytest.mark.parametrize("name,max_depth,scene_fn,res", [("cs_unbounded", 0, "cornell_box_small", 64), ("cs_depth4", 4, "cornell_box_small", 64),
                                                        ("ob_unbounded", 0, "open_box", 48)])
def test_gpu_matches_reference_statistics(pkg, images, name, max_depth, scene_fn, res):
    """Default (reference-faithful) mode against per-pixel mean/variance images of the REAL reference renderer."""
    scene = getattr(pkg.scenes, scene_fn)(res, res)
    r = pkg.Renderer(scene, max_depth=max_depth)
    means = []
    for b in range(8):
        r.clear(); r.render(128, seed=321, first_sample=b * 128); a = r.read_accum(); means.append(a[..., :3] / a[..., 3:])
    r.close()
    m = np.stack(means); mean, var = m.mean(0), m.var(0, ddof=1) / 8
    rm, rv = images[name + "_mean"], images[name + "_var"]
    assert np.allclose(mean.mean((0, 1)), rm.mean((0, 1)), rtol=0.01), (mean.mean((0, 1)), rm.mean((0, 1)))
    z = np.abs(mean - rm) / np.sqrt(var + rv + 1e-12)
    frac = float((z > 4).mean())
    print(name, "gpu", mean.mean((0, 1)), "reference", rm.mean((0, 1)), "pixels > 4 sigma %.3f%%" % (100 * frac))
    assert frac <= 0.003                                                  # SURVEY section 8(d) / BASELINE.md: <= 0.3 % of pixels beyond 4 sigma (measured <= 0.16 %)


SCENES2 = {"vm_": ("veach-mis", {"light_lon": 12, "light_lat": 6, "plate_cells": 4}, (64, 36)),
           "bt_": ("bathroom2", {"detail": 12, "tex_size": 32}, (64, 36))}


@pytest.fixture(scope="module")
def g2():
    return _npz("ref_scenes2.npz")


@pytest.mark.parametrize("tag", sorted(SCENES2))
def test_gpu_matches_reference_statistics_more_scenes(pkg, g2, tag):
    """Default (reference-faithful, A-9) mode against the REAL reference's per-pixel mean / variance on S-veach small (480 light
    triangles, Blinn-Phong exponents up to 5000) and S-bath small (image textures, mirror, chrome) -- tests/golden/ref_scenes2.npz."""
    name, kw, res = SCENES2[tag]
    scene = pkg.scenes.SCENES[name](res[0], res[1], **kw)
    r = pkg.Renderer(scene)
    means = []
    for b in range(16):
        r.clear(); r.render(256, seed=321, first_sample=b * 256); a = r.read_accum(); means.append(a[..., :3] / a[..., 3:])
    r.close()
    m = np.stack(means); mean, var = m.mean(0), m.var(0, ddof=1) / 16
    rm, rv = g2[tag + "unbounded_mean"], g2[tag + "unbounded_var"]
    npix = rm.shape[0] * rm.shape[1]
    se = np.sqrt(var.sum((0, 1)) + rv.sum((0, 1))) / npix             # standard error of the difference of the two image means
    dm = np.abs(mean.mean((0, 1)) - rm.mean((0, 1)))
    z = np.abs(mean - rm) / np.sqrt(var + rv + 1e-12)
    frac = float((z > 4).mean())
    print(tag, "gpu", mean.mean((0, 1)), "reference", rm.mean((0, 1)), "se", se, "pixels > 4 sigma %.3f%%" % (100 * frac))
    assert np.all(dm <= 4 * se) and np.all(dm <= 0.03 * rm.mean((0, 1)))
    assert frac <= 0.003                                                  # the contract's 0.3 % (measured <= 0.16 %)


@pytest.mark.parametrize("tag", sorted(SCENES2))
@pytest.mark.parametrize("tree", ["host-sah", "device-lbvh"])
def test_probe_trace4_vs_reference_more_scenes(pkg, g2, tag, tree):
    """BVH::hit / has_hit of the REAL reference on 2 000 rays per scene through the production trace kernel."""
    name, kw, res = SCENES2[tag]
    r = pkg.Renderer(pkg.scenes.SCENES[name](res[0], res[1], **kw), flags=pkg.FLAG_GPU_BVH_BUILD if tree == "device-lbvh" else 0)
    t, tri, u, v = r.probe_trace4(g2[tag + "ray_o"], g2[tag + "ray_d"])
    anyh = r.probe_trace4(g2[tag + "ray_o"], g2[tag + "ray_d"], t2=g2[tag + "ray_t2"], any_hit=True)[1]
    r.close()
    ref_tri = g2[tag + "ray_rec"][:, 11].astype(np.int32); ref_hit = g2[tag + "ray_hit"] == 1
    same = (tri == np.where(ref_hit, ref_tri, -1))
    assert same.mean() >= 0.998, same.mean()
    ok = same & ref_hit
    assert np.allclose(t[ok], g2[tag + "ray_rec"][ok, 0], rtol=2e-5, atol=2e-6)
    assert (anyh == g2[tag + "ray_any"]).mean() >= 0.998


def test_self_occlusion_rate_matches_oracle(pkg, orc):
    """SURVEY A-9 as a scalar: share of light samples rejected by the sampled triangle's own fp64 any-hit test."""
    scene = pkg.scenes.cornell_box_small(64, 64)
    r = pkg.Renderer(scene); r.render(32, seed=8); c = r.counters(); r.close()
    _, oc, _ = orc.Oracle(scene).render(32, seed=8)
    rg = c.self_shadow_hits / c.self_shadow_tests
    ro = oc["self_shadow_hits"] / oc["self_shadow_tests"]
    print("self-occlusion rate gpu %.4f oracle %.4f" % (rg, ro))
    assert 0.3 < ro < 0.9 and abs(rg - ro) < 0.02
    # rays_shadow counts only traversed shadow rays; the oracle (like the reference) traverses the self-blocked ones too
    assert abs((c.rays_shadow + c.self_shadow_hits) - oc["rays_shadow"]) <= 0.02 * oc["rays_shadow"]


# ------------------------------------------------------------------------------------------------ function-level probes
def test_probe_rng_is_the_oracle_stream(pkg, orc):
    r = pkg.Renderer(pkg.scenes.open_box(8, 8))
    keys = np.array([[p, s, b] for p in (0, 1, 77, 639999) for s in (0, 5, 1023, 4_000_000) for b in (0, 1, 2, 17)], np.uint32)
    for seed in (0, 12345, (7 << 32) + 3):
        got = r.probe_rng(keys, seed)
        want = np.array([orc.Oracle.rng_block(int(k[0]), int(k[1]), int(k[2]), seed) for k in keys])
        assert np.array_equal(got, want)
    r.close()


def test_probe_cast_ray_vs_reference(pkg, paths):
    r = pkg.Renderer(pkg.scenes.cornell_box_small(64, 64))
    got = r.probe_cast_ray(paths["cs_cam_xy"], paths["cs_cam_xi"])
    r.close()
    assert np.allclose(got, paths["cs_cam_od"].astype(np.float32), rtol=0, atol=1.2e-7)


def test_probe_trace_vs_reference(pkg, paths):
    """BVH::hit / has_hit on the reference's own random rays: same triangle, same distance, on >= 99.9 % of rays."""
    p = paths
    r = pkg.Renderer(pkg.scenes.cornell_box_small(64, 64))
    t, tri, u, v = r.probe_trace(p["cs_ray_o"], p["cs_ray_d"])
    anyh = r.probe_trace(p["cs_ray_o"], p["cs_ray_d"], t2=p["cs_ray_t2"], any_hit=True)[1]
    r.close()
    ref_tri = p["cs_ray_rec"][:, 11].astype(np.int32); ref_hit = p["cs_ray_hit"] == 1
    same = (tri == np.where(ref_hit, ref_tri, -1))
    assert same.mean() >= 0.999, same.mean()
    ok = same & ref_hit
    assert np.allclose(t[ok], p["cs_ray_rec"][ok, 0], rtol=2e-5, atol=2e-6)
    assert (anyh == p["cs_ray_any"]).mean() >= 0.999


def _same_numbers(t_a, t_b, u_a, u_b, v_a, v_b, rtol=4e-6, atol_uv=4e-6):
    assert np.allclose(t_a, t_b, rtol=rtol, atol=1e-7), np.abs(t_a - t_b).max()
    assert np.allclose(u_a, u_b, rtol=0, atol=atol_uv) and np.allclose(v_a, v_b, rtol=0, atol=atol_uv)


def _trace4_vs_reference(r, p):
    t, tri, u, v = r.probe_trace4(p["cs_ray_o"], p["cs_ray_d"])
    anyh = r.probe_trace4(p["cs_ray_o"], p["cs_ray_d"], t2=p["cs_ray_t2"], any_hit=True)[1]
    ref_tri = p["cs_ray_rec"][:, 11].astype(np.int32); ref_hit = p["cs_ray_hit"] == 1
    same = (tri == np.where(ref_hit, ref_tri, -1))
    assert same.mean() >= 0.999, same.mean()
    ok = same & ref_hit
    assert np.allclose(t[ok], p["cs_ray_rec"][ok, 0], rtol=2e-5, atol=2e-6)
    assert (anyh == p["cs_ray_any"]).mean() >= 0.999, (anyh == p["cs_ray_any"]).mean()
    return t, tri, u, v, anyh


@pytest.mark.parametrize("tree", ["host-sah", "device-lbvh"])
@pytest.mark.parametrize("grid", [0, 1])
def test_probe_trace4_hot_kernel_vs_reference(pkg, paths, tree, grid):
    """The PRODUCTION traversal (wf_trace8_kernel: 8-wide compressed nodes, LDS top levels, LDS + overflow group stack, chunked ray list)
    on the reference's own 4 000 rays: BVH::hit (same triangle, same t) and BVH::has_hit (same verdict with the reference's t2) on
    >= 99.9 % of rays -- with the host SAH tree and the device-built LBVH, on the full persistent grid and on ONE block
    (MCPT_WF_GRID=1: every chunk of the ray list comes from the atomic cursor)."""
    if grid: os.environ["MCPT_WF_GRID"] = str(grid)
    try:
        r = pkg.Renderer(pkg.scenes.cornell_box_small(64, 64), flags=pkg.FLAG_GPU_BVH_BUILD if tree == "device-lbvh" else 0)
    finally:
        os.environ.pop("MCPT_WF_GRID", None)
    t4, tri4, u4, v4, any4 = _trace4_vs_reference(r, paths)
    # and against the binary-tree cross-check traversal: the same triangle-test source (tri_test in pt_device.h), inlined into two
    # kernels that contract its FMAs differently -> equal to a few ulp wherever both pick the same triangle
    t2_, tri2, u2, v2 = r.probe_trace(paths["cs_ray_o"], paths["cs_ray_d"])
    r.close()
    same = tri4 == tri2
    assert same.mean() >= 0.9995
    _same_numbers(t4[same], t2_[same], u4[same], u2[same], v4[same], v2[same])


def test_probe_trace4_sparse_ray_lists(pkg, paths):
    """Slots without a pending extend ray (dead / draining paths: bit 0 of ray_d.w clear) sit between live ones in the trace kernel's ray list
    whenever a job is running out -- the last iterations of every job, all but the first of a one-sample-per-pixel call.  mcpt_probe_trace4
    leaves such holes where a ray's direction is all-zero: the reference's 4 000 rays with holes punched in -- one ray in 17 left, every
    second ray, whole 256-slot blocks empty, one block with a single ray -- must come back with the answers the full list gives, ray for ray
    (same triangle, same t / u / v bit for bit: the traversal of a ray does not depend on its neighbours), and the holes as misses.
    (Round 4 also tried handing the kernel COMPACTED lists for such blocks: no gain on a one-sample call -- its launches are bound by their
    longest ray, not by the holes -- and +2.5 % on the steady state; dropped)"""
    r = pkg.Renderer(pkg.scenes.cornell_box_small(64, 64))
    o, d = paths["cs_ray_o"], paths["cs_ray_d"]
    n = len(o)
    t0, tri0, u0, v0 = r.probe_trace4(o, d)
    idx = np.arange(n)
    patterns = {"1 in 17": idx % 17 == 3, "every second": idx % 2 == 0, "blocks 1, 2, 5 empty": ~np.isin(idx // 256, (1, 2, 5)),
                "one ray in block 3": (idx // 256 != 3) | (idx == 3 * 256 + 77), "only block 0": idx < 256, "three rays": np.isin(idx, (5, 1000, 3999))}
    for name, keep in patterns.items():
        dd = d.copy(); dd[~keep] = 0.0
        t, tri, u, v = r.probe_trace4(o, dd)
        assert np.all(tri[~keep] == -1), name
        assert np.array_equal(tri[keep], tri0[keep]) and np.array_equal(t[keep], t0[keep]) and np.array_equal(u[keep], u0[keep]) and np.array_equal(v[keep], v0[keep]), name
    r.close()


def test_probe_hit_shade_vs_reference(pkg, paths):
    """Triangle::hit's shading record (Triangle.cpp:35-46, 68-76: interpolated + normalised vertex normal, interpolated uv, front flag) on
    the reference's own 4 000 rays: the production trace kernel finds the hit (mcpt_probe_trace4), load_hit_shade -- the function the shade
    kernel calls -- turns (triangle, u, v) into the record, and the reference's fp64 record (cs_ray_rec[:, 4:10]) is the yardstick:
    normal and uv to 2e-6 absolute (fp32 barycentrics against fp64 ones), front flag equal wherever |n.d| is not within rounding of 0."""
    p = paths
    r = pkg.Renderer(pkg.scenes.cornell_box_small(64, 64))
    t, tri, u, v = r.probe_trace4(p["cs_ray_o"], p["cs_ray_d"])
    ref_tri = p["cs_ray_rec"][:, 11].astype(np.int32); ok = (p["cs_ray_hit"] == 1) & (tri == ref_tri)
    assert ok.sum() >= 0.99 * (p["cs_ray_hit"] == 1).sum()
    rec = r.probe_hit_shade(tri[ok], u[ok], v[ok], p["cs_ray_d"][ok]); r.close()
    ref = p["cs_ray_rec"][ok]
    assert np.allclose(rec[:, 0:3], ref[:, 4:7], atol=2e-6), np.abs(rec[:, 0:3] - ref[:, 4:7]).max()
    assert np.allclose(rec[:, 3:5], ref[:, 7:9], atol=2e-6), np.abs(rec[:, 3:5] - ref[:, 7:9]).max()
    nd = np.abs((ref[:, 4:7] * p["cs_ray_d"][ok]).sum(-1))
    clear = nd > 1e-5
    assert np.array_equal(rec[clear, 5], ref[clear, 9].astype(np.float32)) and clear.mean() > 0.99


def _needle_forest(pkg, n_needles=6000, seed=5):
    """Long thin triangles along x scattered over a 1 x 1 cross-section: a ray crossing the bundle diagonally in the y-z plane enters
    every child box of every level and hits almost nothing, so the near-first traversal keeps up to 3 deferred children per level."""
    rng = np.random.RandomState(seed)
    m = pkg.scenes._Mesh()
    for _ in range(n_needles):
        y, z = rng.uniform(0, 1, 2); dy, dz = rng.normal(size=2) * 2e-4
        a = m.add_vertex((0.0, y, z), (0, 1, 0), (0, 0)); b = m.add_vertex((4.0, y + dy, z + dz), (0, 1, 0), (1, 0))
        c = m.add_vertex((2.0, y + 3e-4, z + 3e-4), (0, 1, 0), (0.5, 1))
        m.add_tri(a, b, c, 0)
    m.add_quad((-1, 3, -1), (5, 3, -1), (5, 3, 2), (-1, 3, 2), (0, -1, 0), 1)
    mats = [pkg.scenes.Material("needle", kd=(0.5, 0.5, 0.5)), pkg.scenes.Material("light", kd=(0.0, 0.0, 0.0), radiance=(5.0, 5.0, 5.0))]
    return m.finish("needles", mats, pkg.scenes._qcam((2.0, 0.5, 4.0), (2.0, 0.5, 0.0), (0, 1, 0), 40.0, 16, 16))


@pytest.mark.parametrize("tree", ["host-sah", "device-lbvh"])
def test_probe_trace4_overflow_stack_and_random_rays(pkg, orc, tree):
    """Rays that push the per-lane group stack past its LDS levels (WF8_LDS_STACK = 8 since r04: the forest is 80 000 needles, an 8-level
    wide tree) into the global overflow area (counted by the kernel), compared with the binary-tree traversal on 200 000 rays and with
    the fp64 oracle (the reference's BVH::hit restated) on 3 000 of them."""
    scene = _needle_forest(pkg, n_needles=80000)
    rng = np.random.RandomState(11)
    n = 200_000
    o = np.stack([rng.uniform(0.2, 3.8, n), rng.uniform(-0.5, -0.1, n), rng.uniform(-0.5, 1.5, n)], 1)
    tgt = np.stack([o[:, 0] + rng.normal(size=n) * 0.05, rng.uniform(1.1, 1.5, n), rng.uniform(-0.5, 1.5, n)], 1)
    d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    fl = pkg.FLAG_COUNT_TRAVERSAL | (pkg.FLAG_GPU_BVH_BUILD if tree == "device-lbvh" else 0)
    r = pkg.Renderer(scene, flags=fl)
    t4, tri4, u4, v4 = r.probe_trace4(o, d)
    c = r.counters()
    t2_, tri2, u2, v2 = r.probe_trace(o, d)
    tlim = rng.uniform(0.5, 3.0, n)
    any4 = r.probe_trace4(o, d, t2=tlim, any_hit=True)[1]
    any2 = r.probe_trace(o, d, t2=tlim, any_hit=True)[1]
    r.close()
    print(tree, "stack spills", c.stack_spills, "box tests/ray %.1f" % (c.box_tests / n), "hit rate %.3f" % (tri4 >= 0).mean())
    assert c.stack_spills > 0, "the workload did not reach the overflow stack"
    same = tri4 == tri2
    assert same.mean() >= 0.9999, same.mean()
    _same_numbers(t4[same], t2_[same], u4[same], u2[same], v4[same], v2[same], rtol=2e-5, atol_uv=1e-4)   # 4-unit-long slivers 3e-4 wide
    assert (any4 == any2).mean() >= 0.9999
    o_ = orc.Oracle(scene)
    k = 3000
    ref = [o_.bvh_hit(o[i], d[i]) for i in range(k)]
    ref_hit = np.array([h for h, _ in ref]) == 1; ref_t = np.array([rec[0] for _, rec in ref]); ref_tri = np.array([int(rec[11]) for _, rec in ref])
    agree = (tri4[:k] >= 0) == ref_hit
    assert agree.mean() >= 0.998, agree.mean()                            # needles are 3e-4 wide: fp32 vs fp64 at their edges
    both = agree & ref_hit & (tri4[:k] == ref_tri)
    assert both.sum() >= 0.99 * ref_hit.sum() and np.allclose(t4[:k][both], ref_t[both], rtol=2e-5, atol=2e-6)


def test_trace4_equals_binary_traversal_on_bench_scene(pkg):
    """1 000 000 random rays through S-cornell (39 612 triangles, the bench scene): production kernel == cross-check traversal."""
    scene = pkg.scenes.cornell_box(64, 64)
    rng = np.random.RandomState(2)
    n = 1_000_000
    o = rng.uniform(0.02, 0.98, (n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = pkg.Renderer(scene)
    t4, tri4, u4, v4 = r.probe_trace4(o, d); t2_, tri2, u2, v2 = r.probe_trace(o, d)
    tlim = rng.uniform(0.05, 1.0, n)
    any4 = r.probe_trace4(o, d, t2=tlim, any_hit=True)[1]; any2 = r.probe_trace(o, d, t2=tlim, any_hit=True)[1]
    r.close()
    same = tri4 == tri2
    assert (tri4 >= 0).mean() > 0.8 and same.mean() >= 0.9999, same.mean()   # (the box is open towards the camera) ties on shared edges only
    _same_numbers(t4[same], t2_[same], u4[same], u2[same], v4[same], v2[same])
    assert (any4 == any2).mean() >= 0.9999


def test_probe_bsdf_vs_reference(pkg, kats):
    k = kats
    sel = k["bsdf_kind"] != 3
    r = pkg.Renderer(pkg.scenes.open_box(8, 8))
    out = r.probe_bsdf(k["bsdf_n"][sel], k["bsdf_wi"][sel], k["bsdf_kd"][sel], k["bsdf_ks"][sel], k["bsdf_ns"][sel], k["bsdf_wo"][sel], k["bsdf_xi"][sel])
    r.close()
    ev, smp = k["bsdf_eval"][sel], k["bsdf_sample"][sel]
    # inputs are rounded to fp32 at the probe boundary (the reference keeps normal/wi in fp64), GPU libm differs by ulps, and
    # Blinn-Phong with Ns = 5000 amplifies both: relative 2e-3 on values, exact on the discrete outcome (mirror flag, failure)
    assert np.allclose(out[:, 0:4], ev, rtol=3e-3, atol=1e-5)
    assert np.array_equal(out[:, 11], smp[:, 7])
    fail_ref = smp[:, 6] == 0
    assert (np.abs(out[:, 10][fail_ref]) < 1e-6).mean() > 0.98           # failed samples (pdf 0) fail on the device too
    good = ~fail_ref & (out[:, 10] != 0)
    assert good.sum() > 400 and good.sum() >= 0.98 * (~fail_ref).sum()
    assert np.allclose(out[good, 4:7], smp[good, 0:3], atol=2e-3)
    # f and pdf of the sample, by exponent: Blinn-Phong with Ns >= 1000 amplifies the fp32 rounding of the probe's inputs (the reference
    # keeps normal / wi in fp64) through pow(cos, Ns); below that the values agree to 2e-3
    ns = k["bsdf_ns"][sel]
    lo = good & (ns < 1000); hi = good & (ns >= 1000)
    assert lo.sum() > 200 and hi.sum() > 100
    assert np.allclose(out[lo, 7:11], smp[lo, 3:7], rtol=2e-3, atol=1e-5)
    assert np.allclose(out[hi, 7:11], smp[hi, 3:7], rtol=2e-2, atol=1e-4)


def test_probe_texture_vs_reference(pkg, kats):
    """Texture::get_color (model.cpp:30-41) on the device against the reference's own lookups: an 8x5 image, uv from -2..3 plus the
    edge cases (0, 1, 0.9995, -0.0001).  The device receives uv in fp32 (it interpolates them in fp32), the reference in fp64."""
    S = pkg.scenes
    base = S.open_box(8, 8)
    mats = list(base.materials)
    mats[0] = S.Material(mats[0].name, kd=mats[0].kd, map_kd="t.ppm", texture=np.ascontiguousarray(kats["tex_img"], np.float32))
    scene = S.SceneData("tex-box", base.vertex, base.normal, base.texcoord, base.face, mats, base.camera, {})
    r = pkg.Renderer(scene)
    got = r.probe_texture(0, kats["tex_uv"])
    const = r.probe_texture(1, kats["tex_uv"][:4])
    r.close()
    same = np.all(got == kats["tex_rgb"], axis=1)
    assert same.mean() >= 0.99, same.mean()                              # fp32 uv can land on the other side of a texel edge
    assert np.all(const == np.asarray(base.materials[1].kd, np.float32))  # Texture(Color3f): one texel, whatever the uv


def test_probe_sample_light_vs_reference(pkg, paths):
    p = paths
    r = pkg.Renderer(pkg.scenes.cornell_box_small(64, 64))
    out = r.probe_sample_light(p["cs_ls_p"], p["cs_ls_xi"])
    r.close()
    ref = p["cs_ls_out"]
    assert np.allclose(out[:, 0:3], ref[:, 0:3], atol=1e-6)                  # wo
    assert np.allclose(out[:, 3:6], ref[:, 3:6])                             # radiance
    assert np.allclose(out[:, 6], ref[:, 6], rtol=1e-4)                      # pdf (may be negative: back-facing, A-8)
    assert np.allclose(out[:, 7], ref[:, 7], rtol=1e-6)                      # t2 = float |d|
    scene = pkg.scenes.cornell_box_small(64, 64)
    light_faces = {float(i) for i in range(scene.n_faces) if scene.materials[scene.face[i, 0, 3]].name == "light"}
    assert set(np.unique(out[:, 8])) == light_faces                          # the two light triangles, reported in face order


@pytest.mark.parametrize("pipeline", ["wave", "mega"])
def test_probe_paths_vs_oracle(pkg, orc, pipeline):
    """Whole paths from given rays: the production wavefront pipeline (and the cross-check megakernel) against the oracle."""
    scene = pkg.scenes.cornell_box_small(64, 64)
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    rng = np.random.RandomState(3)
    n = 2000
    o = rng.uniform(0.1, 0.9, (n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = _renderer(pkg, scene, pipeline, max_depth=6, flags=flags)
    got = r.probe_paths(o, d, seed=42); r.close()
    oc = orc.Oracle(scene, max_depth=6, flags=flags)
    want = np.array([oc.trace_path_counter(o[i], d[i], i, 42) for i in range(n)])
    close = np.all(np.abs(got - want) <= 1e-4 * np.maximum(1.0, np.abs(want)), axis=1)
    assert close.mean() >= 0.99, close.mean()


# ------------------------------------------------------------------------------------------------ film / API behaviour
def test_deterministic_flag_is_bit_reproducible(pkg):
    scene = pkg.scenes.cornell_box_small(40, 40)
    imgs = []
    for _ in range(2):
        r = pkg.Renderer(scene, max_depth=5, flags=pkg.FLAG_DETERMINISTIC); r.render(24, seed=6); imgs.append(r.read_accum()); r.close()
    assert np.array_equal(imgs[0], imgs[1])
    r = pkg.Renderer(scene, max_depth=5); r.render(24, seed=6); a = r.read_accum(); r.close()
    assert np.allclose(a, imgs[0], rtol=2e-5, atol=1e-5)


def test_sample_split_accumulates_like_reference_frames(pkg):
    """`spp` calls of Render::render == one mcpt_render(spp); split calls add up; clear / write / read round-trip."""
    scene = pkg.scenes.cornell_box_small(37, 21)                       # not a multiple of the 8x8 tile
    r = pkg.Renderer(scene, max_depth=5)
    r.render(12, seed=2); whole = r.read_accum()
    r.clear(); r.render(5, seed=2, first_sample=0); r.render(7, seed=2, first_sample=5); parts = r.read_accum()
    assert np.array_equal(whole[..., 3], np.full((21, 37), 12.0)) and np.array_equal(parts[..., 3], whole[..., 3])
    assert np.allclose(parts, whole, rtol=2e-5, atol=1e-5)
    r.write_accum(whole * 2); assert np.array_equal(r.read_accum(), whole * 2)
    r.clear(); assert not r.read_accum().any()
    r.close()


def test_tonemap_matches_reference_film(pkg, orc, kats):
    """Scene::getPixelsColor on the device vs the reference's own output for the same accumulator (Scene.cpp:23-33)."""
    acc = kats["film_accum"]; h, w = acc.shape[:2]
    scene = pkg.scenes.cornell_box_small(w, h)
    r = pkg.Renderer(scene); r.write_accum(acc)
    u8 = r.tonemap(); flipped = r.tonemap(flip_y=True); r.close()
    ref = kats["film_u8"].astype(np.int32)
    assert np.abs(u8.astype(np.int32) - ref).max() <= 1 and (u8 == kats["film_u8"]).mean() > 0.99    # powf ulp at a rounding edge
    assert np.array_equal(flipped, u8[::-1])


def test_external_accumulator_and_stream(pkg):
    import torch
    scene = pkg.scenes.open_box(16, 16)
    r = pkg.Renderer(scene, max_depth=4)
    buf = torch.zeros(16 * 16 * 4, dtype=torch.float32, device="cuda")
    r.bind_accum(buf.data_ptr()); r.set_torch_stream(torch.cuda.current_stream())     # torch's default stream = the legacy null stream
    r.render(4, seed=1); r.sync(); torch.cuda.synchronize()
    a = buf.cpu().numpy().reshape(16, 16, 4)
    assert np.all(a[..., 3] == 4) and a[..., :3].sum() > 0
    r.bind_accum(0); r.set_stream(0)
    assert not r.read_accum().any()                                     # the internal buffer was never touched
    r.close()


@pytest.mark.parametrize("which", ["side", "null"])
def test_render_is_stream_ordered_with_the_callers_stream(pkg, which):
    """bench.py's usage: film clear, render and a read of the film on ONE caller stream, with long kernels queued in front.  If the
    library launched on its own stream instead (handle 0 used to mean that), the render would race the clear and the clone."""
    import torch
    scene = pkg.scenes.open_box(64, 64)
    r = pkg.Renderer(scene, max_depth=4)
    buf = torch.zeros(64 * 64 * 4, dtype=torch.float32, device="cuda")
    r.bind_accum(buf.data_ptr())
    s = torch.cuda.Stream() if which == "side" else torch.cuda.default_stream()
    r.set_torch_stream(s)
    big = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for _ in range(30):
            big = (big @ big) * 1e-3                 # ~100 ms of queued work in front of the clear
        buf.fill_(7.0)
        buf.zero_()
        r.render(4, seed=1)
        snap = buf.clone()                           # ordered after the render's join on the same stream
        buf.zero_()
    s.synchronize(); torch.cuda.synchronize()
    a = snap.cpu().numpy().reshape(64, 64, 4)
    assert np.all(a[..., 3] == 4), np.unique(a[..., 3])
    assert not buf.cpu().numpy().any()
    r.bind_accum(0); r.set_stream(0); r.close()


# ------------------------------------------------------------------------------------------------ BASELINE.json sizes: properties
def test_full_size_properties_cornell_800(pkg):
    """configs[1] geometry (800x800, depth 8, 39 612 triangles) at reduced spp: sample-count plane, linearity of the film in
    the sample range, finiteness, ray accounting."""
    scene = pkg.scenes.cornell_box(800, 800)
    r = pkg.Renderer(scene, max_depth=8)
    r.render(8, seed=1, first_sample=0); a = r.read_accum(); c1 = r.counters()
    r.render(8, seed=1, first_sample=8); ab = r.read_accum(); c2 = r.counters()
    r.clear(); r.reset_counters(); r.render(16, seed=1); whole = r.read_accum(); c = r.counters()
    r.close()
    assert np.all(a[..., 3] == 8) and np.all(ab[..., 3] == 16) and np.all(whole[..., 3] == 16)
    assert np.isfinite(whole).all()          # (negative values are legal: back-facing light pdfs are not clamped, SURVEY A-8)
    assert np.allclose(ab, whole, rtol=3e-5, atol=1e-5)
    assert c.paths == 800 * 800 * 16 == c.rays_primary
    assert c2.rays - c1.rays > 0 and abs(c.rays - c2.rays) <= 1e-6 * c.rays
    assert 4.0 < c.rays / c.paths < 9.0
    m = (whole[..., :3] / 16).mean((0, 1))
    assert np.allclose(m, [0.3264, 0.2208, 0.0696], rtol=0.02)          # image mean of S-cornell (CPU reference: 0.3305 0.2229 0.0701 at 4 spp)


def test_cpp_cli_end_to_end(pkg, tmp_path):
    """The C++ host path (Model -> Render -> Scene -> PNG) against the Python plumbing on the same seed: same film, same PNG."""
    import subprocess
    from PIL import Image
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "monte-carlo-path-tracer_amd", "csrc", "mcpt_cli")
    s = pkg.scenes.cornell_box_small(48, 40)
    obj = s.write(str(tmp_path))
    out = subprocess.check_output([cli, obj, "--spp", "8", "--depth", "4", "--seed", "5", "--deterministic", "--out", str(tmp_path / "img")]).decode()
    assert "frame: 8" in out and "Mray/s" in out
    png = np.asarray(Image.open(str(tmp_path / "img8.png")))
    r = pkg.Renderer(s, max_depth=4, flags=pkg.FLAG_DETERMINISTIC); r.render(8, seed=5); want = r.tonemap(flip_y=True); r.close()
    assert png.shape == want.shape
    assert (np.abs(png.astype(int) - want.astype(int)) <= 1).mean() > 0.995   # host powf vs device powf at a rounding edge
    # --save-every: progressive images from the device film (tonemap kernel + PNG writer), the film itself stays on the device
    out = subprocess.check_output([cli, obj, "--spp", "8", "--batch", "2", "--save-every", "1", "--depth", "4", "--seed", "5", "--deterministic",
                                   "--out", str(tmp_path / "prog")]).decode()
    assert out.count("Image saved successfully") == 4 and all(os.path.exists(str(tmp_path / ("prog%d.png" % k))) for k in (2, 4, 6, 8))
    r = pkg.Renderer(s, max_depth=4, flags=pkg.FLAG_DETERMINISTIC); r.render(4, seed=5); want4 = r.tonemap(flip_y=True); r.close()
    p4 = np.asarray(Image.open(str(tmp_path / "prog4.png")))                                    # (2 + 2 samples vs 4 in one call, 9-digit scene file: last-bit film differences)
    assert p4.shape == want4.shape and (np.abs(p4.astype(int) - want4.astype(int)) <= 1).mean() > 0.995
    final = np.asarray(Image.open(str(tmp_path / "prog8.png")))
    assert (np.abs(final.astype(int) - want.astype(int)) <= 1).mean() > 0.995


def test_cpp_cli_multi_gpu_path_or_its_failure(pkg, tmp_path):
    """`mcpt_cli --gpus 2`: one process, one context per device, ncclCommInitAll + ncclAllReduce of the films (host/main.cpp).  On a node
    with >= 2 GPUs the two-device film must equal the one-device film (same samples, summed once); on a one-GPU box the second context
    cannot be created and the tool must say so, exit non-zero and write NO image (it used to print 'Image saved' after a failed device)."""
    import subprocess
    import torch
    from PIL import Image
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "monte-carlo-path-tracer_amd", "csrc", "mcpt_cli")
    obj = pkg.scenes.cornell_box_small(40, 32).write(str(tmp_path))
    base = [cli, obj, "--spp", "8", "--depth", "4", "--seed", "5", "--deterministic"]
    subprocess.check_call(base + ["--gpus", "1", "--out", str(tmp_path / "one")], stdout=subprocess.DEVNULL)
    subprocess.check_call(base + ["--gpus", "1", "--shard", "tiles", "--out", str(tmp_path / "tiles")], stdout=subprocess.DEVNULL)
    one = np.asarray(Image.open(str(tmp_path / "one8.png"))); tiles = np.asarray(Image.open(str(tmp_path / "tiles8.png")))
    assert np.array_equal(one, tiles)                                     # one device's "share" of the tiles is the whole image
    p2 = subprocess.run(base + ["--gpus", "2", "--out", str(tmp_path / "two")], capture_output=True, text=True)
    if torch.cuda.device_count() >= 2:
        assert p2.returncode == 0, p2.stderr
        a = one.astype(int); b = np.asarray(Image.open(str(tmp_path / "two8.png"))).astype(int)
        assert (np.abs(a - b) <= 1).mean() > 0.999
        subprocess.check_call(base + ["--gpus", "2", "--shard", "tiles", "--out", str(tmp_path / "two_tiles")], stdout=subprocess.DEVNULL)
        assert np.array_equal(one, np.asarray(Image.open(str(tmp_path / "two_tiles8.png"))))   # disjoint tiles: no summation-order effect at all
        # --save-every with two devices shows the WHOLE film so far (ncclReduce into a scratch film on device 0), like the reference's window
        subprocess.check_call(base + ["--gpus", "1", "--batch", "4", "--save-every", "1", "--out", str(tmp_path / "p1")], stdout=subprocess.DEVNULL)
        subprocess.check_call(base + ["--gpus", "2", "--batch", "4", "--save-every", "1", "--out", str(tmp_path / "p2")], stdout=subprocess.DEVNULL)
        h1 = np.asarray(Image.open(str(tmp_path / "p14.png"))).astype(int); h2 = np.asarray(Image.open(str(tmp_path / "p24.png"))).astype(int)
        assert (np.abs(h1 - h2) <= 1).mean() > 0.999
    else:
        assert p2.returncode != 0 and "Error" in p2.stderr
        assert not os.path.exists(str(tmp_path / "two8.png"))


@pytest.mark.parametrize("name,kw,res,depth", [("veach-mis", {}, (1280, 720), 0), ("bathroom2", {"detail": 160}, (1920, 1080), 8)])
def test_full_size_properties_other_configs(pkg, name, kw, res, depth):
    """configs[2] geometry (S-veach 1280x720, 3840 light triangles) and configs[3] geometry (S-bath 1920x1080, 0.59 M triangles,
    four image textures, mirror) at reduced spp: count plane, finiteness, additivity of sample ranges, ray accounting,
    BVH invariants reported by the library."""
    scene = pkg.scenes.SCENES[name](res[0], res[1], **kw)
    r = pkg.Renderer(scene, max_depth=depth)
    info = r.info()
    assert info.n_tris == scene.n_faces and info.bvh_depth <= 30 and info.max_leaf <= 4
    r.render(4, seed=3, first_sample=0); r.render(4, seed=3, first_sample=4); ab = r.read_accum()
    r.clear(); r.reset_counters(); r.render(8, seed=3); whole = r.read_accum(); c = r.counters(); r.close()
    assert np.all(whole[..., 3] == 8) and np.all(ab[..., 3] == 8) and np.isfinite(whole).all()
    assert np.allclose(ab, whole, rtol=1e-4, atol=1e-4)
    assert c.paths == res[0] * res[1] * 8 == c.rays_primary
    assert 2.0 < c.rays / c.paths < 12.0 and c.self_shadow_hits <= c.self_shadow_tests
    m = (whole[..., :3] / 8).mean()
    assert 0.01 < m < 10.0


# ------------------------------------------------------------------------------------------------ configs[4] (C5): the HBM-bound case
@pytest.fixture(scope="module")
def c5_scene(pkg):
    """S-bath stress at BASELINE.json configs[4] size: 3840x2160, 4.05 M triangles (2.6 GB of scene on the device: 8-wide nodes +
    triangle records = 229 MB, beyond L2 and the 256 MB Infinity Cache -> the trace kernel takes its whole-chip grid)."""
    return pkg.scenes.bathroom_stress(3840, 2160, detail=420)


def test_c5_full_size_properties(pkg, c5_scene):
    """configs[4] geometry at reduced spp, depth 16: count plane, finiteness, additivity of sample ranges, ray accounting, BVH
    invariants -- the size-independent properties, at the size where BVH_node::hit (BVH.cpp:95-113) stops being cache-resident."""
    scene = c5_scene
    assert scene.n_faces >= 4_000_000
    r = pkg.Renderer(scene, max_depth=16)
    info = r.info()
    assert info.n_tris == scene.n_faces and info.bvh_depth <= 30 and info.max_leaf <= 4
    assert info.n_nodes * 64 + info.n_tris * 48 > 256 << 20           # traversal data larger than the Infinity Cache
    r.render(2, seed=3, first_sample=0); r.render(2, seed=3, first_sample=2); ab = r.read_accum()
    r.clear(); r.reset_counters(); r.render(4, seed=3); whole = r.read_accum(); c = r.counters(); r.close()
    assert np.all(whole[..., 3] == 4) and np.all(ab[..., 3] == 4) and np.isfinite(whole).all()
    assert np.allclose(ab, whole, rtol=1e-4, atol=1e-4)
    assert c.paths == 3840 * 2160 * 4 == c.rays_primary
    assert 2.0 < c.rays / c.paths < 14.0 and c.self_shadow_hits <= c.self_shadow_tests
    m = (whole[..., :3] / 4).mean((0, 1))
    print("C5 image mean", m, "rays/path %.2f" % (c.rays / c.paths), "kernel %.1f ms" % c.kernel_ms, "%.0f Mray/s" % (c.rays / c.kernel_ms / 1e3))
    assert np.all(m > 0.05) and np.all(m < 10.0)


def test_c5_same_seed_vs_oracle_small_view(pkg, orc, c5_scene):
    """The SAME 4 M-triangle scene at depth 16 through a 64x36 film, same counter-based seed, against the fp64 oracle (the
    reference's midpoint BVH + unordered recursion restated): deep tree, the non-cache-resident trace grid and the overflow stack
    against fp64, which the full-size property test cannot show."""
    scene = c5_scene.with_resolution(64, 36)
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    spp = 8
    r = pkg.Renderer(scene, max_depth=16, flags=flags | pkg.FLAG_COUNT_TRAVERSAL)
    r.render(spp, seed=55); g = r.read_accum(); c = r.counters(); r.close()
    o = orc.Oracle(scene, max_depth=16, flags=flags)
    cpu, oc, secs = o.render(spp, seed=55); o.close()
    gm, cm = g[..., :3] / spp, cpu[..., :3] / spp
    frac = _frac_beyond(gm, cm)
    print("C5 small view: pixels beyond tolerance %.3f%%  mean gpu %s cpu %s  box tests/ray %.1f tri tests/ray %.2f spills %d  oracle %.1f s" % (
        100 * frac, gm.mean((0, 1)), cm.mean((0, 1)), c.box_tests / c.rays, c.tri_tests / c.rays, c.stack_spills, secs))
    assert np.all(g[..., 3] == spp)
    assert frac <= 0.005                                  # measured 0.22 %
    assert np.allclose(gm.mean((0, 1)), cm.mean((0, 1)), rtol=1e-4)
    assert c.rays_primary == oc["rays_primary"] == 64 * 36 * spp
    assert abs(int(c.rays_continuation) - oc["rays_continuation"]) <= 0.01 * oc["rays_continuation"]


def test_c5_device_built_tree_same_seed_vs_oracle(pkg, orc, c5_scene):
    """The 4 M-triangle scene with the tree built ON THE DEVICE (PLOC over the Morton order + level-synchronous 8-wide collapse and
    quantisation, MCPT_FLAG_GPU_BVH_BUILD): the binary tree comes out deeper than the cross-check kernels' 64-entry stack (69 levels),
    which a wavefront-only context accepts -- the production kernel walks the 8-wide collapse (stack sized from its depth) -- while
    mcpt_probe_trace, which would walk the binary tree, refuses.  Same seed against the fp64 oracle through a 64x36 film."""
    scene = c5_scene.with_resolution(64, 36)
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    spp = 8
    os.environ["MCPT_VALIDATE_BVH"] = "1"                                  # host-side soundness walk of the device-made 8-wide tree
    try:
        r = pkg.Renderer(scene, max_depth=16, flags=flags | pkg.FLAG_GPU_BVH_BUILD | pkg.FLAG_COUNT_TRAVERSAL)
    finally:
        os.environ.pop("MCPT_VALIDATE_BVH", None)
    info = r.info()
    r.render(spp, seed=55); g = r.read_accum(); c = r.counters()
    if info.bvh_depth > 63:
        with pytest.raises(pkg.McptError, match="probe_trace4"):
            r.probe_trace(np.array([[2.0, 1.5, 4.7]]), np.array([[0.0, 0.0, -1.0]]))
    t4 = r.probe_trace4(np.array([[2.0, 1.5, 4.7]]), np.array([[0.0, 0.0, -1.0]]))
    r.close()
    assert t4[1][0] >= 0
    cpu, oc, _ = orc.Oracle(scene, max_depth=16, flags=flags).render(spp, seed=55)
    gm, cm = g[..., :3] / spp, cpu[..., :3] / spp
    frac = _frac_beyond(gm, cm)
    print("C5 device tree: binary depth %d, %d nodes; pixels beyond tolerance %.3f%%, box tests/ray %.1f, spills %d, bvh build %.0f ms" % (
        info.bvh_depth, info.n_nodes, 100 * frac, c.box_tests / c.rays, c.stack_spills, info.bvh_build_ms))
    assert np.all(g[..., 3] == spp) and frac <= 0.005
    assert np.allclose(gm.mean((0, 1)), cm.mean((0, 1)), rtol=1e-4)


def test_smoke_entry_point():
    import __graft_entry__ as ge
    ge.smoke()


def test_many_materials_use_the_global_table(pkg, orc):
    """More than WF_LDS_MATS (32) materials: the shade kernel reads materials from global memory instead of its LDS copy."""
    base = pkg.scenes.cornell_box_small(40, 40)
    reps = 9                                                            # 5 materials x 9 = 45 > 32
    mats = [pkg.scenes.Material("%s_%d" % (m.name, k), m.kd, m.ks, m.ns, m.radiance) for k in range(reps) for m in base.materials]
    face = base.face.copy()
    face[:, :, 3] = base.face[:, :, 3] + len(base.materials) * (np.arange(base.n_faces) % reps)[:, None]
    scene = pkg.scenes.SceneData("many-mats", base.vertex, base.normal, base.texcoord, face, mats, base.camera)
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    r = pkg.Renderer(scene, max_depth=5, flags=flags); r.render(16, seed=4); g = r.read_accum(); r.close()
    cpu, _, _ = orc.Oracle(scene, max_depth=5, flags=flags).render(16, seed=4)
    assert _frac_beyond(g[..., :3] / 16, cpu[..., :3] / 16) <= 0.01
    r2 = pkg.Renderer(base, max_depth=5, flags=flags); r2.render(16, seed=4); g2 = r2.read_accum(); r2.close()
    assert _frac_beyond(g[..., :3] / 16, g2[..., :3] / 16) <= 0.01       # same picture as with the 5-entry (LDS) table


@pytest.mark.parametrize("res", [(1, 1), (3, 2), (64, 64)])
def test_depth_one_and_tiny_images(pkg, orc, res):
    """max_depth = 1: emission at the first hit + one light sample + the emitter-MIS term of the first BSDF-sampled ray."""
    scene = pkg.scenes.cornell_box_small(res[0], res[1])
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    r = pkg.Renderer(scene, max_depth=1, flags=flags); r.render(32, seed=9); g = r.read_accum(); c = r.counters(); r.close()
    cpu, oc, _ = orc.Oracle(scene, max_depth=1, flags=flags).render(32, seed=9)
    assert g.shape == (res[1], res[0], 4) and np.all(g[..., 3] == 32)
    assert np.allclose(g[..., :3].sum() / 32, cpu[..., :3].sum() / 32, rtol=5e-3, atol=1e-4)
    assert c.paths == res[0] * res[1] * 32 and c.rays_continuation <= c.paths     # at most one continuation ray per path


@pytest.mark.parametrize("world", [2, 3, 8])
def test_interleaved_tile_shares_are_disjoint_and_sum_to_the_image(pkg, world):
    """BASELINE.json configs[3] says "pixel-tile shard": mcpt_render_tiles(world, rank) renders the 8x8 tiles t with t % world == rank.
    The shares rendered one after another on this GPU (what `world` GPUs would do at once) must not overlap, must each hold all the
    samples of their pixels, and must sum -- the RCCL all-reduce of the multi-GPU run -- to the film of one mcpt_render call."""
    scene = pkg.scenes.cornell_box_small(83, 45)                          # 11 x 6 tiles, ragged right / top edges
    spp = 6
    r = pkg.Renderer(scene, max_depth=5, flags=pkg.FLAG_DETERMINISTIC)
    r.render(spp, seed=4); whole = r.read_accum()
    total = np.zeros_like(whole); owners = np.zeros(whole.shape[:2], np.int32)
    for rank in range(world):
        r.clear(); r.render_tiles(spp, 4, 0, world, rank); part = r.read_accum()
        assert set(np.unique(part[..., 3])) <= {0.0, float(spp)}
        owners += (part[..., 3] > 0)
        total += part
    r.close()
    assert np.all(owners == 1)                                            # every pixel belongs to exactly one share
    tiles = (np.arange(45)[:, None] // 8) * 11 + (np.arange(83)[None, :] // 8)
    r2 = pkg.Renderer(scene, max_depth=5, flags=pkg.FLAG_DETERMINISTIC); r2.render_tiles(spp, 4, 0, world, 1); one = r2.read_accum(); r2.close()
    assert np.array_equal(one[..., 3] > 0, tiles % world == 1)            # ... the interleaved one
    assert np.array_equal(total, whole)                                   # same samples, same per-pixel order (deterministic mode)


def test_tile_shares_through_the_megakernel_paths(pkg):
    """The same partition through the cross-check megakernel (MCPT_PIPELINE=mega) and the recursive integrator, which map work items to
    tiles in kernels.hip rather than in the wavefront shade kernel."""
    scene = pkg.scenes.cornell_box_small(50, 30)
    for kw, pipe in (({"integrator": pkg.INTEGRATOR_RECURSIVE_NEE}, "wave"), ({}, "mega")):
        r = _renderer(pkg, scene, pipe, max_depth=4, flags=pkg.FLAG_DETERMINISTIC, **kw)
        r.render(5, seed=8); whole = r.read_accum()
        total = np.zeros_like(whole)
        for rank in range(3):
            r.clear(); r.render_tiles(5, 8, 0, 3, rank); total += r.read_accum()
        r.close()
        assert np.array_equal(total, whole), (kw, pipe)


def test_frame_by_frame_equals_one_call(pkg):
    """The reference's usage pattern: `frames` calls of one sample each == one call of `frames` samples."""
    scene = pkg.scenes.cornell_box_small(32, 32)
    r = pkg.Renderer(scene, max_depth=6, flags=pkg.FLAG_DETERMINISTIC)
    for f in range(6):
        r.render(1, seed=3, first_sample=f)
    a = r.read_accum(); r.clear(); r.render(6, seed=3); b = r.read_accum(); r.close()
    assert np.array_equal(a[..., 3], b[..., 3]) and np.allclose(a, b, rtol=2e-5, atol=1e-5)


def test_small_jobs_share_the_trace_grid_without_changing_the_film(pkg, monkeypatch):
    """A job of about a path per trace lane (the one-sample frame of the reference's display loop, main.cpp:26-33) gives each sub-pipeline half of the
    CUs for its trace launches (mcpt_api.cpp: Run::grid).  The grid is scheduling only: one sample per pixel, disjoint tile sets per
    sub-pipeline -> the film is the same bit for bit with the split on and off, at the bounded depth (a job of known length) and at the
    reference's unbounded depth (a polled job), and the next multi-sample call on the same context is untouched by it."""
    scene = pkg.scenes.cornell_box_small(200, 120)
    for depth in (5, 0):
        films = []
        for split in ("1", "0"):
            monkeypatch.setenv("MCPT_WF_SMALL_JOB_SPLIT", split)
            r = pkg.Renderer(scene, max_depth=depth)
            for f in range(3):
                r.render(1, seed=9, first_sample=f)
            films.append(r.read_accum()); r.close()
        assert np.all(films[0][..., 3] == 3) and np.array_equal(films[0], films[1]), depth


def test_single_sample_calls_split_tiles_over_both_sub_pipelines(pkg, orc):
    """A call with one sample has nothing to split by sample index: the two sub-pipelines take alternate tiles of the call's share
    instead, and the loop runs its known number of iterations (max_depth + 3) before the first look at the control block.  Same film
    as one multi-sample call, as the fp64 oracle, and per-rank tile shares still add up -- on a film with an odd number of tiles."""
    scene = pkg.scenes.cornell_box_small(72, 40)                        # 9 x 5 = 45 tiles
    spp = 6
    flags = pkg.FLAG_CORRECT_SHADOW_T2                                  # the mode whose samples match the oracle's one for one
    r = pkg.Renderer(scene, max_depth=5, flags=flags)
    r.reset_counters()
    for f in range(spp):
        r.render(1, seed=3, first_sample=f)
    a = r.read_accum(); c = r.counters()
    assert c.iterations == spp * 2 * (5 + 3), c.iterations             # both sub-pipelines ran, each exactly max_depth + 3 iterations
    r.clear(); r.render(spp, seed=3); b = r.read_accum()
    assert np.all(a[..., 3] == spp) and np.array_equal(a[..., 3], b[..., 3]) and np.allclose(a, b, rtol=2e-5, atol=1e-5)
    total = np.zeros_like(a)
    for rank in range(3):
        r.clear()
        for f in range(spp):
            r.render_tiles(1, 3, f, 3, rank)
        total += r.read_accum()
    assert np.array_equal(total[..., 3], a[..., 3]) and np.allclose(total, a, rtol=2e-5, atol=1e-5)
    r.clear(); r.render(1, seed=3, first_sample=0); one = r.read_accum(); r.close()
    cpu, _, _ = orc.Oracle(scene, max_depth=5, flags=flags).render(1, seed=3)
    assert np.all(one[..., 3] == 1) and _frac_beyond(one[..., :3], cpu[..., :3]) <= 0.01
    # unbounded depth has no known length: the polled loop still ends
    r = pkg.Renderer(scene, max_depth=0); r.render(1, seed=3); u = r.read_accum(); r.close()
    assert np.all(u[..., 3] == 1) and np.isfinite(u).all()


def test_block_private_work_items_cover_every_sample_once(pkg):
    """90 % of a call's work items are handed out without atomics: shade block b owns the 64-item units k * n_blocks + b, the rest comes
    from the shared cursors.  That only happens when a call has at least four items per pool slot -- a small pool (developer knob
    MCPT_WF_POOL_LOG2) makes a test-sized film qualify.  Every pixel must get every sample exactly once, the film must equal the
    all-shared hand-out (MCPT_WF_PRIVATE_ITEMS=0) and the default pool's, and interleaved tile shares must still add up."""
    scene = pkg.scenes.cornell_box_small(136, 88)                       # 17 x 11 tiles: nothing divides evenly
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    spp = 12

    def film(env, shares=None):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            r = pkg.Renderer(scene, max_depth=4, flags=flags)
            if shares is None:
                r.render(spp, seed=21); out = r.read_accum()
            else:
                out = np.zeros((88, 136, 4), np.float32)
                for rank in range(shares):
                    r.clear(); r.render_tiles(spp, 21, 0, shares, rank); out += r.read_accum()
            r.close()
            return out
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v

    ref = film({})                                                      # 2^23-slot pools: everything from the shared cursors
    assert np.all(ref[..., 3] == spp)
    for log2 in ("11", "13"):
        a = film({"MCPT_WF_POOL_LOG2": log2})
        b = film({"MCPT_WF_POOL_LOG2": log2, "MCPT_WF_PRIVATE_ITEMS": "0"})
        assert np.all(a[..., 3] == spp) and np.all(b[..., 3] == spp), log2
        assert np.allclose(a, b, rtol=2e-5, atol=1e-5) and np.allclose(a, ref, rtol=2e-5, atol=1e-5), log2
    t = film({"MCPT_WF_POOL_LOG2": "11"}, shares=3)
    assert np.array_equal(t[..., 3], ref[..., 3]) and np.allclose(t, ref, rtol=2e-5, atol=1e-5)


def test_random_api_call_sequences_keep_their_invariants(pkg):
    """A seeded random walk over the C ABI -- render / clear / read / write / tonemap / counters / reset / probes in any order, back to back
    with no pauses -- checking after every observation what must hold: every pixel has exactly the samples rendered since the last clear,
    the device tonemap is the tonemap of that film, the path counter is pixels x samples since the last reset, a probe between two renders
    disturbs nothing, and the final film equals one fresh render of the same sample ranges.  (Found the unordered scratch fills of
    mcpt_tonemap: a default-stream memset overtaken by the kernel on the context's non-blocking stream.)"""
    scene = pkg.scenes.cornell_box_small(56, 40)
    W, H = 56, 40
    rng = np.random.RandomState(17)
    rays_o = np.tile(np.array([[0.5, 0.5, 2.0]]), (64, 1)); rays_d = np.tile(np.array([[0.0, 0.0, -1.0]]), (64, 1))
    for flags in (0, pkg.FLAG_DETERMINISTIC):
        r = pkg.Renderer(scene, max_depth=4, flags=flags)
        count, next_sample, ranges, paths_since_reset = 0, 0, [], 0
        base = np.zeros((H, W, 4), np.float32)                        # what write_accum put under the rendered samples
        r.reset_counters()
        for step in range(70):
            op = rng.choice(["render", "render", "render", "read", "tonemap", "counters", "reset", "clear", "probe", "rewrite"])
            if op == "render":
                n = int(rng.choice([1, 1, 2, 3, 5]))
                r.render(n, seed=9, first_sample=next_sample); ranges.append((next_sample, n)); next_sample += n; count += n; paths_since_reset += n * W * H
            elif op == "read":
                a = r.read_accum()
                assert np.all(a[..., 3] == count + base[..., 3]), (flags, step)
            elif op == "tonemap":
                t = r.tonemap(flip_y=True).astype(int); a = r.read_accum()
                with np.errstate(invalid="ignore", divide="ignore"):
                    m = np.clip(np.nan_to_num(a[..., :3] / a[..., 3:]), 0, 1)
                want = (np.sqrt(m) * 255.99).astype(np.uint8)[::-1].astype(int)
                if count + base[0, 0, 3] > 0:
                    assert (np.abs(t - want) <= 1).all(), (flags, step)
            elif op == "counters":
                assert r.counters().paths == paths_since_reset, (flags, step)
            elif op == "reset":
                r.reset_counters(); paths_since_reset = 0
            elif op == "clear":
                r.clear(); count = 0; ranges = []; base[:] = 0
            elif op == "probe":
                t, tri, _, _ = r.probe_trace4(rays_o, rays_d)
                assert np.all(tri == tri[0]) and np.all(t > 0)
            elif op == "rewrite":                                       # read the film, write it back: a no-op for what follows
                a = r.read_accum(); r.write_accum(a)
        a = r.read_accum()
        assert np.all(a[..., 3] == count)
        fresh = pkg.Renderer(scene, max_depth=4, flags=flags)
        for first, n in ranges:
            fresh.render(n, seed=9, first_sample=first)
        b = fresh.read_accum(); fresh.close(); r.close()
        assert np.allclose(a, b, rtol=3e-5, atol=1e-5), flags


def test_extreme_film_shapes(pkg, orc):
    """1 x 1, a single row, a single column, sizes that are not multiples of the 8 x 8 tile: every pixel gets its samples, nothing is written
    outside the film (the accumulator has exactly w * h records), and the film matches the oracle's."""
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    for w, h in ((1, 1), (257, 1), (1, 130), (9, 7), (3, 65)):
        scene = pkg.scenes.open_box(w, h)
        for kw in ({}, {"flags_extra": pkg.FLAG_DETERMINISTIC}):
            fl = flags | kw.get("flags_extra", 0)
            r = pkg.Renderer(scene, max_depth=4, flags=fl)
            for f in range(0, 12, 4):
                r.render(4, seed=13, first_sample=f)
            g = r.read_accum(); r.close()
            assert g.shape == (h, w, 4) and np.all(g[..., 3] == 12) and np.isfinite(g).all(), (w, h, kw)
        cpu, _, _ = orc.Oracle(scene, max_depth=4, flags=flags).render(12, seed=13)
        assert _frac_beyond(g[..., :3] / 12, cpu[..., :3] / 12) <= 0.02, (w, h)


def test_degenerate_triangles_in_the_scene(pkg, orc):
    """Zero-area triangles -- three collinear vertices, three identical vertices -- as ordinary geometry AND as a light (area 0: the
    reference divides by it in Render.cpp:213-216 and in the emitter MIS, :153-160; whatever comes out is a NaN or inf that
    Scene::set_Pixel scrubs or keeps).  The builders must accept them, the film must stay free of NaN, and GPU and oracle must agree."""
    S = pkg.scenes
    b = S.open_box(40, 32)
    v = np.vstack([b.vertex, [[0.2, 0.5, 0.2], [0.4, 0.5, 0.4], [0.6, 0.5, 0.6], [0.7, 0.3, 0.7]]])      # three collinear points + one more
    n0 = len(b.vertex)
    light_mat = max(i for i, m in enumerate(b.materials) if any(m.radiance))
    wall_mat = min(i for i, m in enumerate(b.materials) if not any(m.radiance))
    def tri(a, bb, c, m): return [[a, 0, 0, m], [bb, 0, 0, m], [c, 0, 0, m]]
    extra = np.array([tri(n0, n0 + 1, n0 + 2, wall_mat), tri(n0 + 3, n0 + 3, n0 + 3, wall_mat), tri(n0, n0 + 1, n0 + 2, light_mat)], np.int32)
    scene = S.SceneData("degenerate", v, b.normal, b.texcoord, np.concatenate([b.face, extra]), b.materials, b.camera)
    st, info, msg = pkg.check_scene(scene)
    assert st == 0, msg
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    for fl in (flags, flags | pkg.FLAG_GPU_BVH_BUILD):
        r = pkg.Renderer(scene, max_depth=4, flags=fl); r.render(16, seed=4); g = r.read_accum(); r.close()
        assert np.all(g[..., 3] == 16) and not np.isnan(g).any(), fl
    cpu, _, _ = orc.Oracle(scene, max_depth=4, flags=flags).render(16, seed=4)
    assert not np.isnan(cpu).any()
    both = np.isfinite(g[..., :3]).all(-1) & np.isfinite(cpu[..., :3]).all(-1)
    assert np.array_equal(np.isfinite(g[..., :3]).all(-1), np.isfinite(cpu[..., :3]).all(-1))          # inf (kept by set_Pixel) in the same pixels
    assert _frac_beyond(g[both][:, :3] / 16, cpu[both][:, :3] / 16) <= 0.02


def test_many_materials_and_many_lights_take_the_global_memory_tables(pkg, orc):
    """The shade kernel stages up to 16 materials and 8 lights in LDS and reads larger tables from global memory.  A scene with 40
    materials (diffuse, glossy, mirror, emissive) spread over the faces and ~30 light triangles takes the other path for both."""
    S = pkg.scenes
    b = S.cornell_box_small(48, 40)
    rng = np.random.RandomState(23)
    mats = []
    for i in range(40):
        kind = i % 5
        kd = tuple(rng.uniform(0.1, 0.8, 3)); ks = (0.0, 0.0, 0.0); ns = 1.0; rad = (0.0, 0.0, 0.0)
        if kind == 1: ks, ns = tuple(rng.uniform(0.1, 0.4, 3)), float(rng.choice([8.0, 60.0, 900.0]))
        if kind == 2: ks, ns = (0.3, 0.3, 0.3), 10000.0
        if kind == 3: rad = tuple(rng.uniform(1.0, 6.0, 3))
        mats.append(S.Material("m%d" % i, kd, ks, ns, rad))
    face = b.face.copy()
    pick = rng.randint(0, 40, len(face))
    pick[rng.rand(len(face)) < 0.9] //= 1                                 # (all faces re-assigned)
    emissive = [i for i in range(40) if i % 5 == 3]
    lights = rng.choice(len(face), 30, replace=False)
    for f in range(len(face)):
        m = int(pick[f])
        if m % 5 == 3 and f not in lights: m = (m + 1) % 40               # keep the number of light triangles at ~30
        face[f, :, 3] = m
    for f in lights: face[f, :, 3] = int(rng.choice(emissive))
    scene = S.SceneData("many-tables", b.vertex, b.normal, b.texcoord, face, mats, b.camera)
    st, info, msg = pkg.check_scene(scene)
    assert st == 0 and info.n_lights > 8, (msg, info.n_lights)
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    r = pkg.Renderer(scene, max_depth=5, flags=flags); r.render(16, seed=6); g = r.read_accum(); r.close()
    cpu, _, _ = orc.Oracle(scene, max_depth=5, flags=flags).render(16, seed=6)
    frac = _frac_beyond(g[..., :3] / 16, cpu[..., :3] / 16)
    print("many tables: %d lights, pixels beyond tolerance %.3f %%" % (info.n_lights, 100 * frac))
    assert np.all(g[..., 3] == 16) and frac <= 0.02
    assert np.allclose((g[..., :3] / 16).mean((0, 1)), (cpu[..., :3] / 16).mean((0, 1)), rtol=2e-3)


def test_two_contexts_render_concurrently_from_two_threads(pkg):
    """Two contexts on one device, driven from two host threads at the same time (ctypes drops the GIL inside the calls; mcpt_cli --gpus
    uses one thread per context the same way): same films as when each renders alone."""
    import threading
    scenes = [pkg.scenes.cornell_box_small(64, 48), pkg.scenes.open_box(40, 56)]
    alone = []
    for sc in scenes:
        r = pkg.Renderer(sc, max_depth=5); r.render(24, seed=31); alone.append(r.read_accum()); r.close()
    rs = [pkg.Renderer(sc, max_depth=5) for sc in scenes]
    out, errs = [None, None], []

    def work(i):
        try:
            for f in range(0, 24, 3):
                rs[i].render(3, seed=31, first_sample=f)
            out[i] = rs[i].read_accum()
        except Exception as e:                                          # noqa: BLE001 -- reported below
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for r in rs: r.close()
    assert not errs, errs
    for i in range(2):
        assert np.array_equal(out[i][..., 3], alone[i][..., 3]) and np.allclose(out[i], alone[i], rtol=3e-5, atol=1e-5), i


def test_facade_classes_keep_the_film_on_the_device_until_it_is_read(pkg, tmp_path):
    """host/Render + host/Scene used the way the reference's main.cpp uses its classes: render(scene) once per sample, film read at the
    end.  The samples stay in HBM between calls (Scene::attach / sync); two Renders sharing a Scene, a Scene that dies with unread
    samples and a Render that dies before its Scene are all folded in correctly.  Checked sample for sample against the C ABI."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "monte-carlo-path-tracer_amd", "csrc"); host = os.path.join(root, "monte-carlo-path-tracer_amd", "host")
    exe = str(tmp_path / "facade_main")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + host, os.path.join(root, "tests", "facade_main.cpp"),
                           os.path.join(csrc, "libmcpt_host.a"), "-o", exe, "-L" + csrc, "-lmcpt_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-lpthread",
                           "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"])
    scene = pkg.scenes.cornell_box_small(40, 24)
    obj = scene.write(str(tmp_path / "scene"))
    q = lambda a: np.array([[float("%.9g" % x) for x in row] for row in a])          # the file holds 9 significant digits
    scene = pkg.scenes.SceneData(scene.name, q(scene.vertex), q(scene.normal), q(scene.texcoord), scene.face, scene.materials, scene.camera)
    frames, depth = 5, 4
    out = str(tmp_path / "film.bin")
    line = subprocess.check_output([exe, obj, str(frames), str(depth), out], timeout=300).decode().split("\n")[-2].split()
    got = np.fromfile(out, np.float32).reshape(24, 40, 4)
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    r = pkg.Renderer(scene, max_depth=depth, flags=flags)
    for f in list(range(frames)) + [frames + 2]:
        r.render(1, seed=11, first_sample=f)
    r.render(2, seed=12, first_sample=0)
    want = r.read_accum(); r.close()
    want[0, 0, :3] += (0.25, 0.5, 0.75); want[0, 0, 3] += 1
    assert np.array_equal(got[..., 3], want[..., 3]) and got[1, 1, 3] == frames + 3
    assert np.allclose(got, want, rtol=2e-5, atol=1e-5)
    m = np.clip(want[0, 0, :3] / want[0, 0, 3], 0, 1)
    assert [int(x) for x in line[2:5]] == [int(v) for v in (np.sqrt(m) * 255.99).astype(np.uint8)]   # Scene::getPixelsColor (Scene.cpp:25-29)


def test_shade_early_gathers_with_miss_lanes(pkg, orc):
    """The shade kernel requests the hit triangle's shading record, its fp64 plane and the diffuse texel EARLY (phase 1; r03 commits 7bf7f96 /
    245810f), indexed by the hit record the trace kernel wrote.  A miss is hit.x = -1, whose masked index would be 0x0fffffff -- far outside
    every triangle stream: an r03 A/B variant that issued such a gather for ALL lanes (before the class sort, where miss / dead lanes still
    take part) died with a memory access fault on three scenes (gpurun_out/ab_touch.log).  The shipped requests sit behind
    `key != K_END`, which holds only for a live slot whose extend ray this iteration's trace launch answered with a hit (index < n_tris).
    Here the waves are MOSTLY miss lanes: (a) the camera looks out of the open side of a box -- primary rays miss or graze one wall --,
    (b) the camera looks at the box from far away (a few pixels hit), (c) one textured triangle among untextured ones, seen from outside
    (texel gathers for a handful of lanes of a wave, none for the rest).  Same seed against the oracle; every pixel keeps its sample count."""
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    base = pkg.scenes.open_box(48, 48)
    C = pkg.scenes.Camera
    lo, hi = base.vertex.min(0), base.vertex.max(0); mid = 0.5 * (lo + hi); ext = float((hi - lo).max())
    cam0 = base.camera
    away = C(tuple(cam0.eye), tuple(2 * np.asarray(cam0.eye) - np.asarray(cam0.lookat)), cam0.up, cam0.fovy, 48, 48)       # looks the other way: every primary ray misses
    far = C(tuple(np.asarray(cam0.eye) + 12 * ext * (np.asarray(cam0.eye) - mid) / np.linalg.norm(np.asarray(cam0.eye) - mid)), tuple(mid), cam0.up, cam0.fovy, 48, 48)
    tex = pkg.scenes.value_noise_texture(16, 3)
    mats = list(base.materials) + [pkg.scenes.Material("tex", kd=(0.5, 0.5, 0.5), map_kd="one.ppm", texture=tex)]
    face = base.face.copy(); face[0, :, 3] = len(mats) - 1                # ONE textured triangle (half of the floor)
    scenes = [("away", pkg.scenes.SceneData("away", base.vertex, base.normal, base.texcoord, base.face, base.materials, away)),
              ("far", pkg.scenes.SceneData("far", base.vertex, base.normal, base.texcoord, base.face, base.materials, far)),
              ("one-texture", pkg.scenes.SceneData("onetex", base.vertex, base.normal, base.texcoord, face, mats, cam0))]
    for name, sc in scenes:
        for fl in (flags, 0):                                              # the same-seed comparison in the stable mode; the default mode once for the fault alone
            r = pkg.Renderer(sc, max_depth=5, flags=fl); r.render(16, seed=9); g = r.read_accum(); c = r.counters(); r.close()
            assert np.all(g[..., 3] == 16) and np.isfinite(g).all(), name
            if fl:
                cpu, oc, _ = orc.Oracle(sc, max_depth=5, flags=fl).render(16, seed=9)
                frac = _frac_beyond(g[..., :3] / 16, cpu[..., :3] / 16)
                hit_share = float((cpu[..., :3].sum(-1) > 0).mean())
                print("%s: lit pixels %.3f, pixels beyond tolerance %.3f %%, rays %d (oracle %d)" % (name, hit_share, 100 * frac, c.rays, oc["rays_primary"] + oc["rays_continuation"] + oc["rays_shadow"]))
                assert frac <= 0.002, name
                assert c.rays_primary == 48 * 48 * 16
        if name == "away": assert g[..., :3].max() == 0.0                # nothing is hit, nothing is shaded


def test_drain_compaction_changes_nothing_but_the_sweep(pkg, tmp_path):
    """Round 4: at the end of a job the live slots are moved to the front of the pool once at most half of the swept ones are alive, and the kernels
    sweep only them (wf_compact_*).  A slot's number means nothing to the path it holds, so the film must be the film of the same
    job without compaction up to fp32 summation order: S-cornell 1024x1024 x 6 spp (two sub-pipelines of 3 M items on pools of 3 M slots: big enough
    for the compaction to be armed), MCPT_WF_COMPACT = 1 against 0 in two child processes (the knob is read when the pools are allocated); the run with
    compaction must report at least one (MCPT_WF_DEBUG prints the control block's counters), every pixel keeps its 6 samples, ray counts are equal."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import __graft_entry__ as ge; pkg = ge.load_package();"
            "r = pkg.Renderer(pkg.scenes.cornell_box(1024, 1024), max_depth=0); r.render(6, seed=5); a = r.read_accum(); c = r.counters(); r.close();"
            "np.save(sys.argv[1], a); print('RAYS', c.rays, c.paths)") % root
    films, rays, logs = [], [], []
    for knob in ("1", "0"):
        out = str(tmp_path / ("film%s.npy" % knob))
        env = dict(os.environ, MCPT_WF_COMPACT=knob, MCPT_WF_DEBUG="1")
        p = subprocess.run([sys.executable, "-c", code, out], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        films.append(np.load(out)); rays.append([l for l in p.stdout.splitlines() if l.startswith("RAYS")][0]); logs.append(p.stderr)
    import re
    done = [int(x) for x in re.findall(r"compactions=(\d+)", logs[0])]
    assert done and max(done) >= 1, "the compaction never ran: " + logs[0][-400:]
    assert not [x for x in re.findall(r"compactions=(\d+)", logs[1]) if int(x) > 0]
    a, b = films
    assert np.all(a[..., 3] == 6) and np.all(b[..., 3] == 6)
    assert rays[0] == rays[1], rays                                       # the same paths, ray for ray
    assert np.allclose(a[..., :3], b[..., :3], rtol=2e-5, atol=1e-5)      # the same samples, summed in another order


def test_facade_getPixelsColor_runs_on_the_device(pkg, tmp_path):
    """Round 4 (the reference's own loop, main.cpp:26-33: render(scene); getPixelsColor(); every frame): while the whole film is on the device
    Scene::getPixelsColor hands out the DEVICE's tonemap of it (mcpt_tonemap_map: kernel + 3 B per pixel into pinned memory) instead of reading
    16 B per pixel back and running pow on the host.  Checked through the classes: the image after the last frame equals the host path's image
    of the same film (both are float(sqrt(clamp(mean))) * 255.99 truncated: <= 1 LSB apart, Scene.cpp:25-29), the samples are all still there
    afterwards (5 frames -> count 5 everywhere), the film equals five one-sample mcpt_render calls, and a host-side part switches the reader back
    to the folding path."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "monte-carlo-path-tracer_amd", "csrc"); host = os.path.join(root, "monte-carlo-path-tracer_amd", "host")
    exe = str(tmp_path / "facade_pixels")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + host, os.path.join(root, "tests", "facade_pixels.cpp"), os.path.join(csrc, "libmcpt_host.a"), "-o", exe,
                           "-L" + csrc, "-lmcpt_hip", "-lz", "-lpthread", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"])
    scene = pkg.scenes.cornell_box_small(40, 24)
    obj = scene.write(str(tmp_path / "scene"))
    q = lambda a: np.array([[float("%.9g" % x) for x in row] for row in a])
    scene = pkg.scenes.SceneData(scene.name, q(scene.vertex), q(scene.normal), q(scene.texcoord), scene.face, scene.materials, scene.camera)
    frames, depth = 5, 4
    outs = [str(tmp_path / n) for n in ("dev.rgb", "host.rgb", "film.bin")]
    line = subprocess.check_output([exe, obj, str(frames), str(depth)] + outs, timeout=300).decode().split("\n")[-2].split()   # (the loader prints "[Model] <path>" first, like the reference)
    assert line == ["40", "24", str(frames + 1)]
    dev = np.fromfile(outs[0], np.uint8).reshape(24, 40, 3).astype(int); hst = np.fromfile(outs[1], np.uint8).reshape(24, 40, 3).astype(int)
    film = np.fromfile(outs[2], np.float32).reshape(24, 40, 4)
    assert np.all(film[..., 3] == frames)
    assert np.abs(dev - hst).max() <= 1 and (dev != hst).mean() < 0.02
    r = pkg.Renderer(scene, max_depth=depth, flags=pkg.FLAG_CORRECT_SHADOW_T2)
    for f in range(frames): r.render(1, seed=21, first_sample=f)         # consecutive known-length calls: enqueued without waiting for each other
    want = r.read_accum()
    tm, tm2 = r.tonemap(), r.tonemap_map(); r.close()
    assert np.allclose(film, want, rtol=2e-5, atol=1e-5)
    assert np.array_equal(tm, tm2) and np.abs(tm.astype(int) - dev).max() <= 1
    m = np.clip(want[..., :3] / want[..., 3:], 0, 1)
    assert np.abs((np.sqrt(m) * 255.99).astype(np.uint8).astype(int) - dev).max() <= 1


def test_two_triangle_scene_and_explicit_item_sizes(pkg, orc):
    """Smallest scene the builders accept (one light quad = a single leaf under an artificial root), with and without the device
    BVH flag (which falls back to the host path for <= 2 triangles), and explicit samples_per_item values around the automatic one."""
    base = pkg.scenes.open_box(24, 24)
    light = pkg.scenes.SceneData("light-only", base.vertex, base.normal, base.texcoord, base.face[-2:], base.materials, base.camera)
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    cpu, _, _ = orc.Oracle(light, max_depth=3, flags=flags).render(8, seed=2)
    for fl in (flags, flags | pkg.FLAG_GPU_BVH_BUILD):
        r = pkg.Renderer(light, max_depth=3, flags=fl); r.render(8, seed=2); g = r.read_accum(); r.close()
        assert np.all(g[..., 3] == 8) and _frac_beyond(g[..., :3] / 8, cpu[..., :3] / 8) <= 0.01
    scene = pkg.scenes.cornell_box_small(40, 24)
    ref = None
    for spi in (0, 1, 3, 8, 64):
        r = pkg.Renderer(scene, max_depth=4, flags=flags, samples_per_item=spi); r.render(24, seed=5); g = r.read_accum(); r.close()
        assert np.all(g[..., 3] == 24)
        if ref is None: ref = g
        else: assert np.allclose(g, ref, rtol=1e-4, atol=1e-4), spi        # same samples, different summation order


def test_exact_ties_have_a_defined_winner(pkg):
    """Eight coincident copies of the floor, each with its own colour: every floor hit is an eight-way EXACT tie.  The reference lets the
    first triangle in its traversal order win (t < t2 strict, Triangle.cpp:66 / SURVEY A-4); the 8-wide trace kernel tests a ray's leaf
    groups in an order that depends on when its wave ran the leaf block, so it picks the lowest leaf-order index among equal distances
    (tri_accept_closest_tie) -- the winner is a function of the ray, not of the schedule: two renders of the same samples agree, however
    the samples are split over calls, and the floor has ONE colour per triangle pair (no salt-and-pepper mix of the eight)."""
    scene = pkg.scenes.tie_floor(96, 96)
    r = pkg.Renderer(scene, max_depth=4, flags=pkg.FLAG_CORRECT_SHADOW_T2)
    r.render(16, seed=5); a = r.read_accum()
    r.clear(); r.render(8, seed=5, first_sample=0); r.render(8, seed=5, first_sample=8); b = r.read_accum()
    r.clear(); r.render(16, seed=5); c = r.read_accum(); r.close()
    assert np.all(a[..., 3] == 16) and np.isfinite(a).all()
    assert np.allclose(a, b, rtol=1e-4, atol=1e-4) and np.allclose(a, c, rtol=1e-4, atol=1e-4)


def test_exact_ties_follow_the_reference_order_when_asked(pkg):
    """MCPT_FLAG_REFERENCE_TIE_ORDER: the production trace kernel names the SAME face as the real reference's BVH::hit on every ray into the
    eight coincident floors (tests/golden/ref_ties.npz, generated by make_golden.py from oracle/_ref) -- the winner of an exact tie is the
    copy that comes first in the reference's own BVH::triangles order (BVH.cpp:15-54, :95-113), whatever tree this library built.  Without
    the flag the rule is this library's own (lowest leaf-order index): still ONE winner per floor triangle, generally another copy."""
    with np.load(os.path.join(G, "ref_ties.npz")) as z:
        o, d, tri, t, nf = z["ray_o"], z["ray_d"], z["tri"], z["t"], int(z["n_face"])
    scene = pkg.scenes.tie_floor(96, 96)
    assert len(scene.face) == nf
    floor = np.isin(tri, np.r_[0:2, nf - 14:nf])
    assert floor.sum() > 1000
    for fl in (0, pkg.FLAG_GPU_BVH_BUILD):
        r = pkg.Renderer(scene, max_depth=4, flags=pkg.FLAG_REFERENCE_TIE_ORDER | fl)
        t4, tri4, _, _ = r.probe_trace4(o, d); r.close()
        assert np.array_equal(tri4[floor], tri[floor])                                   # exact ties: the reference's winner, ray by ray
        assert (tri4 == tri).mean() >= 0.999 and np.allclose(t4[tri4 == tri], t[tri4 == tri], rtol=2e-5, atol=2e-6)
    r = pkg.Renderer(scene, max_depth=4); _, own, _, _ = r.probe_trace4(o, d); r.close()
    first_tri = np.isin(own[floor], np.r_[0, nf - 14:nf:2])                                # copies of the floor's first / second triangle
    assert len(set(own[floor][first_tri])) == 1 and len(set(own[floor][~first_tri])) == 1 and np.isin(own[floor], np.r_[0:2, nf - 14:nf]).all()


# ------------------------------------------------------------------------------------------------ the bench configurations themselves
@pytest.mark.parametrize("tag", ["c2", "c3", "c4s", "c4", "c5w"])
def test_full_size_block_statistics_vs_reference(pkg, tag):
    """The BENCH configurations at their own resolution against the REAL reference (tests/golden/ref_fullsize_<tag>.npz, written by
    make_golden.py from oracle/_ref: 8 batches x 16 spp = 128 spp -- c4 / c5w: 8 x 8 = 64 spp --, as 8x8-pixel block means with their
    batch-to-batch variance): c2 = S-cornell 800x800 depth 8, c3 = S-veach 1280x720 unbounded, c4s = S-bath (93 k triangles) 1920x1080
    unbounded; round 4: c4 = BASELINE configs[3] with its REAL triangle count (S-bath 0.59 M triangles, 1920x1080, unbounded) and c5w = the
    4.05 M-triangle scene of configs[4] at depth 16 through a 480x270 film of the same view (the reference's regex OBJ parser and its
    midpoint BVH over 4 M triangles: paid once, in the build container).  The GPU renders 32 batches x 32 spp, so the statistic
    z = |difference of block means| / sqrt(var_gpu + var_ref) is carried by the reference's variance estimate from 8 batches: Student t with
    7 degrees of freedom -- 0.52 % of blocks beyond 4 and a median |z| of 0.71 are what IDENTICAL renderers give.  Asserted: image mean
    within 1 %, at most 0.8 % of blocks beyond 4 (r03: 1 %; measured 0.26 - 0.41 %), median |z| in [0.6, 0.85]."""
    path = os.path.join(G, "ref_fullsize_%s.npz" % tag)
    if not os.path.exists(path): pytest.skip(path + " not generated")
    g = _npz("ref_fullsize_%s.npz" % tag)
    name, kw, (w, h) = {"c2": ("cornell-box", {}, (800, 800)), "c3": ("veach-mis", {}, (1280, 720)), "c4s": ("bathroom2", {"detail": 64}, (1920, 1080)),
                        "c4": ("bathroom2", {"detail": 160}, (1920, 1080)), "c5w": ("bathroom2", {"detail": 420}, (480, 270))}[tag]
    r = pkg.Renderer(pkg.scenes.SCENES[name](w, h, **kw), max_depth=int(g["depth"]))
    B, S, b = 32, 32, int(g["block"])
    bm = []
    for k in range(B):
        r.clear(); r.render(S, seed=777, first_sample=k * S); a = r.read_accum(); m = a[..., :3] / a[..., 3:]
        bm.append(m[:h // b * b, :w // b * b].reshape(h // b, b, w // b, b, 3).mean((1, 3)))
    r.close()
    bm = np.stack(bm); gm, gv = bm.mean(0), bm.var(0, ddof=1) / B
    rm, rv = g["mean"], g["var"]
    z = np.abs(gm - rm) / np.sqrt(gv + rv + 1e-14)
    f4, f3, med = float((z > 4).mean()), float((z > 3).mean()), float(np.median(z))
    rel = (gm.mean((0, 1)) - rm.mean((0, 1))) / rm.mean((0, 1))
    print("%s: image mean gpu %s reference %s (rel %s); blocks beyond 4 sigma %.3f %%, beyond 3 sigma %.3f %% (t7: 0.52 / 1.99), median |z| %.3f (t7: 0.711)" % (
        tag, gm.mean((0, 1)), rm.mean((0, 1)), rel, 100 * f4, 100 * f3, med))
    assert np.all(np.abs(rel) <= 0.01)
    assert f4 <= 0.008 and 0.6 <= med <= 0.85


def test_c1_own_size_vs_oracle(pkg, orc):
    """BASELINE.json configs[0] at its own size -- cornell-box 256x256, 16 spp, depth 4 -- same seed against the fp64 oracle (the
    config is the reference's CPU plumbing case; here it pins the GPU path at that size): >= 99.9 % of pixels within 1e-4 * max(1, mean)."""
    scene = pkg.scenes.cornell_box(256, 256)
    flags = pkg.FLAG_CORRECT_SHADOW_T2
    r = pkg.Renderer(scene, max_depth=4, flags=flags); r.render(16, seed=11); g = r.read_accum(); c = r.counters(); r.close()
    cpu, oc, _ = orc.Oracle(scene, max_depth=4, flags=flags).render(16, seed=11)
    assert np.all(g[..., 3] == 16)
    frac = _frac_beyond(g[..., :3] / 16, cpu[..., :3] / 16)
    print("C1 256x256x16 spp depth 4: pixels beyond tolerance %.3f %%" % (100 * frac))
    assert frac <= 0.001
    assert abs(c.rays - (oc["rays_primary"] + oc["rays_continuation"] + oc["rays_shadow"])) <= 0.01 * c.rays


def test_clone_to_device_shares_nothing_but_the_scene(pkg):
    """mcpt_clone_to_device (what `mcpt_cli --gpus N` uses for devices 1 .. N-1; here onto the same device): the clone renders the same
    film bit for bit (deterministic mode) without a BVH build of its own, and the two contexts' films, counters and streams are independent.
    mcpt_tonemap_buffer of a context's own film is mcpt_tonemap."""
    scene = pkg.scenes.cornell_box_small(48, 40)
    a = pkg.Renderer(scene, max_depth=5, flags=pkg.FLAG_DETERMINISTIC)
    b = a.clone(0)
    ia, ib = a.info(), b.info()
    assert (ib.n_tris, ib.n_nodes, ib.wide_nodes, ib.wide_depth, ib.bvh_depth) == (ia.n_tris, ia.n_nodes, ia.wide_nodes, ia.wide_depth, ia.bvh_depth) and ib.bvh_build_ms == 0.0
    a.render(8, seed=9); fa = a.read_accum()
    assert np.all(b.read_accum() == 0)                                     # the clone's film is its own
    b.render(8, seed=9); fb = b.read_accum()
    assert np.array_equal(fa, fb)
    assert b.counters().paths == 48 * 40 * 8 and a.counters().paths == 48 * 40 * 8
    assert np.array_equal(a.tonemap(), a.tonemap_buffer(a.accum_device_ptr()))
    a.close()
    b.render(8, seed=9, first_sample=8); assert np.all(b.read_accum()[..., 3] == 16)       # the clone outlives its source
    b.close()


def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible():
    """`bench.py --gpus 2` on the nccl (= RCCL) backend: its own launcher starts two ranks, each renders its sample range on its own GPU and the
    fp32 films are summed by one all-reduce inside the timed region -- the path the driver's SCALE run takes.  Skipped on a one-GPU box (the
    gloo rehearsal of the same code path runs on CPU: tests/test_bench_launcher.py); a rank that dies takes the launch down (bounded)."""
    import json, subprocess, sys
    import torch
    if torch.cuda.device_count() < 2: pytest.skip("needs two visible GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MCPT_BENCH_LAUNCH_TIMEOUT"] = "600"
    for shard in ("samples", "tiles"):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--spp", "64", "--no-cpu-baseline", "--shard", shard],
                           env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["film_count_plane_ok"] is True
        assert out["scaling"] == ("strong" if shard == "tiles" else "weak") and out["value"] > 0


def test_path_pools_grow_with_the_job_and_keep_the_film_exact(pkg):
    """The path pools are allocated by the first render call and grown to the largest job seen (mcpt_api.cpp: ensure_pool): a context that
    rendered one sample of a small film holds a small pool; a later, larger job re-allocates it between two calls -- nothing of a job lives in
    the pool across calls, so the film after both equals a fresh context's, and device_bytes says what is held."""
    scene = pkg.scenes.cornell_box_small(160, 120)
    flags = pkg.FLAG_DETERMINISTIC
    r = pkg.Renderer(scene, max_depth=5, flags=flags)
    b0 = r.info().device_bytes
    r.render(1, seed=4); b1 = r.info().device_bytes
    r.render(40, seed=4, first_sample=1); b2 = r.info().device_bytes; grown = r.read_accum()
    r.render(1, seed=4, first_sample=41); b3 = r.info().device_bytes; r.close()       # a smaller job afterwards: the pool stays
    assert b0 < b1 <= b2 == b3 and b1 - b0 < 64 << 20                                 # (one slot per pixel: 160 x 120 x 164 B, not 2.75 GB)
    f = pkg.Renderer(scene, max_depth=5, flags=flags); f.render(41, seed=4); fresh = f.read_accum(); f.close()
    assert np.array_equal(grown[..., 3], fresh[..., 3]) and np.allclose(grown, fresh, rtol=2e-6, atol=1e-6)
