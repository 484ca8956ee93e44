"""CPU tests (no GPU): the oracle against the REAL reference on two more scenes than round 1 pinned (tests/golden/ref_scenes2.npz,
written by tests/golden/make_golden.py from oracle/_ref): S-veach small -- 480 light triangles on four spheres, Blinn-Phong
exponents 10 / 100 / 1000 / 5000 -- and S-bath small -- image textures (Texture::get_color inside a path), a mirror (Ns = 10000:
the specular_reflection lobe and its MIS bypass, Render.cpp:148-151), glossy chrome (Ns = 2000).  Same layers as
test_oracle_vs_reference.py: BVH::hit / has_hit records, Render::sample, full injected-xi paths of Render::ray_tracing (bit for
bit up to libm ulps), and the COUNTER-mode renderer against the reference's per-pixel statistics in its default (A-9) mode."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES2 = {"vm_": ("veach-mis", {"light_lon": 12, "light_lat": 6, "plate_cells": 4}, (64, 36)),
           "bt_": ("bathroom2", {"detail": 12, "tex_size": 32}, (64, 36))}


@pytest.fixture(scope="module")
def g2():
    with np.load(os.path.join(G, "ref_scenes2.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module", params=sorted(SCENES2))
def case(request, pkg, orc):
    name, kw, res = SCENES2[request.param]
    scene = pkg.scenes.SCENES[name](res[0], res[1], **kw)
    return request.param, scene, orc.Oracle(scene)


def test_bvh_is_the_reference_tree(case, g2):               # BVH.cpp:15-54
    tag, _, o = case
    nodes, leaves, depth, max_leaf, tris, lights = g2[tag + "bvh"]
    i = o.info()
    assert (i["nodes"], i["leaves"], i["depth"], i["tris"], i["lights"]) == (nodes, leaves, depth, tris, lights)


def test_bvh_hit_and_has_hit(case, g2):                     # BVH.cpp:90-136
    tag, _, o = case
    hit, rec, anyh = g2[tag + "ray_hit"], g2[tag + "ray_rec"], g2[tag + "ray_any"]
    for i in range(len(hit)):
        h, r = o.bvh_hit(g2[tag + "ray_o"][i], g2[tag + "ray_d"][i])
        assert h == hit[i]
        if h:
            assert np.array_equal(r, rec[i])                 # t, point, normal, uv, front, lightarea, triangle index
        assert o.bvh_has_hit(g2[tag + "ray_o"][i], g2[tag + "ray_d"][i], 1e-4, g2[tag + "ray_t2"][i]) == anyh[i]
    assert hit.mean() > 0.3 and 0.02 < anyh.mean() < 0.98


def test_sample_light(case, g2):                            # Render.cpp:202-223 (480 lights: the pick `min(int(xi n), n-1)` matters)
    tag, _, o = case
    for pnt, xi, out in zip(g2[tag + "ls_p"], g2[tag + "ls_xi"], g2[tag + "ls_out"]):
        got, used = o.sample_light(pnt, xi)
        assert used == 3 and np.array_equal(got, out)


def test_full_paths_mis_integrator(case, g2):               # Render.cpp:111-175 with textures / mirror / many lights inside the path
    tag, _, o = case
    worst = 0.0; same_used = 0
    n = len(g2[tag + "path_L"])
    for i in range(n):
        L, used = o.trace_pixel(int(g2[tag + "path_xy"][i, 0]), int(g2[tag + "path_xy"][i, 1]), g2[tag + "path_xi"][i])
        # Both scenes have BLACK emitters (Kd = Ks = 0: the window, the sphere lights).  There the reference samples a lobe with
        # UNINITIALISED weights (BSDF.cpp:179-180, SURVEY A-12) and walks on with a throughput of zero, drawing random numbers that
        # can no longer change the radiance; the oracle (and the device) define that case as the end of the path.  So the radiance must
        # agree on EVERY path, the number of random numbers consumed only where no black surface was hit.
        assert used <= g2[tag + "path_used"][i]
        same_used += int(used == g2[tag + "path_used"][i])
        ref = g2[tag + "path_L"][i]
        worst = max(worst, float(np.abs(L - ref).max() / max(1e-6, np.abs(ref).max())))
    assert worst <= 1e-6, worst
    assert same_used >= 0.85 * n, same_used
    assert g2[tag + "path_used"].max() > 30 and (g2[tag + "path_L"].sum(1) > 0).mean() > 0.2    # paths into the Russian-roulette regime (bounces > 3)


def test_counter_mode_matches_reference_statistics(case, g2):
    """Default (reference-faithful, A-9) mode: oracle images on the shared counter-based generator vs the reference's mt19937 images."""
    tag, scene, o = case
    means = []
    for b in range(16):
        acc, _, _ = o.render(64, seed=99, first_sample=b * 64)
        means.append(acc[..., :3] / acc[..., 3:])
    m = np.stack(means); mean, var = m.mean(0), m.var(0, ddof=1) / 16
    rm, rv = g2[tag + "unbounded_mean"], g2[tag + "unbounded_var"]
    # image mean: within 4 standard errors of the difference (S-veach's 3-cm lights make it a high-variance picture: one standard error
    # of its image mean is ~2 % at 1024 spp, so the fixed 1 % of the Cornell tests would be a coin flip here) and within 5 % outright
    npix = rm.shape[0] * rm.shape[1]
    se = np.sqrt(var.sum((0, 1)) + rv.sum((0, 1))) / npix
    dm = np.abs(mean.mean((0, 1)) - rm.mean((0, 1)))
    assert np.all(dm <= 4 * se) and np.all(dm <= 0.05 * rm.mean((0, 1))), (mean.mean((0, 1)), rm.mean((0, 1)), se)
    z = np.abs(mean - rm) / np.sqrt(var + rv + 1e-12)
    frac = float((z > 4).mean())
    print(tag, "oracle", mean.mean((0, 1)), "reference", rm.mean((0, 1)), "pixels > 4 sigma %.3f%%" % (100 * frac))
    assert frac <= 0.01
