#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref, built by oracle/build_ref.sh from /root/reference).

The reference has no tests, fixtures or golden vectors of its own, and its scene files are absent, so
every pin is produced here: inputs are drawn with fixed numpy seeds, pushed through the reference's own functions via
oracle/ref/ref_driver.cpp (random numbers injected through oracle/ref/ref_shim.h where a function draws any), and the
inputs + outputs are stored.  Only DATA is stored -- no reference source.  Run where /root/reference exists:

    python tests/golden/make_golden.py

Files:
  ref_kats.npz         function-level known-answer vectors (AABB, triangle, BSDF, texture, camera, light sampling, film)
  ref_paths.npz        path-level vectors: (pixel, xi sequence) -> radiance of Render::ray_tracing, both integrators,
                       plus BVH::hit / has_hit on random rays, on the S-cornell-small scene
  ref_images.npz       per-pixel mean / variance images of the reference renderer (64x64) for statistical parity
  ref_scenes2.npz      (round 2) the same path-level pins on two more scenes -- S-veach small (480 light triangles, Blinn-Phong
                       exponents 10..5000) and S-bath small (image textures, mirror Ns = 10000, glossy chrome): injected-xi full
                       paths, BVH::hit / has_hit ray records, light samples, and default-mode mean / variance images
                       (`python tests/golden/make_golden.py scenes2` writes only this file)
  ref_fullsize_<c>.npz (round 3) 8x8-block mean / variance-of-the-mean maps of the real reference at the bench configurations' own sizes
                       (c2: S-cornell 800x800 depth 8; c3: S-veach 1280x720; c4s: S-bath 93 k triangles 1920x1080), 128 spp each;
                       round 4: c4 = S-bath 0.59 M triangles 1920x1080 and c5w = the 4.05 M-triangle S-bath, depth 16, 480x270 film, 64 spp each
                       (`... make_golden.py fullsize c2` etc.: one configuration per process, tens of CPU-minutes each)
  ref_ties.npz         (round 3) BVH::hit of the real reference on rays into eight coincident floors: which face wins an exact tie
                       (`... make_golden.py ties`)
  ref_loader.npz       (round 2) the reference's own Model(filename) parse of tests/golden/loader_quirks/quirk.obj (this project's
                       quirk-exercising input) and of a scenes.py-written S-bath small (`... make_golden.py loader`)
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402

pkg = orc.pkg


def rand_xi(rng, n):
    return (rng.randint(0, 1 << 24, n).astype(np.float32) / np.float32(1 << 24)).astype(np.float32)


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def kats(ref):
    rng = np.random.RandomState(20251004)
    out = {}
    # ---- AABB::Intersection
    n = 3000
    A = rng.uniform(-1, 1, (n, 3)); B = A + rng.uniform(0, 1, (n, 3)) * (rng.rand(n, 3) > 0.1)   # some flat boxes
    o = rng.uniform(-2, 2, (n, 3))
    aim = A + (B - A) * rng.uniform(-0.15, 1.15, (n, 3))                                           # mostly towards the box, some near misses
    d = np.where(rng.rand(n, 1) < 0.75, unit(aim - o), unit(rng.normal(size=(n, 3))))
    d[::17, 0] = 0.0; d[::23, 1] = 0.0                                                            # axis-parallel rays (inf / nan slabs)
    o[::34] = np.where(rng.rand(len(o[::34]), 3) < 0.5, A[::34], o[::34])                          # origins on a slab plane: 0 * inf = nan
    d = unit(d)
    t1 = np.full(n, 1e-4); t2 = np.where(rng.rand(n) < 0.5, np.finfo(np.float64).max, rng.uniform(0.1, 3, n))
    res = np.array([ref.aabb_intersect(A[i], B[i], o[i], d[i], t1[i], t2[i]) for i in range(n)], np.int32)
    out.update(aabb_A=A, aabb_B=B, aabb_o=o, aabb_d=d, aabb_t1=t1, aabb_t2=t2, aabb_hit=res)
    # ---- Triangle::hit / isIntersect / area
    n = 3000
    v = rng.uniform(-1, 1, (n, 9)); v[::11, 3:6] = v[::11, 0:3] + 1e-4 * rng.normal(size=(len(v[::11]), 3))   # slivers -> |a| threshold
    vn = unit(rng.normal(size=(n, 3, 3))).reshape(n, 9); uv = rng.uniform(0, 1, (n, 6))
    tgt = (v.reshape(n, 3, 3) * rng.dirichlet((1, 1, 1), n)[:, :, None]).sum(1) + 0.05 * rng.normal(size=(n, 3)) * (rng.rand(n, 1) < 0.3)
    o = rng.uniform(-2, 2, (n, 3)); d = unit(tgt - o)
    t2 = np.where(rng.rand(n) < 0.6, np.finfo(np.float64).max, rng.uniform(0.1, 4, n))
    em = (rng.rand(n) < 0.3).astype(np.int32)
    hit = np.zeros(n, np.int32); rec = np.zeros((n, 13)); anyh = np.zeros(n, np.int32); area = np.zeros(n, np.float32)
    for i in range(n):
        hit[i], rec[i] = ref.tri_hit(v[i], vn[i], uv[i], em[i], o[i], d[i], 1e-4, t2[i])
        anyh[i] = ref.tri_any(v[i], o[i], d[i], 1e-4, t2[i])
        area[i] = ref.tri_area(v[i])
    out.update(tri_v=v, tri_vn=vn, tri_uv=uv, tri_em=em, tri_o=o, tri_d=d, tri_t2=t2, tri_hit=hit, tri_rec=rec, tri_any=anyh, tri_area=area)
    # ---- BSDF: setup / eval / sample over the three lobe configurations
    n = 1500
    nrm = unit(rng.normal(size=(n, 3))); wi = unit(rng.normal(size=(n, 3))); wo = unit(rng.normal(size=(n, 3))).astype(np.float32)
    kd = rng.uniform(0, 1, (n, 3)); ks = rng.uniform(0, 1, (n, 3)); ns = rng.choice([1.0, 10.0, 50.0, 1000.0, 5000.0, 10000.0, 20000.0], n)
    kind = rng.randint(0, 4, n)
    ks[kind == 0] = 0.0                        # diffuse only
    kd[kind == 3] = 0.0; ks[kind == 3] = 0.0   # black: uninitialised weights in the reference (A-12) -> recorded but not compared
    nrm[::13] = np.array([0.95, 0.1, 0.1]) / np.linalg.norm([0.95, 0.1, 0.1])    # |n.x| > 0.9 branch of the ONB
    xi = np.stack([rand_xi(rng, n), rand_xi(rng, n), rand_xi(rng, n)], 1)
    setup = np.zeros((n, 18), np.float32); ev = np.zeros((n, 4), np.float32); smp = np.zeros((n, 8), np.float32); used = np.zeros(n, np.int32)
    for i in range(n):
        setup[i] = ref.bsdf_setup(nrm[i], wi[i], kd[i], ks[i], ns[i])
        ev[i] = ref.bsdf_eval(nrm[i], wi[i], kd[i], ks[i], ns[i], wo[i])
        smp[i], used[i] = ref.bsdf_sample(nrm[i], wi[i], kd[i], ks[i], ns[i], xi[i])
    out.update(bsdf_n=nrm, bsdf_wi=wi, bsdf_wo=wo, bsdf_kd=kd, bsdf_ks=ks, bsdf_ns=ns, bsdf_kind=kind, bsdf_xi=xi, bsdf_setup=setup,
               bsdf_eval=ev, bsdf_sample=smp, bsdf_used=used)
    # ---- utils
    a = rng.uniform(0, 5, 500).astype(np.float32); b = rng.uniform(0, 5, 500).astype(np.float32); a[::50] = 0; b[::50] = 0
    out.update(ph_a=a, ph_b=b, ph=np.array([ref.power_heuristic(float(x), float(y)) for x, y in zip(a, b)], np.float32))
    c = np.concatenate([rng.uniform(-0.5, 1.5, 200), [0.999, 0.9991, 1.0, 0.0, -0.0]]).astype(np.float32)
    out.update(clamp_in=c, clamp_out=np.array([ref.L.ref_clamp01(float(x)) for x in c]))
    # ---- Texture::get_color on an 8x5 image
    img = rng.uniform(0, 1, (5, 8, 3)).astype(np.float32)
    uvs = np.concatenate([rng.uniform(-2, 3, (300, 2)), [[0.0, 0.0], [1.0, 1.0], [0.9995, 0.9995], [-0.0001, 2.0]]])
    out.update(tex_img=img, tex_uv=uvs, tex_rgb=np.array([ref.texture_get_color(img, u, v) for u, v in uvs], np.float32))
    # ---- film: set_Pixel NaN scrub + getPixelsColor tonemap on the loaded scene's film (whatever its size)
    return out


def scene_vectors(ref, scene, tag):
    rng = np.random.RandomState(777)
    out = {}
    w, h = scene.camera.width, scene.camera.height
    # Render::cast_Ray
    n = 400
    xy = np.stack([rng.randint(0, w, n), rng.randint(0, h, n)], 1).astype(np.int32); xi = np.stack([rand_xi(rng, n), rand_xi(rng, n)], 1)
    od = np.zeros((n, 6))
    for i in range(n):
        od[i], _ = ref.cast_ray(int(xy[i, 0]), int(xy[i, 1]), xi[i])
    out.update({tag + "cam_xy": xy, tag + "cam_xi": xi, tag + "cam_od": od})
    # BVH::hit / has_hit on random rays from inside the box
    n = 4000
    o = rng.uniform(0.05, 0.95, (n, 3)); d = unit(rng.normal(size=(n, 3)))
    rec = np.zeros((n, 12)); hit = np.zeros(n, np.int32); anyh = np.zeros(n, np.int32); t2 = rng.uniform(0.05, 1.5, n)
    for i in range(n):
        hit[i], rec[i] = ref.bvh_hit(o[i], d[i])
        anyh[i] = ref.bvh_has_hit(o[i], d[i], 1e-4, t2[i])
    out.update({tag + "ray_o": o, tag + "ray_d": d, tag + "ray_hit": hit, tag + "ray_rec": rec, tag + "ray_t2": t2, tag + "ray_any": anyh})
    # Render::sample from points inside the box
    n = 600
    p = rng.uniform(0.05, 0.9, (n, 3)); xi = np.stack([rand_xi(rng, n) for _ in range(3)], 1)
    ls = np.zeros((n, 14))
    for i in range(n):
        ls[i], _ = ref.sample_light(p[i], xi[i])
    out.update({tag + "ls_p": p, tag + "ls_xi": xi, tag + "ls_out": ls})
    # full paths through Render::ray_tracing(Ray&): pixel + injected xi sequence -> radiance
    n, m = 3000, 192
    xy = np.stack([rng.randint(0, w, n), rng.randint(0, h, n)], 1).astype(np.int32)
    xi = np.stack([rand_xi(rng, m) for _ in range(n)], 0)
    L = np.zeros((n, 3), np.float32); used = np.zeros(n, np.int32)
    for i in range(n):
        L[i], used[i] = ref.trace_pixel(int(xy[i, 0]), int(xy[i, 1]), xi[i])
    assert ref.underflow() == 0 and used.max() < m, "xi budget too small"
    keep = used.max() + 2
    out.update({tag + "path_xy": xy, tag + "path_xi": xi[:, :keep].copy(), tag + "path_L": L, tag + "path_used": used})
    # the dead recursive integrator Render::ray_tracing(Ray&,int)
    n = 800
    o = rng.uniform(0.1, 0.9, (n, 3)); d = unit(rng.normal(size=(n, 3))); xi = np.stack([rand_xi(rng, m) for _ in range(n)], 0)
    L = np.zeros((n, 3), np.float32); used = np.zeros(n, np.int32)
    for i in range(n):
        L[i], used[i] = ref.trace_path(o[i], d[i], xi[i], recursive=True)
    assert used.max() < m
    out.update({tag + "rec_o": o, tag + "rec_d": d, tag + "rec_xi": xi[:, :used.max() + 2].copy(), tag + "rec_L": L, tag + "rec_used": used})
    # light self-occlusion (SURVEY A-9): fraction of light samples whose shadow ray the sampled triangle itself blocks
    n = 4000
    o = rng.uniform(0.1, 0.9, (n, 3)); d = unit(rng.normal(size=(n, 3))); xi = np.stack([rand_xi(rng, n) for _ in range(3)], 1)
    ok = np.zeros(n, np.int32); res = np.zeros((n, 2), np.int32)
    for i in range(n):
        ok[i], res[i] = ref.shadow_probe(o[i], d[i], xi[i], 1e-4)
    out.update({tag + "sp_o": o, tag + "sp_d": d, tag + "sp_xi": xi, tag + "sp_ok": ok, tag + "sp_res": res})
    return out


def film_vectors(ref):
    rng = np.random.RandomState(5)
    ref.clear()
    w, h = ref.width, ref.height
    vals = rng.uniform(0, 2, (h, w, 3)).astype(np.float32)
    vals[0, 0] = [np.nan, 0.5, 0.25]; vals[1, 1] = [0.3, np.nan, np.nan]; vals[2, 2] = [np.inf, 0.1, 0.2]
    for rep in range(3):
        for y in range(h):
            for x in range(w):
                ref.set_pixel(x, y, vals[y, x] * (rep + 1))
    return {"film_vals": vals, "film_accum": ref.accum(), "film_u8": ref.pixels_u8()}


def image_stats(scene, lib_depth, max_bounces, frames, batches):
    """Mean and variance-of-the-mean images from `batches` independent batches of `frames` spp of the real reference."""
    ref = orc.Reference(depth_variant=lib_depth)
    tmp = tempfile.mkdtemp(prefix="mcpt_golden_")
    ref.load(scene.write(tmp))
    ref.stream_mode()
    if max_bounces:
        ref.set_max_bounces(max_bounces)
    means = []
    for b in range(batches):
        ref.clear(); ref.render(frames)
        a = ref.accum(); means.append(a[..., :3] / a[..., 3:])
    means = np.stack(means)
    return means.mean(0).astype(np.float32), (means.var(0, ddof=1) / batches).astype(np.float32)



# ---- full-size block statistics (round 3): the BENCH configurations themselves, pinned to the real reference ------------------------
FULLSIZE = {   # tag: (generator, kwargs, (w, h), depth limit (0 = the reference's unbounded loop), batches, frames per batch)
    "c2": ("cornell-box", {}, (800, 800), 8, 8, 16),
    "c3": ("veach-mis", {}, (1280, 720), 0, 8, 16),
    "c4s": ("bathroom2", {"detail": 64}, (1920, 1080), 0, 8, 16),
    # round 4: the two large configurations with their REAL triangle counts (the reference's regex OBJ parser takes minutes on these: paid once, here)
    "c4": ("bathroom2", {"detail": 160}, (1920, 1080), 0, 8, 8),          # BASELINE configs[3]: 0.59 M triangles, 64 spp
    "c5w": ("bathroom2", {"detail": 420}, (480, 270), 16, 8, 8),          # BASELINE configs[4]'s 4.05 M-triangle scene, depth 16, through a 480x270 film of the same view
}


def block_means(img, b=8):
    h, w = img.shape[0] // b * b, img.shape[1] // b * b
    return img[:h, :w].reshape(h // b, b, w // b, b, img.shape[2]).mean((1, 3))


def fullsize(tag):
    """8 x 8-pixel block means of the real reference at a bench configuration's own resolution: mean and variance-of-the-mean over
    `batches` independent batches (single pixels fed by a hundred heavy-tailed samples are far from normal; block means are not)."""
    name, kw, (w, h), depth, batches, frames = FULLSIZE[tag]
    scene = pkg.scenes.SCENES[name](w, h, **kw)
    ref = orc.Reference(depth_variant=depth > 0)
    ref.load(scene.write(tempfile.mkdtemp(prefix="mcpt_golden_")))
    ref.stream_mode()
    if depth: ref.set_max_bounces(depth)
    import time
    t0 = time.time(); bm = []
    for b in range(batches):
        ref.clear(); ref.render(frames)
        a = ref.accum(); bm.append(block_means(a[..., :3] / a[..., 3:]))
        print("[fullsize %s] batch %d / %d  (%.0f s)" % (tag, b + 1, batches, time.time() - t0), flush=True)
    bm = np.stack(bm)
    np.savez_compressed(os.path.join(HERE, "ref_fullsize_%s.npz" % tag), mean=bm.mean(0).astype(np.float32), var=(bm.var(0, ddof=1) / batches).astype(np.float32),
                        spp=np.int32(batches * frames), batches=np.int32(batches), block=np.int32(8), depth=np.int32(depth),
                        image_mean=bm.mean((0, 1, 2)).astype(np.float64))
    print("[fullsize %s] image mean %s" % (tag, bm.mean((0, 1, 2))))


SCENES2 = {   # tag: (generator, kwargs, (w, h), box the probe rays / shading points are drawn from)
    "vm_": ("veach-mis", {"light_lon": 12, "light_lat": 6, "plate_cells": 4}, (64, 36), ((-6.0, 0.0, -5.0), (6.0, 6.0, 8.0))),
    "bt_": ("bathroom2", {"detail": 12, "tex_size": 32}, (64, 36), ((0.1, 0.1, 0.1), (3.9, 2.5, 4.9))),
}


def scene2_vectors(ref, scene, tag, box):
    """Path-level pins on one more scene (same record layouts as scene_vectors)."""
    rng = np.random.RandomState(4242)
    out = {}
    w, h = scene.camera.width, scene.camera.height
    lo, hi = np.array(box[0]), np.array(box[1])
    n = 2000
    o = rng.uniform(lo, hi, (n, 3)); d = unit(rng.normal(size=(n, 3)))
    rec = np.zeros((n, 12)); hit = np.zeros(n, np.int32); anyh = np.zeros(n, np.int32); t2 = rng.uniform(0.05, 0.6, n) * np.linalg.norm(hi - lo)
    for i in range(n):
        hit[i], rec[i] = ref.bvh_hit(o[i], d[i])
        anyh[i] = ref.bvh_has_hit(o[i], d[i], 1e-4, t2[i])
    out.update({tag + "ray_o": o, tag + "ray_d": d, tag + "ray_hit": hit, tag + "ray_rec": rec, tag + "ray_t2": t2, tag + "ray_any": anyh})
    n = 400
    p = rng.uniform(lo, hi, (n, 3)); xi = np.stack([rand_xi(rng, n) for _ in range(3)], 1)
    ls = np.zeros((n, 14))
    for i in range(n):
        ls[i], _ = ref.sample_light(p[i], xi[i])
    out.update({tag + "ls_p": p, tag + "ls_xi": xi, tag + "ls_out": ls})
    n, m = 1200, 320
    xy = np.stack([rng.randint(0, w, n), rng.randint(0, h, n)], 1).astype(np.int32)
    xi = np.stack([rand_xi(rng, m) for _ in range(n)], 0)
    L = np.zeros((n, 3), np.float32); used = np.zeros(n, np.int32)
    for i in range(n):
        L[i], used[i] = ref.trace_pixel(int(xy[i, 0]), int(xy[i, 1]), xi[i])
    assert ref.underflow() == 0 and used.max() < m, "xi budget too small"
    out.update({tag + "path_xy": xy, tag + "path_xi": xi[:, :used.max() + 2].copy(), tag + "path_L": L, tag + "path_used": used})
    st = ref.bvh_stats()
    out[tag + "bvh"] = np.array([st["nodes"], st["leaves"], st["depth"], st["max_leaf"], ref.num_tris(), ref.num_lights()], np.int64)
    return out


def scenes2():
    out = {}
    for tag, (name, kw, res, box) in SCENES2.items():
        scene = pkg.scenes.SCENES[name](res[0], res[1], **kw)
        tmp = tempfile.mkdtemp(prefix="mcpt_golden_")
        ref = orc.Reference()
        ref.load(scene.write(tmp))
        out.update(scene2_vectors(ref, scene, tag, box))
        m, v = image_stats(scene, False, 0, 64, 16)
        out.update({tag + "unbounded_mean": m, tag + "unbounded_var": v})
        print(tag, "done: tris", ref.num_tris(), "lights", ref.num_lights(), "image mean", m.mean((0, 1)))
    np.savez_compressed(os.path.join(HERE, "ref_scenes2.npz"), **out)


def loader():
    """ref_loader.npz: what the reference's OWN Model(filename) (model.cpp:44-281) parses from (a) tests/golden/loader_quirks/quirk.obj
    -- this project's test input exercising the parser's quirks: a 4-corner face, a/b/c with b != c, text after the numbers of a `v`
    line, a line starting with a blank, `#` in the middle of an MTL line, a material without Kd, an unknown usemtl, a PNG map_Kd, a
    <light> for a material that does not exist -- and (b) the OBJ/MTL/XML/PPM files scenes.py writes for S-bath small."""
    out = {}
    ref = orc.Reference()
    cases = {"quirk_": os.path.join(HERE, "loader_quirks", "quirk.obj"),
             "jpeg_": os.path.join(HERE, "loader_quirks", "jpeg.obj"),          # map_Kd = a baseline 4:2:0 JPEG: stb_image's decode is the reference here
             "order_": os.path.join(HERE, "loader_quirks", "order.obj"),        # usemtl before its mtllib line: the reference resolves usemtl after reading the whole file
             "bath_": pkg.scenes.bathroom_stress(64, 36, detail=12, tex_size=32).write(tempfile.mkdtemp(prefix="mcpt_golden_"))}
    for tag, path in cases.items():
        m = ref.parse_model(path)
        for k in ("vertex", "normal", "texcoord", "face", "materials", "camera"):
            out[tag + k] = m[k]
        out[tag + "size"] = np.array([m["width"], m["height"]], np.int32)
        for i, t in enumerate(m["texels"]):
            out[tag + "texels%d" % i] = t
    np.savez_compressed(os.path.join(HERE, "ref_loader.npz"), **out)
    print("loader golden written:", {k: v.shape for k, v in out.items() if k.startswith("quirk_")})


def ties():
    """(round 3) BVH::hit of the real reference on rays that hit the eight coincident floors of scenes.tie_floor: which face wins an EXACT
    eight-way tie is decided by the reference's own triangle order after BVH::build (BVH.cpp:15-54, :95-113) -- the known answers for
    MCPT_FLAG_REFERENCE_TIE_ORDER."""
    scene = pkg.scenes.tie_floor(96, 96)
    ref = orc.Reference()
    ref.load(scene.write(tempfile.mkdtemp(prefix="mcpt_golden_")))
    rng = np.random.RandomState(20251005)
    n = 4000
    V = np.asarray(scene.vertex, np.float64); F = np.asarray(scene.face)
    fl = F[:2, :, 0]                                                        # the floor's two triangles (vertex indices)
    # targets: uniformly inside the floor's triangles; origins: anywhere above, inside the box
    pick = rng.randint(0, 2, n); b = rng.rand(n, 2); flip = b.sum(1) > 1; b[flip] = 1 - b[flip]
    tv = V[fl[pick]]                                                        # (n, 3 corners, 3)
    target = tv[:, 0] + b[:, :1] * (tv[:, 1] - tv[:, 0]) + b[:, 1:] * (tv[:, 2] - tv[:, 0])
    lo, hi = V.min(0), V.max(0)
    up = np.argmax(np.abs(np.cross(tv[0, 1] - tv[0, 0], tv[0, 2] - tv[0, 0])))   # the floor's normal axis
    origin = lo + (hi - lo) * rng.uniform(0.1, 0.9, (n, 3)); origin[:, up] = lo[up] + (hi[up] - lo[up]) * rng.uniform(0.3, 0.9, n)
    d = unit(target - origin)
    tri = np.zeros(n, np.int32); t = np.zeros(n)
    for i in range(n):
        h, out = ref.bvh_hit(origin[i], d[i]); tri[i] = int(out[11]) if h else -1; t[i] = out[0]
    floor_faces = set(range(2)) | set(range(len(F) - 14, len(F)))
    on_floor = np.array([int(x) in floor_faces for x in tri])
    print("ties: %d rays, %d end on a floor copy; winners:" % (n, on_floor.sum()), dict(zip(*np.unique(tri[on_floor], return_counts=True))))
    np.savez_compressed(os.path.join(HERE, "ref_ties.npz"), ray_o=origin, ray_d=d, tri=tri, t=t, n_face=np.int32(len(F)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "ties":
        ties(); return
    os.environ["OMP_NUM_THREADS"] = "1"     # the reference's global mt19937 is racy across threads; single-threaded = reproducible
    if len(sys.argv) > 1 and sys.argv[1] == "scenes2":
        scenes2(); return
    if len(sys.argv) > 1 and sys.argv[1] == "loader":
        loader(); return
    if len(sys.argv) > 2 and sys.argv[1] == "fullsize":                    # one configuration per process (tens of CPU-minutes each): c2 | c3 | c4s | c4 | c5w
        fullsize(sys.argv[2]); return
    scene = pkg.scenes.cornell_box_small(64, 64)
    tmp = tempfile.mkdtemp(prefix="mcpt_golden_")
    ref = orc.Reference()
    ref.load(scene.write(tmp))
    k = kats(ref)
    k.update(film_vectors(ref))
    np.savez_compressed(os.path.join(HERE, "ref_kats.npz"), **k)
    p = scene_vectors(ref, scene, "cs_")
    st = ref.bvh_stats()
    p.update(cs_bvh=np.array([st["nodes"], st["leaves"], st["depth"], st["max_leaf"], ref.num_tris(), ref.num_lights()], np.int64))
    np.savez_compressed(os.path.join(HERE, "ref_paths.npz"), **p)
    print("kats + paths written")
    # images need fresh Reference objects per scene (the driver holds one scene); ctypes loads the same .so once, so run
    # each in a subprocess-free way: Reference() reloads via ref_load.
    imgs = {}
    m, v = image_stats(scene, False, 0, 64, 16); imgs.update(cs_unbounded_mean=m, cs_unbounded_var=v)
    m, v = image_stats(scene, True, 4, 64, 16); imgs.update(cs_depth4_mean=m, cs_depth4_var=v)
    ob = pkg.scenes.open_box(48, 48)
    m, v = image_stats(ob, False, 0, 64, 16); imgs.update(ob_unbounded_mean=m, ob_unbounded_var=v)
    np.savez_compressed(os.path.join(HERE, "ref_images.npz"), **imgs)
    print("images written")
    scenes2()
    loader()


if __name__ == "__main__":
    main()
