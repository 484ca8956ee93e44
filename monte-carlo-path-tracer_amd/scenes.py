"""Synthetic stand-ins for the cg24 scenes the reference renders (cornell-box, veach-mis, bathroom2).

The real scene folders are git-ignored in the reference (``.gitignore:3``, ``src/main.cpp:7-12``) and are
not available offline, so every test / bench / golden vector uses the generators below.
A scene is produced in the reference's own ``Model`` layout (``src/model.h:51-60``): indexed ``vertex`` /
``normal`` / ``texture`` arrays in fp64, ``face`` = imat3x4 rows ``[v, vn, vt, material]`` per corner,
``materials`` and ``camerainfo`` -- and can be written as the OBJ + MTL + XML triple the reference's
``Model(filename)`` parses (``src/model.cpp:44-281``).

All numbers are first formatted as text and then parsed back, so the arrays handed to the HIP library are
bit-identical to what the reference's ``stringstream >> double`` / ``stod`` produce from the files.
Every face corner uses equal v/vt/vn indices (the reference swaps vt and vn).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


def _q(x: float) -> float:
    """Round-trip through the text form used in the files (9 significant digits)."""
    return float("%.9g" % x)


@dataclass
class Material:
    name: str
    kd: tuple = (0.0, 0.0, 0.0)
    ks: tuple = (0.0, 0.0, 0.0)
    ns: float = 1.0
    radiance: tuple = (0.0, 0.0, 0.0)     # from the XML <light mtlname radiance> (model.cpp:264-279)
    map_kd: Optional[str] = None          # texture file name (relative to the mtl), if any
    texture: Optional[np.ndarray] = None  # HxWx3 uint8 image behind map_kd


@dataclass
class Camera:
    eye: tuple
    lookat: tuple
    up: tuple
    fovy: float
    width: int
    height: int


@dataclass
class SceneData:
    name: str
    vertex: np.ndarray      # (nv,3) f64
    normal: np.ndarray      # (nn,3) f64
    texcoord: np.ndarray    # (nt,2) f64
    face: np.ndarray        # (nf,3,4) int32: per corner [v, vn, vt, material]  (glm::imat3x4, model.h:57)
    materials: List[Material]
    camera: Camera
    meta: Dict = field(default_factory=dict)

    @property
    def n_faces(self) -> int:
        return int(self.face.shape[0])

    def with_resolution(self, w: int, h: int) -> "SceneData":
        cam = Camera(self.camera.eye, self.camera.lookat, self.camera.up, self.camera.fovy, int(w), int(h))
        return SceneData(self.name, self.vertex, self.normal, self.texcoord, self.face, self.materials, cam, dict(self.meta))

    # ------------------------------------------------------------------ file output
    def write(self, directory: str) -> str:
        """Write <name>.obj / .mtl / .xml (+ textures as binary PPM) and return the .obj path."""
        os.makedirs(directory, exist_ok=True)
        obj = os.path.join(directory, self.name + ".obj")
        with open(obj, "w") as f:
            f.write("mtllib %s.mtl\n" % self.name)
            for v in self.vertex:
                f.write("v %.9g %.9g %.9g\n" % tuple(v))
            for n in self.normal:
                f.write("vn %.9g %.9g %.9g\n" % tuple(n))
            for t in self.texcoord:
                f.write("vt %.9g %.9g\n" % tuple(t))
            cur = -1
            for fc in self.face:
                m = int(fc[0, 3])
                if m != cur:
                    f.write("usemtl %s\n" % self.materials[m].name)
                    cur = m
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (
                    fc[0, 0] + 1, fc[0, 1] + 1, fc[0, 2] + 1,
                    fc[1, 0] + 1, fc[1, 1] + 1, fc[1, 2] + 1,
                    fc[2, 0] + 1, fc[2, 1] + 1, fc[2, 2] + 1))
        with open(os.path.join(directory, self.name + ".mtl"), "w") as f:
            for m in self.materials:
                f.write("newmtl %s\n" % m.name)
                f.write("Kd %.9g %.9g %.9g\n" % tuple(m.kd))
                f.write("Ks %.9g %.9g %.9g\n" % tuple(m.ks))
                f.write("Ns %.9g\n" % m.ns)
                if m.map_kd is not None:
                    f.write("map_Kd %s\n" % m.map_kd)
                    write_ppm(os.path.join(directory, m.map_kd), m.texture)
                f.write("\n")
        c = self.camera
        with open(os.path.join(directory, self.name + ".xml"), "w") as f:
            f.write('<?xml version="1.0" encoding="utf-8"?>\n')
            f.write('<camera type="perspective" width="%d" height="%d" fovy="%.9g">\n' % (c.width, c.height, c.fovy))
            f.write('  <eye x="%.9g" y="%.9g" z="%.9g"/>\n' % tuple(c.eye))
            f.write('  <lookat x="%.9g" y="%.9g" z="%.9g"/>\n' % tuple(c.lookat))
            f.write('  <up x="%.9g" y="%.9g" z="%.9g"/>\n' % tuple(c.up))
            f.write('</camera>\n')
            for m in self.materials:
                if any(r != 0 for r in m.radiance):
                    f.write('<light mtlname="%s" radiance="%.9g,%.9g,%.9g"/>\n' % ((m.name,) + tuple(m.radiance)))
        return obj


def write_ppm(path: str, img: np.ndarray) -> None:
    """Binary P6 -- a format stb_image (the reference's texture loader, model.cpp:8-23) reads."""
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


_P10 = np.array([float("1e%d" % e) for e in range(-40, 41)])          # exact decimal powers (correctly rounded by the parser)


def _qv(x) -> np.ndarray:
    """``_q`` for whole arrays: every element rounded to 9 significant decimal digits and read back, bit for bit what
    ``float("%.9g" % x)`` gives (checked against it on a sample every time a scene is finished).  x = m * 10^-j with a
    9-digit integer m: the candidates m0 - 1, m0, m0 + 1 are turned into doubles by ONE correctly rounded division or
    multiplication by an exact power of ten -- the parser's own fast path -- and the nearest wins (ties to the even m)."""
    x = np.asarray(x, np.float64)
    out = x.copy()
    fast = np.isfinite(x) & (np.abs(x) >= 1e-13) & (np.abs(x) < 1e30)                    # |j| <= 22 below: 10^|j| is an exact double
    slow = np.isfinite(x) & (x != 0) & ~fast                                           # (a handful of 1e-17s from sin(pi) and the like)
    if slow.any():
        out[slow] = [float("%.9g" % v) for v in x[slow].tolist()]
    a = np.abs(x[fast])
    if a.size == 0:
        return out
    k = np.floor(np.log10(a)).astype(np.int64)
    k = np.where(a < _P10[k + 40], k - 1, k); k = np.where(a >= _P10[k + 41], k + 1, k)   # guard log10 at powers of ten
    j = 8 - k
    pos = j >= 0
    sc = np.where(pos, _P10[np.where(pos, j, 0) + 40], 1.0); sc2 = np.where(pos, 1.0, _P10[np.where(pos, 0, -j) + 40])
    m0 = np.rint(np.where(pos, a * sc, a / sc2))
    best = bd = None
    for dm in (-1.0, 0.0, 1.0):
        m = m0 + dm
        c = np.where(pos, m / sc, m * sc2)
        d = np.abs(a - c)
        if best is None:
            best, bd = c, d
        else:
            take = (d < bd) | ((d == bd) & (np.fmod(m, 2.0) == 0.0))
            best = np.where(take, c, best); bd = np.where(take, d, bd)
    out[fast] = np.copysign(best, x[fast])
    return out


class _Mesh:
    """Accumulates unified-index geometry (one v / vn / vt triple per corner index).  Vertices and faces are kept as numpy
    blocks in insertion order; the builders below compute whole grids / spheres at once, in the element-wise operation order
    of the scalar formulas (so the numbers are the ones a per-vertex loop gives) -- a 4 M-triangle scene takes seconds."""

    def __init__(self):
        self.v: List[np.ndarray] = []      # (k,3) blocks, not yet rounded to the files' text form
        self.n: List[np.ndarray] = []
        self.t: List[np.ndarray] = []      # (k,2)
        self.f: List[np.ndarray] = []      # (k,4) int blocks: i0, i1, i2, material
        self.nv = 0

    def add_vertices(self, p, n, uv) -> int:
        p = np.asarray(p, np.float64).reshape(-1, 3)
        n = np.broadcast_to(np.asarray(n, np.float64), p.shape) if np.ndim(n) == 1 else np.asarray(n, np.float64).reshape(-1, 3)
        uv = np.asarray(uv, np.float64).reshape(-1, 2)
        self.v.append(p); self.n.append(np.array(n)); self.t.append(uv)
        base = self.nv
        self.nv += p.shape[0]
        return base

    def add_vertex(self, p, n, uv) -> int:
        return self.add_vertices([tuple(float(x) for x in p)], [tuple(float(x) for x in n)], [tuple(float(x) for x in uv)])

    def add_tri(self, a, b, c, mat):
        self.f.append(np.array([[a, b, c, mat]], np.int64))

    def add_quad(self, p0, p1, p2, p3, n, mat, uv=((0, 0), (1, 0), (1, 1), (0, 1))):
        i = [self.add_vertex(p, n, t) for p, t in zip((p0, p1, p2, p3), uv)]
        self.add_tri(i[0], i[1], i[2], mat)
        self.add_tri(i[0], i[2], i[3], mat)

    def _quads(self, idx, mat, first=None, second=None):
        """Two triangles (a, b, c), (a, c, d) per cell of the vertex-index lattice `idx`, cells in row-major order."""
        a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, 1:], idx[1:, :-1]
        m = np.full_like(a, mat)
        tri = np.stack([np.stack([a, b, c, m], -1), np.stack([a, c, d, m], -1)], 2)      # (rows, cols, 2, 4)
        keep = np.ones(tri.shape[:3], bool)
        if first is not None: keep[:, :, 0] = first[:, None]
        if second is not None: keep[:, :, 1] = second[:, None]
        self.f.append(tri[keep].reshape(-1, 4).astype(np.int64))

    def add_lattice(self, p, n, uv, mat, first=None, second=None):
        """(rows, cols, 3) positions / normals and (rows, cols, 2) texture coordinates -> vertices + the cells' triangles."""
        rows, cols = p.shape[:2]
        base = self.add_vertices(p.reshape(-1, 3), np.asarray(n, np.float64).reshape(-1, 3) if np.ndim(n) == 3 else n, uv.reshape(-1, 2))
        self._quads(base + np.arange(rows * cols, dtype=np.int64).reshape(rows, cols), mat, first, second)

    def add_grid(self, origin, du, dv, nu, nv, n, mat, uv_scale=1.0):
        """A planar rectangle origin + s*du + t*dv tessellated into nu x nv cells."""
        o = np.asarray(origin, float); du = np.asarray(du, float); dv = np.asarray(dv, float)
        ii = np.arange(nu + 1, dtype=np.float64); jj = np.arange(nv + 1, dtype=np.float64)
        p = (o[None, None, :] + du[None, None, :] * (ii / nu)[None, :, None]) + dv[None, None, :] * (jj / nv)[:, None, None]
        uv = np.stack(np.broadcast_arrays((uv_scale * ii / nu)[None, :], (uv_scale * jj / nv)[:, None]), -1)
        self.add_lattice(p, np.asarray(n, np.float64), uv, mat)

    def add_uv_sphere(self, center, radius, n_lon, n_lat, mat, flip=False):
        """UV sphere, smooth normals; n_lon*(n_lat-1)*2 triangles."""
        cx, cy, cz = center
        th = [math.pi * j / n_lat for j in range(n_lat + 1)]; ph = [2.0 * math.pi * i / n_lon for i in range(n_lon + 1)]
        st = np.array([math.sin(x) for x in th])[:, None]; ct = np.array([math.cos(x) for x in th])[:, None]
        cp = np.array([math.cos(x) for x in ph])[None, :]; sp = np.array([math.sin(x) for x in ph])[None, :]
        d = np.stack(np.broadcast_arrays(st * cp, ct, st * sp), -1)                   # (n_lat+1, n_lon+1, 3)
        p = np.stack([cx + radius * d[..., 0], cy + radius * d[..., 1], cz + radius * d[..., 2]], -1)
        uv = np.stack(np.broadcast_arrays((np.arange(n_lon + 1, dtype=np.float64) / n_lon)[None, :],
                                          (np.arange(n_lat + 1, dtype=np.float64) / n_lat)[:, None]), -1)
        jrow = np.arange(n_lat)
        self.add_lattice(p, -d if flip else d, uv, mat, first=jrow != 0, second=jrow != n_lat - 1)

    def finish(self, name, materials, camera, meta=None) -> SceneData:
        v = np.concatenate(self.v).reshape(-1, 3); n = np.concatenate(self.n).reshape(-1, 3); t = np.concatenate(self.t).reshape(-1, 2)
        raw = (v, n, t)
        v, n, t = _qv(v), _qv(n), _qv(t)
        rng = np.random.RandomState(12345)                                            # the text round trip itself, on a sample
        for r_, q_ in zip(raw, (v, n, t)):
            pick = rng.randint(0, r_.size, size=min(r_.size, 4096))
            assert all(float("%.9g" % a) == b for a, b in zip(r_.ravel()[pick].tolist(), q_.ravel()[pick].tolist())), "_qv departs from the text round trip"
        # the reference keeps the file's face order but needs `usemtl` runs; keep insertion order
        f = np.concatenate(self.f).reshape(-1, 4)
        face = np.zeros((f.shape[0], 3, 4), np.int32)
        for j in range(3):
            face[:, j, 0] = f[:, j]; face[:, j, 1] = f[:, j]; face[:, j, 2] = f[:, j]; face[:, j, 3] = f[:, 3]
        return SceneData(name, v, n, t, face, materials, camera, meta or {})


def _qcam(eye, lookat, up, fovy, w, h) -> Camera:
    return Camera(tuple(_q(x) for x in eye), tuple(_q(x) for x in lookat), tuple(_q(x) for x in up), _q(fovy), int(w), int(h))


# ---------------------------------------------------------------------------------------------- S-cornell
def cornell_box(width=800, height=800, sphere_lon=200, sphere_lat=100, wall_cells=1) -> SceneData:
    """S-cornell: unit box, five diffuse walls, ceiling quad light (17,12,4), glossy UV sphere.

    Default tessellation gives 200*(100-1)*2 = 39 600 sphere triangles + 10 wall + 2 light triangles.
    """
    mats = [
        Material("white", kd=(0.725, 0.71, 0.68)),
        Material("red", kd=(0.63, 0.065, 0.05)),
        Material("green", kd=(0.14, 0.45, 0.091)),
        Material("light", kd=(0.65, 0.65, 0.65), radiance=(17.0, 12.0, 4.0)),
        Material("glossy", kd=(0.3, 0.3, 0.3), ks=(0.5, 0.5, 0.5), ns=50.0),
    ]
    WHITE, RED, GREEN, LIGHT, GLOSSY = range(5)
    m = _Mesh()
    c = wall_cells
    m.add_grid((0, 0, 0), (1, 0, 0), (0, 0, 1), c, c, (0, 1, 0), WHITE)      # floor
    m.add_grid((0, 1, 0), (1, 0, 0), (0, 0, 1), c, c, (0, -1, 0), WHITE)     # ceiling
    m.add_grid((0, 0, 0), (1, 0, 0), (0, 1, 0), c, c, (0, 0, 1), WHITE)      # back wall
    m.add_grid((0, 0, 0), (0, 0, 1), (0, 1, 0), c, c, (1, 0, 0), RED)        # left
    m.add_grid((1, 0, 0), (0, 0, 1), (0, 1, 0), c, c, (-1, 0, 0), GREEN)     # right
    m.add_quad((0.35, 0.999, 0.35), (0.65, 0.999, 0.35), (0.65, 0.999, 0.65), (0.35, 0.999, 0.65), (0, -1, 0), LIGHT)
    m.add_uv_sphere((0.5, 0.3, 0.5), 0.3, sphere_lon, sphere_lat, GLOSSY)
    cam = _qcam((0.5, 0.5, 2.3), (0.5, 0.5, 0.0), (0, 1, 0), 40.0, width, height)
    return m.finish("cornell-box", mats, cam, {"kind": "S-cornell"})


def cornell_box_small(width=64, height=64) -> SceneData:
    """Cheap variant for CPU tests: same box, 24x12 sphere (528 triangles)."""
    return cornell_box(width, height, sphere_lon=24, sphere_lat=12)


# ---------------------------------------------------------------------------------------------- S-veach
def veach_mis(width=1280, height=720, light_lon=32, light_lat=16, plate_cells=8) -> SceneData:
    """S-veach: four Blinn-Phong plates Ns in {10,100,1000,5000}, four sphere lights of radii
    {.03,.1,.3,.9} with equal power, a diffuse floor and back wall."""
    radii = (0.03, 0.1, 0.3, 0.9)
    base = 800.0 * radii[0] ** 2       # radiance * r^2 constant -> equal power
    mats = [Material("backdrop", kd=(0.4, 0.4, 0.4))]
    for ns in (10.0, 100.0, 1000.0, 5000.0):
        mats.append(Material("plate%d" % int(ns), kd=(0.07, 0.09, 0.13), ks=(0.5, 0.5, 0.5), ns=ns))
    cols = ((1.0, 0.9, 0.8), (0.9, 1.0, 0.85), (0.8, 0.9, 1.0), (1.0, 0.85, 0.95))
    for r, col in zip(radii, cols):
        e = base / (r * r)
        mats.append(Material("light%d" % int(r * 100), kd=(0.0, 0.0, 0.0), radiance=tuple(_q(e * c) for c in col)))
    m = _Mesh()
    m.add_grid((-8, -0.2, -6), (16, 0, 0), (0, 0, 14), 4, 4, (0, 1, 0), 0)          # floor
    m.add_grid((-8, -0.2, -6), (16, 0, 0), (0, 10, 0), 4, 4, (0, 0, 1), 0)          # back wall
    # plates: tilted towards the camera so each reflects the row of lights
    for k, tilt in enumerate((28.0, 22.0, 16.0, 10.0)):
        z0 = 1.8 - 1.25 * k
        y0 = 0.0 + 0.28 * k
        a = math.radians(tilt)
        dv = (0.0, math.sin(a) * 1.0, -math.cos(a) * 1.0)
        nrm = (0.0, math.cos(a), math.sin(a))
        m.add_grid((-3.2, y0, z0), (6.4, 0, 0), dv, plate_cells, max(1, plate_cells // 4), nrm, 1 + k)
    for k, r in enumerate(radii):
        m.add_uv_sphere((-2.7 + 1.8 * k, 2.6, -1.9), r, light_lon, light_lat, 5 + k)
    cam = _qcam((0.0, 2.2, 9.5), (0.0, 0.9, 0.0), (0, 1, 0), 30.0, width, height)
    return m.finish("veach-mis", mats, cam, {"kind": "S-veach"})


# ---------------------------------------------------------------------------------------------- S-bath
def value_noise_texture(size: int, seed: int, base=(0.6, 0.6, 0.6), amp=0.35) -> np.ndarray:
    """Seeded multi-octave value noise, 3 channels, uint8 (stand-in for the bathroom's tile/wood images)."""
    rng = np.random.RandomState(seed)
    img = np.zeros((size, size, 3), np.float64)
    for octave in range(5):
        cells = 4 << octave
        g = rng.rand(cells + 1, cells + 1, 3)
        xs = np.linspace(0, cells, size, endpoint=False)
        i = xs.astype(int); f = xs - i
        f = f * f * (3 - 2 * f)
        a = g[i][:, i]; b = g[i][:, i + 1]; c = g[i + 1][:, i]; d = g[i + 1][:, i + 1]
        fx = f[None, :, None]; fy = f[:, None, None]
        img += ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy) / (2 ** octave)
    img = img / img.max()
    out = np.clip(np.asarray(base)[None, None, :] + amp * (img - 0.5) * 2.0, 0.0, 1.0)
    return (out * 255.0 + 0.5).astype(np.uint8)


def bathroom_stress(width=1920, height=1080, detail=64, tex_size=256) -> SceneData:
    """S-bath: a procedural room -- textured floor / walls, a mirror (Ns=10000), glossy
    fixtures built from tessellated spheres and a bumpy displaced "towel" grid, lit by a window quad.
    Triangle count grows ~ detail^2: detail=64 -> ~0.12 M, detail=160 -> ~0.6 M, detail=420 -> ~4.1 M."""
    mats = [
        Material("floor", kd=(0.6, 0.6, 0.6), map_kd="floor.ppm", texture=value_noise_texture(tex_size, 1, (0.55, 0.5, 0.45))),
        Material("wall", kd=(0.7, 0.7, 0.7), map_kd="wall.ppm", texture=value_noise_texture(tex_size, 2, (0.7, 0.72, 0.75), 0.2)),
        Material("wood", kd=(0.4, 0.3, 0.2), ks=(0.1, 0.1, 0.1), ns=30.0, map_kd="wood.ppm",
                 texture=value_noise_texture(tex_size, 3, (0.45, 0.3, 0.18), 0.3)),
        Material("towel", kd=(0.7, 0.2, 0.2), map_kd="towel.ppm", texture=value_noise_texture(tex_size, 4, (0.7, 0.25, 0.25), 0.25)),
        Material("mirror", kd=(0.0, 0.0, 0.0), ks=(1.0, 1.0, 1.0), ns=10000.0),
        Material("ceramic", kd=(0.8, 0.8, 0.8), ks=(0.4, 0.4, 0.4), ns=400.0),
        Material("chrome", kd=(0.05, 0.05, 0.05), ks=(0.8, 0.8, 0.8), ns=2000.0),
        Material("window", kd=(0.0, 0.0, 0.0), radiance=(25.0, 23.0, 20.0)),
    ]
    FLOOR, WALL, WOOD, TOWEL, MIRROR, CERAMIC, CHROME, WINDOW = range(8)
    m = _Mesh()
    W, H, D = 4.0, 2.6, 5.0
    g = max(2, detail // 4)
    m.add_grid((0, 0, 0), (W, 0, 0), (0, 0, D), g, g, (0, 1, 0), FLOOR, uv_scale=4.0)
    m.add_grid((0, H, 0), (W, 0, 0), (0, 0, D), g, g, (0, -1, 0), WALL, uv_scale=2.0)
    m.add_grid((0, 0, 0), (W, 0, 0), (0, H, 0), g, g, (0, 0, 1), WALL, uv_scale=3.0)
    m.add_grid((0, 0, 0), (0, 0, D), (0, H, 0), g, g, (1, 0, 0), WALL, uv_scale=3.0)
    m.add_grid((W, 0, 0), (0, 0, D), (0, H, 0), g, g, (-1, 0, 0), WALL, uv_scale=3.0)
    m.add_grid((0, 0, D), (W, 0, 0), (0, H, 0), g, g, (0, 0, -1), WALL, uv_scale=3.0)
    m.add_quad((0.8, 1.0, 0.01), (2.6, 1.0, 0.01), (2.6, 2.2, 0.01), (0.8, 2.2, 0.01), (0, 0, 1), MIRROR)
    m.add_quad((W - 0.01, 1.2, 1.5), (W - 0.01, 1.2, 3.2), (W - 0.01, 2.3, 3.2), (W - 0.01, 2.3, 1.5), (-1, 0, 0), WINDOW)
    # vanity: a box of wood grids
    m.add_grid((0.6, 0.85, 0.05), (2.2, 0, 0), (0, 0, 0.7), g, g, (0, 1, 0), WOOD, uv_scale=2.0)
    m.add_grid((0.6, 0.0, 0.75), (2.2, 0, 0), (0, 0.85, 0), g, g, (0, 0, 1), WOOD, uv_scale=2.0)
    # fixtures: spheres with ~detail^2 triangles each
    lon, lat = 2 * detail, detail
    m.add_uv_sphere((1.7, 0.95, 0.4), 0.22, lon, lat, CERAMIC)          # basin
    m.add_uv_sphere((1.7, 1.25, 0.12), 0.05, lon // 2, lat // 2, CHROME)   # tap
    m.add_uv_sphere((3.2, 0.45, 4.0), 0.45, lon, lat, CERAMIC)          # tub end
    m.add_uv_sphere((0.6, 0.35, 3.6), 0.35, lon, lat, CERAMIC)          # stool
    m.add_uv_sphere((2.2, 0.2, 2.4), 0.2, lon // 2, lat // 2, CHROME)
    # towel: displaced grid hanging on the left wall (bumpy -> deep, irregular BVH)
    nu = nv = 2 * detail
    o = np.array((0.05, 0.9, 2.0)); du = np.array((0.0, 0.0, 1.2)); dv = np.array((0.0, 1.0, 0.0))
    sv = [i / nu for i in range(nu + 1)]; tv = [j / nv for j in range(nv + 1)]
    s40 = np.array([math.sin(40 * x) for x in sv])[None, :]; c40 = np.array([math.cos(40 * x) for x in sv])[None, :]
    s34 = np.array([math.sin(34 * x) for x in tv])[:, None]; c34 = np.array([math.cos(34 * x) for x in tv])[:, None]
    S, T = np.meshgrid(np.array(sv), np.array(tv))                                     # [j, i]
    arg = 91 * S + 57 * T
    s91 = np.array([math.sin(x) for x in arg.ravel().tolist()]).reshape(arg.shape)     # (libm's sin, like the per-vertex loop this replaces)
    bump = 0.03 * s40 * c34 + 0.02 * s91
    p = ((o[None, None, :] + du[None, None, :] * S[..., None]) + dv[None, None, :] * T[..., None]) + np.stack([0.06 + bump, np.zeros_like(bump), np.zeros_like(bump)], -1)
    nx = np.stack(np.broadcast_arrays(np.ones_like(bump), (-0.03 * 34 * -s40) * s34, (-0.03 * 40 * c40) * c34), -1)
    nx = nx / np.sqrt((nx[..., 0] * nx[..., 0] + nx[..., 1] * nx[..., 1]) + nx[..., 2] * nx[..., 2])[..., None]
    m.add_lattice(p, nx, np.stack([S * 2, T * 2], -1), TOWEL)
    cam = _qcam((2.0, 1.5, 4.7), (1.8, 1.1, 0.0), (0, 1, 0), 55.0, width, height)
    return m.finish("bathroom2", mats, cam, {"kind": "S-bath", "detail": detail})


# tiny scenes for KATs ------------------------------------------------------------------------------
def open_box(width=32, height=32) -> SceneData:
    """14-triangle Cornell box without the sphere (the survey's smallest probe scene)."""
    s = cornell_box(width, height, sphere_lon=3, sphere_lat=2)
    keep = [k for k in range(s.n_faces) if s.face[k, 0, 3] != 4]
    s2 = SceneData("open-box", s.vertex, s.normal, s.texcoord, s.face[keep].copy(), s.materials, s.camera, {"kind": "open-box"})
    return s2


def tie_floor(width=96, height=96) -> SceneData:
    """S-cornell-small with EIGHT coincident copies of its floor, each in its own colour: every floor hit is an eight-way exact tie -- the
    known-answer scene for the tie-breaking rule (the reference: the copy that comes first in its BVH::triangles order, BVH.cpp:95-113)."""
    base = cornell_box_small(width, height)
    floor = base.face[:2].copy()                                           # add_grid(floor) comes first: two triangles
    cols = [(0.9, 0.1, 0.1), (0.1, 0.9, 0.1), (0.1, 0.1, 0.9), (0.9, 0.9, 0.1), (0.9, 0.1, 0.9), (0.1, 0.9, 0.9), (0.5, 0.5, 0.5), (0.2, 0.2, 0.2)]
    mats = list(base.materials); faces = [base.face]
    for k, c in enumerate(cols[1:]):
        mats.append(Material("floor%d" % k, kd=c)); f = floor.copy(); f[:, :, 3] = len(mats) - 1; faces.append(f)
    return SceneData("ties", base.vertex, base.normal, base.texcoord, np.concatenate(faces), mats, base.camera, {})


SCENES = {
    "tie-floor": tie_floor,
    "cornell-box": cornell_box,
    "cornell-box-small": cornell_box_small,
    "veach-mis": veach_mis,
    "bathroom2": bathroom_stress,
    "open-box": open_box,
}
