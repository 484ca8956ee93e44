"""ctypes plumbing over the C ABI of ``csrc/libmcpt_hip.so`` (``include/mcpt.h``).

This is NOT the product's host layer -- that is C++ (``host/``: ``Model`` / ``Scene`` / ``Render`` with the
reference's method names, and the ``mcpt_cli`` driver).  Python exists here only so that ``tests/``,
``bench.py`` and ``__graft_entry__.py`` can call the same C entry points, and so that ``torch`` can lend
device memory, streams and ``torch.distributed`` (RCCL) to the multi-GPU bench.

There is no CPU fallback: if the HIP library is missing this module raises at load time, and
``Renderer`` raises when ``mcpt_create`` finds no device.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import scenes  # noqa: F401  (re-export)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmcpt_hip.so")

MCPT_OK = 0
INTEGRATOR_MIS = 0
INTEGRATOR_RECURSIVE_NEE = 1
FLAG_CORRECT_SHADOW_T2 = 0x1
FLAG_DETERMINISTIC = 0x2
FLAG_COUNT_TRAVERSAL = 0x4
FLAG_GPU_BVH_BUILD = 0x8
FLAG_REFERENCE_TIE_ORDER = 0x10


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_float))]


class MaterialC(C.Structure):
    _fields_ = [("ks", C.c_double * 3), ("ns", C.c_double), ("radiance", C.c_double * 3),
                ("map_kd", C.c_int32), ("reserved", C.c_int32)]


class CameraC(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("lookat", C.c_double * 3), ("up", C.c_double * 3),
                ("fovy", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("vertex", C.POINTER(C.c_double)), ("n_vertex", C.c_uint32),
                ("normal", C.POINTER(C.c_double)), ("n_normal", C.c_uint32),
                ("texcoord", C.POINTER(C.c_double)), ("n_texcoord", C.c_uint32),
                ("face", C.POINTER(C.c_int32)), ("n_face", C.c_uint32),
                ("materials", C.POINTER(MaterialC)), ("n_materials", C.c_uint32),
                ("textures", C.POINTER(Texture)), ("n_textures", C.c_uint32),
                ("camera", CameraC)]


class Opts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("max_depth", C.c_uint32),
                ("integrator", C.c_uint32), ("flags", C.c_uint32), ("samples_per_item", C.c_uint32),
                ("reserved", C.c_uint32 * 4)]


class Counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays_primary", C.c_uint64), ("rays_continuation", C.c_uint64),
                ("rays_shadow", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("shaded_hits", C.c_uint64), ("texel_fetches", C.c_uint64),
                ("self_shadow_tests", C.c_uint64), ("self_shadow_hits", C.c_uint64),
                ("kernel_ms", C.c_double), ("kernel_ms_total", C.c_double), ("launches", C.c_uint64),
                ("trace_ms_total", C.c_double), ("shade_ms_total", C.c_double), ("iterations", C.c_uint64),
                ("stack_spills", C.c_uint64), ("debug", C.c_uint64 * 4)]

    @property
    def rays(self) -> int:
        return int(self.rays_primary + self.rays_continuation + self.rays_shadow)

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class SceneInfo(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("n_lights", C.c_uint32), ("n_nodes", C.c_uint32),
                ("bvh_depth", C.c_uint32), ("max_leaf", C.c_uint32), ("width", C.c_uint32),
                ("height", C.c_uint32), ("device_bytes", C.c_uint64), ("bvh_build_ms", C.c_double),
                ("upload_ms", C.c_double), ("wide_width", C.c_uint32), ("wide_nodes", C.c_uint32), ("wide_depth", C.c_uint32),
                ("reserved0", C.c_uint32), ("traversal_bytes", C.c_uint64), ("centre", C.c_double * 3), ("wide_tree_hash", C.c_uint64)]


def texture_to_float(img_u8: np.ndarray) -> np.ndarray:
    """What stbi_loadf gives the reference for an 8-bit image (model.cpp:8-23; stb_image.h:1553,1849):
    (c/255)^2.2 per channel, row 0 = first row of the file."""
    x = img_u8.astype(np.float32) / np.float32(255.0)
    return np.power(x, np.float32(2.2)).astype(np.float32)


class DescHolder:
    """Builds an mcpt_scene_desc from a scenes.SceneData and keeps the backing arrays alive."""

    def __init__(self, scene: "scenes.SceneData"):
        self.vertex = np.ascontiguousarray(scene.vertex, np.float64)
        self.normal = np.ascontiguousarray(scene.normal, np.float64)
        self.texcoord = np.ascontiguousarray(scene.texcoord, np.float64)
        self.face = np.ascontiguousarray(scene.face, np.int32)
        self.tex_arrays = []
        n = len(scene.materials)
        self.materials = (MaterialC * n)()
        self.textures = (Texture * n)()
        for i, m in enumerate(scene.materials):
            if m.texture is not None and m.texture.dtype == np.float32:
                t = np.ascontiguousarray(m.texture)               # already what stbi_loadf would return (tests feed reference texels)
            elif m.texture is not None:
                t = np.ascontiguousarray(texture_to_float(m.texture))
            else:  # Texture(Color3f kd): kd parsed with stof (model.cpp:189-193)
                t = np.asarray(m.kd, np.float32).reshape(1, 1, 3).copy()
            self.tex_arrays.append(t)
            self.textures[i].width = t.shape[1]
            self.textures[i].height = t.shape[0]
            self.textures[i].rgb = t.ctypes.data_as(C.POINTER(C.c_float))
            mc = self.materials[i]
            for k in range(3):
                mc.ks[k] = float(m.ks[k]); mc.radiance[k] = float(m.radiance[k])
            mc.ns = float(m.ns)
            mc.map_kd = i
        d = SceneDesc()
        d.vertex = self.vertex.ctypes.data_as(C.POINTER(C.c_double)); d.n_vertex = self.vertex.shape[0]
        d.normal = self.normal.ctypes.data_as(C.POINTER(C.c_double)); d.n_normal = self.normal.shape[0]
        d.texcoord = self.texcoord.ctypes.data_as(C.POINTER(C.c_double)); d.n_texcoord = self.texcoord.shape[0]
        d.face = self.face.ctypes.data_as(C.POINTER(C.c_int32)); d.n_face = self.face.shape[0]
        d.materials = C.cast(self.materials, C.POINTER(MaterialC)); d.n_materials = n
        d.textures = C.cast(self.textures, C.POINTER(Texture)); d.n_textures = n
        cam = scene.camera
        for k in range(3):
            d.camera.eye[k] = cam.eye[k]; d.camera.lookat[k] = cam.lookat[k]; d.camera.up[k] = cam.up[k]
        d.camera.fovy = cam.fovy; d.camera.width = cam.width; d.camera.height = cam.height
        self.desc = d


_lib = None


def load_library() -> C.CDLL:
    """Load csrc/libmcpt_hip.so.  Raises if it has not been built -- there is nothing to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    try:          # torch bundles its own libamdhip64: let it load first so both sides share one HIP runtime in this process
        import torch  # noqa: F401
    except Exception:
        pass
    path = os.environ.get("MCPT_LIB_PATH", LIB_PATH)     # developer override: A/B another build of the same ABI
    if not os.path.exists(path):
        raise RuntimeError("libmcpt_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C monte-carlo-path-tracer_amd/csrc`.  No CPU fallback exists." % path)
    lib = C.CDLL(path)
    P = C.POINTER
    vp = C.c_void_p
    sigs = {
        "mcpt_create": [P(SceneDesc), P(Opts), P(vp)],
        "mcpt_destroy": [vp],
        "mcpt_check_scene": [P(SceneDesc), P(SceneInfo)],
        "mcpt_get_scene_info": [vp, P(SceneInfo)],
        "mcpt_render": [vp, C.c_uint32, C.c_uint64, C.c_uint32],
        "mcpt_render_tiles": [vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32],
        "mcpt_sync": [vp],
        "mcpt_read_accum": [vp, vp],
        "mcpt_write_accum": [vp, vp],
        "mcpt_clear_accum": [vp],
        "mcpt_tonemap": [vp, vp, C.c_int],
        "mcpt_get_counters": [vp, P(Counters)],
        "mcpt_reset_counters": [vp],
        "mcpt_bind_accum": [vp, vp],
        "mcpt_clone_to_device": [vp, C.c_int32, P(vp)],
        "mcpt_tonemap_buffer": [vp, vp, vp, C.c_int],
        "mcpt_tonemap_map": [vp, C.c_int, C.POINTER(C.c_void_p)],
        "mcpt_accum_device_ptr": [vp, P(vp)],
        "mcpt_set_stream": [vp, vp],
        "mcpt_set_null_stream": [vp],
        "mcpt_probe_trace4": [vp, C.c_uint32, vp, vp, vp, C.c_int, vp, vp, vp, vp],
        "mcpt_probe_trace": [vp, C.c_uint32, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp],
        "mcpt_probe_cast_ray": [vp, C.c_uint32, vp, vp, vp],
        "mcpt_probe_hit_shade": [vp, C.c_uint32, vp, vp, vp, vp, vp],
        "mcpt_probe_bsdf": [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp],
        "mcpt_probe_sample_light": [vp, C.c_uint32, vp, vp, vp],
        "mcpt_probe_paths": [vp, C.c_uint32, vp, vp, C.c_uint64, vp],
        "mcpt_probe_rng": [vp, C.c_uint32, vp, C.c_uint64, vp],
        "mcpt_probe_texture": [vp, C.c_uint32, C.c_uint32, vp, vp],
    }
    for name, args in sigs.items():
        if not hasattr(lib, name) and "MCPT_LIB_PATH" in os.environ:
            continue                                     # developer A/B against an older build of the library
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.mcpt_last_error.restype = C.c_char_p
    lib.mcpt_abi_version.restype = C.c_uint32
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "mcpt_create", "mcpt_destroy", "mcpt_clone_to_device", "mcpt_tonemap_buffer", "mcpt_check_scene", "mcpt_get_scene_info", "mcpt_last_error", "mcpt_abi_version",
    "mcpt_render", "mcpt_render_tiles", "mcpt_sync", "mcpt_read_accum", "mcpt_write_accum", "mcpt_clear_accum", "mcpt_tonemap", "mcpt_tonemap_map",
    "mcpt_get_counters", "mcpt_reset_counters", "mcpt_bind_accum", "mcpt_accum_device_ptr", "mcpt_set_stream",
    "mcpt_set_null_stream", "mcpt_probe_trace", "mcpt_probe_trace4", "mcpt_probe_cast_ray", "mcpt_probe_hit_shade", "mcpt_probe_bsdf", "mcpt_probe_sample_light",
    "mcpt_probe_paths", "mcpt_probe_rng", "mcpt_probe_texture",
]


class McptError(RuntimeError):
    pass


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def check_scene(scene: "scenes.SceneData"):
    """Host-only validation + BVH build (no device needed).  Returns (status, SceneInfo, message)."""
    lib = load_library()
    holder = DescHolder(scene)
    info = SceneInfo()
    st = lib.mcpt_check_scene(C.byref(holder.desc), C.byref(info))
    return st, info, (lib.mcpt_last_error() or b"").decode() if st != MCPT_OK else ""


class Renderer:
    """One mcpt_ctx = one scene on one GPU (the reference's ``Render`` object)."""

    def __init__(self, scene: "scenes.SceneData", max_depth=0, integrator=INTEGRATOR_MIS, flags=0, device=0,
                 samples_per_item=0):
        self.lib = load_library()
        self.holder = DescHolder(scene)
        self.width, self.height = scene.camera.width, scene.camera.height
        o = Opts()
        o.struct_size = C.sizeof(Opts); o.device = device; o.max_depth = max_depth
        o.integrator = integrator; o.flags = flags; o.samples_per_item = samples_per_item
        self.ctx = C.c_void_p()
        self._check(self.lib.mcpt_create(C.byref(self.holder.desc), C.byref(o), C.byref(self.ctx)))

    def clone(self, device=0) -> "Renderer":
        """A second Renderer for the same scene (mcpt_clone_to_device): no flatten, no BVH build."""
        other = Renderer.__new__(Renderer)
        other.lib = self.lib; other.holder = self.holder; other.width, other.height = self.width, self.height
        other.ctx = C.c_void_p()
        self._check(self.lib.mcpt_clone_to_device(self.ctx, int(device), C.byref(other.ctx)))
        return other

    def _check(self, status):
        if status != MCPT_OK:
            raise McptError("mcpt status %d: %s" % (status, (self.lib.mcpt_last_error() or b"").decode()))

    def close(self):
        if getattr(self, "ctx", None) is not None and self.ctx.value:
            self.lib.mcpt_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- hot path
    def render(self, spp: int, seed: int = 0, first_sample: int = 0):
        self._check(self.lib.mcpt_render(self.ctx, spp, seed, first_sample))

    def render_tiles(self, spp: int, seed: int, first_sample: int, tile_mod: int, tile_rem: int):
        """One interleaved share of the image: the 8x8 tiles t with t % tile_mod == tile_rem."""
        self._check(self.lib.mcpt_render_tiles(self.ctx, spp, seed, first_sample, tile_mod, tile_rem))

    def sync(self):
        self._check(self.lib.mcpt_sync(self.ctx))

    def read_accum(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._check(self.lib.mcpt_read_accum(self.ctx, _ptr(out)))
        return out

    def write_accum(self, rgba: np.ndarray):
        a = np.ascontiguousarray(rgba, np.float32)
        assert a.size == self.width * self.height * 4
        self._check(self.lib.mcpt_write_accum(self.ctx, _ptr(a)))

    def clear(self):
        self._check(self.lib.mcpt_clear_accum(self.ctx))

    def tonemap(self, flip_y=False) -> np.ndarray:
        out = np.zeros((self.height, self.width, 3), np.uint8)
        self._check(self.lib.mcpt_tonemap(self.ctx, _ptr(out), 1 if flip_y else 0))
        return out

    def tonemap_map(self, flip_y=False) -> np.ndarray:
        """mcpt_tonemap_map: a copy of the context's own pinned image (the C pointer stays valid until the next tonemap call)."""
        p = C.c_void_p()
        self._check(self.lib.mcpt_tonemap_map(self.ctx, 1 if flip_y else 0, C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.height, self.width, 3)).copy()

    def accum_device_ptr(self) -> int:
        p = C.c_void_p()
        self._check(self.lib.mcpt_accum_device_ptr(self.ctx, C.byref(p)))
        return p.value

    def tonemap_buffer(self, device_ptr: int, flip_y=False) -> np.ndarray:
        out = np.zeros((self.height, self.width, 3), np.uint8)
        self._check(self.lib.mcpt_tonemap_buffer(self.ctx, C.c_void_p(device_ptr), _ptr(out), 1 if flip_y else 0))
        return out

    def counters(self) -> Counters:
        c = Counters()
        self._check(self.lib.mcpt_get_counters(self.ctx, C.byref(c)))
        return c

    def reset_counters(self):
        self._check(self.lib.mcpt_reset_counters(self.ctx))

    def info(self) -> SceneInfo:
        i = SceneInfo()
        self._check(self.lib.mcpt_get_scene_info(self.ctx, C.byref(i)))
        return i

    def bind_accum(self, device_ptr: int):
        self._check(self.lib.mcpt_bind_accum(self.ctx, C.c_void_p(device_ptr)))

    def set_stream(self, hip_stream: int):
        """A caller-owned stream handle; 0 = back to the context's own stream (see set_null_stream for the default stream)."""
        self._check(self.lib.mcpt_set_stream(self.ctx, C.c_void_p(hip_stream)))

    def set_null_stream(self):
        """Order the context's work with the device's legacy default stream (torch's default stream has handle 0)."""
        self._check(self.lib.mcpt_set_null_stream(self.ctx))

    def set_torch_stream(self, stream):
        """Bind a torch.cuda.Stream: its handle, or the legacy default stream when the handle is 0."""
        h = int(stream.cuda_stream)
        if h:
            self.set_stream(h)
        else:
            self.set_null_stream()

    # ---- probes
    def probe_trace(self, origin, direction, t1=None, t2=None, any_hit=False):
        o = np.ascontiguousarray(origin, np.float64).reshape(-1, 3); d = np.ascontiguousarray(direction, np.float64).reshape(-1, 3)
        n = o.shape[0]
        t1 = np.full(n, 1e-4) if t1 is None else np.ascontiguousarray(t1, np.float64)
        t2 = np.full(n, np.finfo(np.float64).max) if t2 is None else np.ascontiguousarray(t2, np.float64)
        ot = np.zeros(n, np.float32); tri = np.zeros(n, np.int32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        self._check(self.lib.mcpt_probe_trace(self.ctx, n, _ptr(o), _ptr(d), _ptr(t1), _ptr(t2), 1 if any_hit else 0,
                                              _ptr(ot), _ptr(tri), _ptr(u), _ptr(v)))
        return ot, tri, u, v

    def probe_trace4(self, origin, direction, t2=None, any_hit=False):
        """BVH::hit / has_hit through the production wf_trace8_kernel (8-wide compressed tree)."""
        o = np.ascontiguousarray(origin, np.float64).reshape(-1, 3); d = np.ascontiguousarray(direction, np.float64).reshape(-1, 3)
        n = o.shape[0]
        t2 = np.full(n, np.finfo(np.float64).max) if t2 is None else np.ascontiguousarray(t2, np.float64)
        ot = np.zeros(n, np.float32); tri = np.zeros(n, np.int32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        self._check(self.lib.mcpt_probe_trace4(self.ctx, n, _ptr(o), _ptr(d), _ptr(t2), 1 if any_hit else 0,
                                               _ptr(ot), _ptr(tri), _ptr(u), _ptr(v)))
        return ot, tri, u, v

    def probe_hit_shade(self, face, u, v, direction):
        face = np.ascontiguousarray(face, np.int32); u = np.ascontiguousarray(u, np.float32); v = np.ascontiguousarray(v, np.float32)
        d = np.ascontiguousarray(direction, np.float64).reshape(-1, 3); n = face.shape[0]
        out = np.zeros((n, 6), np.float32)
        self._check(self.lib.mcpt_probe_hit_shade(self.ctx, n, _ptr(face), _ptr(u), _ptr(v), _ptr(d), _ptr(out)))
        return out

    def probe_cast_ray(self, xy, xi):
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2); xi = np.ascontiguousarray(xi, np.float32).reshape(-1, 2)
        out = np.zeros((xy.shape[0], 6), np.float32)
        self._check(self.lib.mcpt_probe_cast_ray(self.ctx, xy.shape[0], _ptr(xy), _ptr(xi), _ptr(out)))
        return out

    def probe_bsdf(self, normal, wi, kd, ks, ns, wo, xi):
        arrs = [np.ascontiguousarray(a, np.float32) for a in (normal, wi, kd, ks, ns, wo, xi)]
        n = arrs[4].shape[0]
        out = np.zeros((n, 12), np.float32)
        self._check(self.lib.mcpt_probe_bsdf(self.ctx, n, *[_ptr(a) for a in arrs], _ptr(out)))
        return out

    def probe_sample_light(self, point, xi):
        p = np.ascontiguousarray(point, np.float64).reshape(-1, 3); xi = np.ascontiguousarray(xi, np.float32).reshape(-1, 3)
        out = np.zeros((p.shape[0], 10), np.float32)
        self._check(self.lib.mcpt_probe_sample_light(self.ctx, p.shape[0], _ptr(p), _ptr(xi), _ptr(out)))
        return out

    def probe_paths(self, origin, direction, seed=0):
        o = np.ascontiguousarray(origin, np.float64).reshape(-1, 3); d = np.ascontiguousarray(direction, np.float64).reshape(-1, 3)
        out = np.zeros((o.shape[0], 3), np.float32)
        self._check(self.lib.mcpt_probe_paths(self.ctx, o.shape[0], _ptr(o), _ptr(d), seed, _ptr(out)))
        return out

    def probe_texture(self, material, uv):
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((uv.shape[0], 3), np.float32)
        self._check(self.lib.mcpt_probe_texture(self.ctx, material, uv.shape[0], _ptr(uv), _ptr(out)))
        return out

    def probe_rng(self, pixel_sample_block, seed=0):
        k = np.ascontiguousarray(pixel_sample_block, np.uint32).reshape(-1, 3)
        out = np.zeros((k.shape[0], 4), np.float32)
        self._check(self.lib.mcpt_probe_rng(self.ctx, k.shape[0], _ptr(k), seed, _ptr(out)))
        return out
