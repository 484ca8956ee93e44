"""TEST INFRASTRUCTURE ONLY -- ctypes wrappers for

* ``oracle/liboracle.so``          the CPU restatement (``mcpt_oracle.cpp``), and
* ``oracle/_ref/libmcpt_ref*.so``  the REAL reference compiled by ``build_ref.sh`` (present in the build
                                   container and shipped to the GPU box as a binary; absent => ``Reference``
                                   raises ``ReferenceUnavailable`` and callers fall back to tests/golden/).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import __graft_entry__ as _ge  # noqa: E402

pkg = _ge.load_package()

ORACLE_LIB = os.path.join(_HERE, "liboracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libmcpt_ref.so")
REF_DEPTH_LIB = os.path.join(_HERE, "_ref", "libmcpt_ref_depth.so")

vp = C.c_void_p


def _p(a):
    return None if a is None else a.ctypes.data_as(vp)


def _d3(x):
    return np.ascontiguousarray(x, np.float64).reshape(-1)


def _f(x):
    return np.ascontiguousarray(x, np.float32).reshape(-1)


class ReferenceUnavailable(RuntimeError):
    pass


def build_oracle():
    import subprocess
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


class Oracle:
    """CPU restatement.  COUNTER-mode rendering + SEQ-mode probes."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            if not os.path.exists(ORACLE_LIB):
                build_oracle()
            L = C.CDLL(ORACLE_LIB)
            L.oracle_create.restype = vp
            L.oracle_create.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
            L.oracle_destroy.argtypes = [vp]
            L.oracle_set_opts.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
            L.oracle_render.restype = C.c_double
            L.oracle_render.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint32, vp, vp, C.c_int]
            L.oracle_info.argtypes = [vp, vp]
            L.oracle_light_tri.argtypes = [vp, C.c_int]
            L.oracle_tonemap.argtypes = [vp, C.c_int, vp]
            L.oracle_rng_block.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp]
            L.oracle_cast_ray.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp]
            L.oracle_aabb_intersect.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double]
            L.oracle_tri_hit.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_double, C.c_double, vp]
            L.oracle_tri_any.argtypes = [vp, vp, vp, C.c_double, C.c_double]
            L.oracle_tri_area.argtypes = [vp]; L.oracle_tri_area.restype = C.c_float
            L.oracle_bvh_hit.argtypes = [vp, vp, vp, C.c_double, C.c_double, vp]
            L.oracle_bvh_has_hit.argtypes = [vp, vp, vp, C.c_double, C.c_double]
            L.oracle_bsdf_setup.argtypes = [vp, vp, vp, vp, C.c_double, vp]
            L.oracle_bsdf_eval.argtypes = [vp, vp, vp, vp, C.c_double, vp, vp]
            L.oracle_bsdf_sample.argtypes = [vp, vp, vp, vp, C.c_double, vp, C.c_int, vp]
            L.oracle_power_heuristic.argtypes = [C.c_float, C.c_float]; L.oracle_power_heuristic.restype = C.c_float
            L.oracle_clamp01.argtypes = [C.c_float]; L.oracle_clamp01.restype = C.c_double
            L.oracle_texture_get_color.argtypes = [C.c_int, C.c_int, vp, C.c_double, C.c_double, vp]
            L.oracle_sample_light.argtypes = [vp, vp, vp, C.c_int, vp]
            L.oracle_trace_path.argtypes = [vp, vp, vp, vp, C.c_int, vp]
            L.oracle_trace_path_recursive.argtypes = [vp, vp, vp, vp, C.c_int, vp]
            L.oracle_trace_pixel.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp]
            L.oracle_trace_path_counter.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint64, vp]
            L.oracle_shadow_probe.argtypes = [vp, vp, vp, vp, C.c_int, C.c_double, vp]
            cls._lib = L
        return cls._lib

    def __init__(self, scene, max_depth=0, integrator=0, flags=0):
        self.L = self.lib()
        self.holder = pkg.DescHolder(scene)
        self.width, self.height = scene.camera.width, scene.camera.height
        self.h = self.L.oracle_create(C.addressof(self.holder.desc), max_depth, integrator, flags)

    def close(self):
        if getattr(self, "h", None):
            self.L.oracle_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_opts(self, max_depth=0, integrator=0, flags=0):
        self.L.oracle_set_opts(self.h, max_depth, integrator, flags)

    def info(self):
        o = (C.c_long * 5)(); self.L.oracle_info(self.h, o)
        return dict(tris=o[0], lights=o[1], nodes=o[2], leaves=o[3], depth=o[4])

    def render(self, spp, seed=0, first_sample=0, accum=None, threads=0):
        """Adds spp samples/pixel to accum (h,w,4); returns (accum, counters dict, seconds)."""
        if accum is None:
            accum = np.zeros((self.height, self.width, 4), np.float32)
        cnt = np.zeros(8, np.uint64)
        secs = self.L.oracle_render(self.h, spp, seed, first_sample, _p(accum), _p(cnt), threads)
        names = ["paths", "rays_primary", "rays_continuation", "rays_shadow", "box_tests", "tri_tests",
                 "self_shadow_tests", "self_shadow_hits"]
        return accum, {k: int(v) for k, v in zip(names, cnt)}, secs

    @staticmethod
    def tonemap(accum):
        a = np.ascontiguousarray(accum, np.float32)
        out = np.zeros(a.shape[:-1] + (3,), np.uint8)
        Oracle.lib().oracle_tonemap(_p(a), a.size // 4, _p(out))
        return out

    @staticmethod
    def rng_block(pixel, sample, block, seed=0):
        o = np.zeros(4, np.float32); Oracle.lib().oracle_rng_block(pixel, sample, block, seed, _p(o)); return o

    # ---- SEQ probes (mirror Reference's)
    def cast_ray(self, x, y, xi):
        xi = _f(xi); o = np.zeros(6); n = self.L.oracle_cast_ray(self.h, x, y, _p(xi), xi.size, _p(o)); return o, n

    @staticmethod
    def aabb_intersect(A, B, o, d, t1, t2):
        return Oracle.lib().oracle_aabb_intersect(_p(_d3(A)), _p(_d3(B)), _p(_d3(o)), _p(_d3(d)), t1, t2)

    @staticmethod
    def tri_hit(v9, vn9, uv6, emissive, o, d, t1, t2):
        out = np.zeros(13)
        h = Oracle.lib().oracle_tri_hit(_p(_d3(v9)), _p(_d3(vn9)), _p(_d3(uv6)), int(emissive), _p(_d3(o)), _p(_d3(d)), t1, t2, _p(out))
        return h, out

    @staticmethod
    def tri_any(v9, o, d, t1, t2):
        return Oracle.lib().oracle_tri_any(_p(_d3(v9)), _p(_d3(o)), _p(_d3(d)), t1, t2)

    @staticmethod
    def tri_area(v9):
        return Oracle.lib().oracle_tri_area(_p(_d3(v9)))

    def bvh_hit(self, o, d, t1=1e-4, t2=np.finfo(np.float64).max):
        out = np.zeros(12); h = self.L.oracle_bvh_hit(self.h, _p(_d3(o)), _p(_d3(d)), t1, t2, _p(out)); return h, out

    def bvh_has_hit(self, o, d, t1=1e-4, t2=np.finfo(np.float64).max):
        return self.L.oracle_bvh_has_hit(self.h, _p(_d3(o)), _p(_d3(d)), t1, t2)

    @staticmethod
    def bsdf_setup(n, wi, kd, ks, ns):
        out = np.zeros(18, np.float32)
        Oracle.lib().oracle_bsdf_setup(_p(_d3(n)), _p(_d3(wi)), _p(_d3(kd)), _p(_d3(ks)), ns, _p(out)); return out

    @staticmethod
    def bsdf_eval(n, wi, kd, ks, ns, wo):
        out = np.zeros(4, np.float32); wo = _f(wo)
        Oracle.lib().oracle_bsdf_eval(_p(_d3(n)), _p(_d3(wi)), _p(_d3(kd)), _p(_d3(ks)), ns, _p(wo), _p(out)); return out

    @staticmethod
    def bsdf_sample(n, wi, kd, ks, ns, xi):
        out = np.zeros(8, np.float32); xi = _f(xi)
        c = Oracle.lib().oracle_bsdf_sample(_p(_d3(n)), _p(_d3(wi)), _p(_d3(kd)), _p(_d3(ks)), ns, _p(xi), xi.size, _p(out)); return out, c

    @staticmethod
    def power_heuristic(a, b):
        return Oracle.lib().oracle_power_heuristic(a, b)

    @staticmethod
    def texture_get_color(img_f32, u, v):
        img = np.ascontiguousarray(img_f32, np.float32); out = np.zeros(3, np.float32)
        Oracle.lib().oracle_texture_get_color(img.shape[1], img.shape[0], _p(img), u, v, _p(out)); return out

    def sample_light(self, p, xi):
        xi = _f(xi); out = np.zeros(14); c = self.L.oracle_sample_light(self.h, _p(_d3(p)), _p(xi), xi.size, _p(out)); return out, c

    def trace_path(self, o, d, xi, recursive=False):
        xi = _f(xi); L3 = np.zeros(3, np.float32)
        fn = self.L.oracle_trace_path_recursive if recursive else self.L.oracle_trace_path
        c = fn(self.h, _p(_d3(o)), _p(_d3(d)), _p(xi), xi.size, _p(L3)); return L3, c

    def trace_pixel(self, x, y, xi):
        xi = _f(xi); L3 = np.zeros(3, np.float32)
        c = self.L.oracle_trace_pixel(self.h, x, y, _p(xi), xi.size, _p(L3)); return L3, c

    def trace_path_counter(self, o, d, item, seed=0):
        L3 = np.zeros(3, np.float32)
        self.L.oracle_trace_path_counter(self.h, _p(_d3(o)), _p(_d3(d)), item, seed, _p(L3)); return L3

    def shadow_probe(self, o, d, xi, shrink=1e-4):
        xi = _f(xi); out = np.zeros(2, np.int32)
        ok = self.L.oracle_shadow_probe(self.h, _p(_d3(o)), _p(_d3(d)), _p(xi), xi.size, shrink, _p(out)); return ok, out


class Reference:
    """The real reference (oracle/_ref).  One scene per process (the driver holds globals)."""

    def __init__(self, depth_variant=False):
        path = REF_DEPTH_LIB if depth_variant else REF_LIB
        if not os.path.exists(path):
            raise ReferenceUnavailable(path + " missing: run oracle/build_ref.sh where /root/reference exists")
        L = C.CDLL(path)
        L.ref_render.restype = C.c_double
        L.ref_rand1f.restype = C.c_float
        L.ref_tri_area.restype = C.c_float
        L.ref_power_heuristic.restype = C.c_float; L.ref_power_heuristic.argtypes = [C.c_float, C.c_float]
        L.ref_clamp01.restype = C.c_double; L.ref_clamp01.argtypes = [C.c_float]
        L.ref_aabb_intersect.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double]
        L.ref_tri_hit.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_double, C.c_double, vp]
        L.ref_tri_any.argtypes = [vp, vp, vp, C.c_double, C.c_double]
        L.ref_tri_area.argtypes = [vp]
        L.ref_bvh_hit.argtypes = [vp, vp, C.c_double, C.c_double, vp]
        L.ref_bvh_has_hit.argtypes = [vp, vp, C.c_double, C.c_double]
        L.ref_bsdf_setup.argtypes = [vp, vp, vp, vp, C.c_double, vp]
        L.ref_bsdf_eval.argtypes = [vp, vp, vp, vp, C.c_double, vp, vp]
        L.ref_bsdf_sample.argtypes = [vp, vp, vp, vp, C.c_double, vp]
        L.ref_texture_get_color.argtypes = [C.c_int, C.c_int, vp, C.c_double, C.c_double, vp]
        L.ref_sample_light.argtypes = [vp, vp]
        L.ref_trace_path.argtypes = [vp, vp, vp]
        L.ref_trace_path_recursive.argtypes = [vp, vp, vp]
        L.ref_trace_pixel.argtypes = [C.c_int, C.c_int, vp]
        L.ref_shadow_probe.argtypes = [vp, vp, C.c_double, vp]
        L.ref_cast_ray.argtypes = [C.c_int, C.c_int, vp]
        L.ref_rng_inject.argtypes = [vp, C.c_int]
        L.ref_get_accum.argtypes = [vp]
        L.ref_get_pixels_u8.argtypes = [vp]
        L.ref_set_pixel.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
        L.ref_bvh_stats.argtypes = [vp]
        L.ref_get_triangle.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp]
        self.L = L
        self.depth_variant = depth_variant

    def parse_model(self, obj_path):
        """Model(filename) of the reference alone (model.cpp:44-281), as plain arrays."""
        L = self.L
        L.ref_model_load.argtypes = [C.c_char_p]; L.ref_model_counts.argtypes = [vp]; L.ref_model_arrays.argtypes = [vp, vp, vp, vp]
        L.ref_model_material.argtypes = [C.c_int, vp]; L.ref_model_texels.argtypes = [C.c_int, vp]; L.ref_model_camera.argtypes = [vp]
        L.ref_model_load(obj_path.encode())
        cnt = np.zeros(7, np.int32); L.ref_model_counts(_p(cnt))
        nv, nn, nt, nf, nm, w, h = [int(x) for x in cnt]
        v = np.zeros((nv, 3)); vn = np.zeros((nn, 3)); vt = np.zeros((nt, 2)); f = np.zeros((nf, 3, 4), np.int32)
        L.ref_model_arrays(_p(v), _p(vn), _p(vt), _p(f))
        mats = np.zeros((nm, 14)); texels = []
        for i in range(nm):
            L.ref_model_material(i, _p(mats[i]))
            t = np.zeros((int(mats[i, 11]), 3), np.float32)
            if len(t):
                L.ref_model_texels(i, _p(t))
            texels.append(t)
        cam = np.zeros(10); L.ref_model_camera(_p(cam))
        return dict(vertex=v, normal=vn, texcoord=vt, face=f, materials=mats, texels=texels, camera=cam, width=w, height=h)

    def load(self, obj_path):
        # the reference prints "[Model] path" to stdout (model.cpp:46); harmless
        if self.L.ref_load(obj_path.encode()) != 0:
            raise RuntimeError("reference failed to load " + obj_path)
        self.width, self.height = self.L.ref_width(), self.L.ref_height()

    def set_resolution(self, w, h):
        self.L.ref_set_resolution(w, h); self.width, self.height = w, h

    def set_max_bounces(self, n):
        assert self.depth_variant or n <= 0, "max bounces needs the depth variant"
        self.L.ref_set_max_bounces(n)

    def inject(self, xi):
        self._xi = _f(xi); self.L.ref_rng_mode(1); self.L.ref_rng_inject(_p(self._xi), self._xi.size)

    def stream_mode(self):
        self.L.ref_rng_mode(0)

    def consumed(self):
        return self.L.ref_rng_consumed()

    def underflow(self):
        return self.L.ref_rng_underflow()

    def render(self, frames):
        return self.L.ref_render(frames)

    def clear(self):
        self.L.ref_clear()

    def accum(self):
        a = np.zeros((self.height, self.width, 4), np.float32); self.L.ref_get_accum(_p(a)); return a

    def pixels_u8(self):
        a = np.zeros((self.height, self.width, 3), np.uint8); self.L.ref_get_pixels_u8(_p(a)); return a

    def set_pixel(self, x, y, rgb):
        self.L.ref_set_pixel(x, y, rgb[0], rgb[1], rgb[2])

    def bvh_stats(self):
        o = (C.c_long * 4)(); self.L.ref_bvh_stats(o); return dict(nodes=o[0], leaves=o[1], depth=o[2], max_leaf=o[3])

    def num_tris(self):
        return self.L.ref_num_tris()

    def num_lights(self):
        return self.L.ref_num_lights()

    def light_tri(self, i):
        return self.L.ref_light_tri(i)

    def cast_ray(self, x, y, xi):
        self.inject(xi); o = np.zeros(6); self.L.ref_cast_ray(x, y, _p(o)); return o, self.consumed()

    def aabb_intersect(self, A, B, o, d, t1, t2):
        return self.L.ref_aabb_intersect(_p(_d3(A)), _p(_d3(B)), _p(_d3(o)), _p(_d3(d)), t1, t2)

    def tri_hit(self, v9, vn9, uv6, emissive, o, d, t1, t2):
        out = np.zeros(13)
        h = self.L.ref_tri_hit(_p(_d3(v9)), _p(_d3(vn9)), _p(_d3(uv6)), int(emissive), _p(_d3(o)), _p(_d3(d)), t1, t2, _p(out))
        return h, out

    def tri_any(self, v9, o, d, t1, t2):
        return self.L.ref_tri_any(_p(_d3(v9)), _p(_d3(o)), _p(_d3(d)), t1, t2)

    def tri_area(self, v9):
        return self.L.ref_tri_area(_p(_d3(v9)))

    def bvh_hit(self, o, d, t1=1e-4, t2=np.finfo(np.float64).max):
        out = np.zeros(12); h = self.L.ref_bvh_hit(_p(_d3(o)), _p(_d3(d)), t1, t2, _p(out)); return h, out

    def bvh_has_hit(self, o, d, t1=1e-4, t2=np.finfo(np.float64).max):
        return self.L.ref_bvh_has_hit(_p(_d3(o)), _p(_d3(d)), t1, t2)

    def bsdf_setup(self, n, wi, kd, ks, ns):
        out = np.zeros(18, np.float32)
        self.L.ref_bsdf_setup(_p(_d3(n)), _p(_d3(wi)), _p(_d3(kd)), _p(_d3(ks)), ns, _p(out)); return out

    def bsdf_eval(self, n, wi, kd, ks, ns, wo):
        out = np.zeros(4, np.float32); wo = _f(wo)
        self.L.ref_bsdf_eval(_p(_d3(n)), _p(_d3(wi)), _p(_d3(kd)), _p(_d3(ks)), ns, _p(wo), _p(out)); return out

    def bsdf_sample(self, n, wi, kd, ks, ns, xi):
        self.inject(xi); out = np.zeros(8, np.float32)
        self.L.ref_bsdf_sample(_p(_d3(n)), _p(_d3(wi)), _p(_d3(kd)), _p(_d3(ks)), ns, _p(out)); return out, self.consumed()

    def power_heuristic(self, a, b):
        return self.L.ref_power_heuristic(a, b)

    def texture_get_color(self, img_f32, u, v):
        img = np.ascontiguousarray(img_f32, np.float32); out = np.zeros(3, np.float32)
        self.L.ref_texture_get_color(img.shape[1], img.shape[0], _p(img), u, v, _p(out)); return out

    def sample_light(self, p, xi):
        self.inject(xi); out = np.zeros(14); self.L.ref_sample_light(_p(_d3(p)), _p(out)); return out, self.consumed()

    def trace_path(self, o, d, xi, recursive=False):
        self.inject(xi); L3 = np.zeros(3, np.float32)
        (self.L.ref_trace_path_recursive if recursive else self.L.ref_trace_path)(_p(_d3(o)), _p(_d3(d)), _p(L3))
        return L3, self.consumed()

    def trace_pixel(self, x, y, xi):
        self.inject(xi); L3 = np.zeros(3, np.float32); self.L.ref_trace_pixel(x, y, _p(L3)); return L3, self.consumed()

    def shadow_probe(self, o, d, xi, shrink=1e-4):
        self.inject(xi); out = np.zeros(2, np.int32)
        ok = self.L.ref_shadow_probe(_p(_d3(o)), _p(_d3(d)), shrink, _p(out)); return ok, out
