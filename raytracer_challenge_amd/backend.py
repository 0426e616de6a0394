"""ctypes binding of include/rtw.h — generic over the library that implements it.

``Backend(path)`` loads one shared library exporting the ``rtw_*`` symbols and turns the pure-Python
scene description (:mod:`raytracer_challenge_amd.scene`) into native handles.  The package itself only
ever loads its own HIP library (:func:`raytracer_challenge_amd.hip_backend`); tests load the CPU oracle
through this same class from ``tests/`` — the product never does.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from .scene import (FUEL, GEOMETRY, GROUP_KINDS, JITTER_KINDS, MIXTURE_KINDS, Camera, Element, Material, Pattern, World)

HIT_DTYPE = np.dtype([("t", "<f8"), ("prim", "<i4"), ("push_idx", "<i4")])


class RtwError(RuntimeError):
    pass


class _MaterialC(C.Structure):
    _fields_ = [("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double), ("shininess", C.c_double),
                ("reflective", C.c_double), ("transparency", C.c_double), ("refractive_index", C.c_double),
                ("pattern", C.c_void_p)]


class CameraC(C.Structure):
    _fields_ = [("hsize", C.c_uint64), ("vsize", C.c_uint64), ("field_of_view", C.c_double), ("transform", C.c_double * 16)]


RTW_SYMBOLS = [
    "rtw_last_error", "rtw_backend", "rtw_pattern_debug", "rtw_pattern_plain", "rtw_pattern_jitter", "rtw_pattern_mixture",
    "rtw_pattern_release", "rtw_shape", "rtw_composite", "rtw_parse_obj", "rtw_element_release", "rtw_world_create",
    "rtw_world_add_light", "rtw_world_add_element", "rtw_world_primitive_count", "rtw_world_release", "rtw_render", "rtw_color_at",
]


def _d16(m) -> C.Array:
    return (C.c_double * 16)(*m.flat())


class NativeWorld:
    """Owns one ``rtw_world*``."""

    def __init__(self, backend: "Backend", handle: int, n_lights: int):
        self.backend, self.handle, self.n_lights = backend, handle, n_lights

    @property
    def primitive_count(self) -> int:
        return int(self.backend.lib.rtw_world_primitive_count(self.handle))

    def close(self):
        if self.handle:
            self.backend.lib.rtw_world_release(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Backend:
    def __init__(self, lib_path: str):
        if not os.path.exists(lib_path):
            raise RtwError("native library not found: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % lib_path)
        self.path = lib_path
        self.lib = lib = C.CDLL(lib_path)
        vp, d, i, u64, sz = C.c_void_p, C.c_double, C.c_int, C.c_uint64, C.c_size_t
        dp = C.POINTER(C.c_double)
        sig = {
            "rtw_last_error": (C.c_char_p, []),
            "rtw_backend": (C.c_char_p, []),
            "rtw_pattern_debug": (vp, []),
            "rtw_pattern_plain": (vp, [d, d, d]),
            "rtw_pattern_jitter": (vp, [i, i, d, u64, vp]),
            "rtw_pattern_mixture": (vp, [i, dp, vp, vp]),
            "rtw_pattern_release": (None, [vp]),
            "rtw_shape": (vp, [i, dp, C.POINTER(_MaterialC), i, dp, sz]),
            "rtw_composite": (vp, [dp, C.POINTER(_MaterialC), i, C.POINTER(vp), sz]),
            "rtw_parse_obj": (vp, [C.c_char_p, dp, C.POINTER(_MaterialC), C.POINTER(u64), C.POINTER(u64)]),
            "rtw_element_release": (None, [vp]),
            "rtw_world_create": (vp, []),
            "rtw_world_add_light": (i, [vp, dp, dp]),
            "rtw_world_add_element": (i, [vp, vp]),
            "rtw_world_primitive_count": (u64, [vp]),
            "rtw_world_release": (None, [vp]),
            "rtw_render": (i, [vp, C.POINTER(CameraC), i, vp, u64, vp, vp]),
            "rtw_color_at": (i, [vp, vp, u64, i, vp, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)  # AttributeError = missing export: fail loudly
            fn.restype, fn.argtypes = res, args
        self.name = lib.rtw_backend().decode()

    # ---- errors
    def _err(self) -> str:
        return (self.lib.rtw_last_error() or b"").decode()

    def _check_ptr(self, p, what):
        if not p:
            raise RtwError("%s failed: %s" % (what, self._err()))
        return p

    def _check(self, rc, what):
        if rc != 0:
            raise RtwError("%s failed: %s" % (what, self._err()))

    # ---- scene -> handles
    def _pattern(self, p: Pattern, cache: Dict[int, int], owned: list) -> int:
        key = id(p)
        if key in cache:
            return cache[key]
        lib = self.lib
        if p.tag == "debug":
            h = lib.rtw_pattern_debug()
        elif p.tag == "plain":
            h = lib.rtw_pattern_plain(p.color.r, p.color.g, p.color.b)
        elif p.tag == "jitter":
            child = self._pattern(p.left, cache, owned)
            h = lib.rtw_pattern_jitter(JITTER_KINDS[p.kind], 1 if p.noise.kind == "fractal" else 0, p.noise.scale, p.noise.octaves, child)
        elif p.tag == "mixture":
            l, r = self._pattern(p.left, cache, owned), self._pattern(p.right, cache, owned)
            h = lib.rtw_pattern_mixture(MIXTURE_KINDS[p.kind], _d16(p.transform), l, r)
        else:
            raise RtwError("unknown pattern tag %r" % (p.tag,))
        self._check_ptr(h, "pattern")
        cache[key] = h
        owned.append(h)
        return h

    def _material(self, m: Material, cache, owned) -> _MaterialC:
        return _MaterialC(m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, m.transparency, m.refractive_index,
                          self._pattern(m.pattern, cache, owned))

    def _element(self, e: Element, cache, owned) -> int:
        lib = self.lib
        if e.tag == "shape":
            mat = self._material(e.args.material, cache, owned)
            params = (C.c_double * max(1, len(e.params)))(*e.params)
            h = lib.rtw_shape(GEOMETRY[e.geometry], _d16(e.args.transform), C.byref(mat), 1 if e.args.casts_shadow else 0, params, len(e.params))
            return self._check_ptr(h, "shape")
        if e.tag == "composite":
            kids = [self._element(c, cache, owned) for c in e.children]
            arr = (C.c_void_p * max(1, len(kids)))(*kids)
            mat = self._material(e.material, cache, owned) if e.material is not None else None
            h = lib.rtw_composite(_d16(e.transform), C.byref(mat) if mat is not None else None, GROUP_KINDS[e.kind], arr, len(kids))
            return self._check_ptr(h, "composite")
        if e.tag == "obj":
            mat = self._material(e.material, cache, owned)
            ign, tris = C.c_uint64(0), C.c_uint64(0)
            h = lib.rtw_parse_obj(e.path.encode(), _d16(e.transform), C.byref(mat), C.byref(ign), C.byref(tris))
            return self._check_ptr(h, "parse_obj(%s)" % e.path)
        raise RtwError("unknown element tag %r" % (e.tag,))

    def build_world(self, world: World) -> NativeWorld:
        lib = self.lib
        w = self._check_ptr(lib.rtw_world_create(), "world_create")
        cache: Dict[int, int] = {}
        owned: list = []
        try:
            for l in world.lights:
                inten = (C.c_double * 3)(l.intensity.r, l.intensity.g, l.intensity.b)
                org = (C.c_double * 3)(*l.origin[:3])
                self._check(lib.rtw_world_add_light(w, inten, org), "add_light")
            for e in world.elements:
                self._check(lib.rtw_world_add_element(w, self._element(e, cache, owned)), "add_element")
        except Exception:
            lib.rtw_world_release(w)
            raise
        finally:
            for h in owned:
                lib.rtw_pattern_release(h)
        return NativeWorld(self, w, len(world.lights))

    # ---- the path
    @staticmethod
    def camera_c(camera: Camera) -> CameraC:
        c = CameraC()
        c.hsize, c.vsize, c.field_of_view = camera.hsize, camera.vsize, camera.field_of_view
        c.transform = _d16(camera.transform_matrix)
        return c

    def render(self, nw: NativeWorld, camera: Camera, fuel: int = FUEL, pixel_indices: Optional[np.ndarray] = None, want_hits: bool = True):
        """Image::par_render over all pixels (row-major) or the listed pixel indices.  Returns (rgb[n,3], hits[n])."""
        cam = self.camera_c(camera)
        if pixel_indices is None:
            n, idx_p = camera.hsize * camera.vsize, None
        else:
            pixel_indices = np.ascontiguousarray(pixel_indices, dtype=np.uint64)
            n, idx_p = pixel_indices.size, pixel_indices.ctypes.data
        rgb = np.empty((n, 3), dtype=np.float64)
        hits = np.empty(n, dtype=HIT_DTYPE) if want_hits else None
        self._check(self.lib.rtw_render(nw.handle, C.byref(cam), int(fuel), idx_p, n, rgb.ctypes.data, hits.ctypes.data if want_hits else None), "render")
        return rgb, hits

    def render_digest(self, nw: NativeWorld, camera: Camera, fuel: int = FUEL, pixel_indices: Optional[np.ndarray] = None, device: int = 0) -> np.ndarray:
        """include/rtc.h rtc_render_hit_digest: per pixel, the digest of every closest hit of its ray tree (parity channel)."""
        lib = self.lib
        lib.rtw_world_scene.restype = C.c_void_p
        lib.rtw_world_scene.argtypes = [C.c_void_p, C.c_int]
        lib.rtw_make_camera.restype = C.c_int
        lib.rtw_make_camera.argtypes = [C.c_void_p, C.c_void_p]
        lib.rtc_render_hit_digest.restype = C.c_int
        lib.rtc_render_hit_digest.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        lib.rtc_last_error.restype = C.c_char_p
        scene = lib.rtw_world_scene(nw.handle, int(device))
        if not scene:
            raise RtwError("scene upload failed: %s" % self._err())
        cc = self.camera_c(camera)
        rc_cam = (C.c_double * 21)()   # rtc_camera: 2 x u64 + 3 + 16 doubles
        if lib.rtw_make_camera(C.byref(cc), C.byref(rc_cam)) != 0:
            raise RtwError("camera: %s" % self._err())
        if pixel_indices is None:
            n, idx_p = camera.hsize * camera.vsize, None
        else:
            pixel_indices = np.ascontiguousarray(pixel_indices, dtype=np.uint64)
            n, idx_p = pixel_indices.size, pixel_indices.ctypes.data
        out = np.empty(n, dtype=np.uint64)
        if lib.rtc_render_hit_digest(scene, C.byref(rc_cam), int(fuel), idx_p, 0, n, out.ctypes.data) != 0:
            raise RtwError("rtc_render_hit_digest: %s" % (lib.rtc_last_error() or b"").decode())
        return out

    def color_at(self, nw: NativeWorld, rays: np.ndarray, fuel: int = FUEL):
        """World::color_at for rays given as rows {ox,oy,oz,dx,dy,dz}.  Returns (rgb[n,3], hits[n])."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        rgb = np.empty((n, 3), dtype=np.float64)
        hits = np.empty(n, dtype=HIT_DTYPE)
        self._check(self.lib.rtw_color_at(nw.handle, rays.ctypes.data, n, int(fuel), rgb.ctypes.data, hits.ctypes.data), "color_at")
        return rgb, hits
