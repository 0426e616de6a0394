"""`Image` — mirror of the reference's src/image.rs on the HIP backend.

`Image.par_render(camera, world)` is the reference's entry point of the hot path (src/image.rs:65-81); `read`, `write`,
`ppm` follow :83-112.  Quantisation (`Color::clamp`, src/color.rs:42-46) runs on the GPU, the P3 text is formatted by
the library's host code.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .scene import FUEL, Camera, Color, World


class Image:
    def __init__(self, hsize: int, vsize: int, pixels: np.ndarray = None, _native=None):  # src/image.rs:18-24
        self.hsize, self.vsize = int(hsize), int(vsize)
        self.pixels = np.zeros((self.vsize * self.hsize, 3), dtype=np.float64) if pixels is None else pixels
        self._native = _native  # (backend, NativeWorld) that rendered it, for the device quantiser

    @staticmethod
    def par_render(camera: Camera, world: World, fuel: int = FUEL, backend=None) -> "Image":
        from . import hip_backend
        be = backend or hip_backend()
        nw = be.build_world(world)
        rgb, _ = be.render(nw, camera, fuel, want_hits=False)
        return Image(camera.hsize, camera.vsize, rgb, (be, nw))

    def _idx(self, x: int, y: int) -> int:  # :114-116
        return y * self.hsize + x

    def write(self, x: int, y: int, color: Color):  # :83-86
        self.pixels[self._idx(x, y)] = (color.r, color.g, color.b)

    def read(self, x: int, y: int) -> Color:  # :88-91
        r, g, b = self.pixels[self._idx(x, y)]
        return Color(float(r), float(g), float(b))

    def _lib_and_scene(self):
        from . import hip_backend
        from .scene import World as _W
        if self._native is None:
            be = hip_backend()
            self._native = (be, be.build_world(_W([], [])))
        be, nw = self._native
        lib = be.lib
        lib.rtw_world_scene.restype = C.c_void_p
        lib.rtw_world_scene.argtypes = [C.c_void_p, C.c_int]
        scene = lib.rtw_world_scene(nw.handle, 0)
        if not scene:
            raise RuntimeError("scene upload failed: %s" % be._err())
        return lib, scene

    def quantized(self) -> np.ndarray:
        """Color::clamp of every channel, on the GPU: (vsize*hsize, 3) uint8."""
        lib, scene = self._lib_and_scene()
        lib.rtc_quantize.restype = C.c_int
        lib.rtc_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        src = np.ascontiguousarray(self.pixels, dtype=np.float64)
        out = np.empty(src.shape, dtype=np.uint8)
        if lib.rtc_quantize(scene, src.ctypes.data, src.size, out.ctypes.data) != 0:
            lib.rtc_last_error.restype = C.c_char_p
            raise RuntimeError("rtc_quantize: %s" % (lib.rtc_last_error() or b"").decode())
        return out

    def ppm(self) -> str:  # :93-112
        return ppm_text(self.hsize, self.vsize, self.quantized())


def ppm_text(hsize: int, vsize: int, rgb8: np.ndarray, lib=None) -> str:
    """Image::ppm layout from quantised pixels (host-side formatting in librtc_amd.so; needs no GPU)."""
    if lib is None:
        from . import _LIB
        lib = C.CDLL(_LIB)
    lib.rtc_ppm.restype = C.c_uint64
    lib.rtc_ppm.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_char_p, C.c_uint64]
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    assert rgb8.size == hsize * vsize * 3
    need = lib.rtc_ppm(hsize, vsize, rgb8.ctypes.data, None, 0)
    buf = C.create_string_buffer(int(need) + 1)
    lib.rtc_ppm(hsize, vsize, rgb8.ctypes.data, buf, need + 1)
    return buf.value.decode()
