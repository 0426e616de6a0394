"""raytracer_challenge_amd — MI355X-native back end for the reference's `Image::par_render -> World::color_at` path.

Layout:
  scene.py    host mirror of the reference's scene API (Matrix, Material, Pattern, Element, World, Camera ...)
  scenes.py   the reference's scene programs + the BASELINE synthetic scenes, as data
  backend.py  ctypes binding of include/rtw.h
  image.py    Image::par_render / read / ppm on the HIP backend
  csrc/       hand-written HIP kernels (gfx950) + the C ABI (include/rtc.h, include/rtw.h) -> librtc_amd.so

The product path is the HIP library only: importing :func:`hip_backend` fails loudly if it has not been built.
"""
from .scene import *  # noqa: F401,F403
from .scene import EPSILON, FUEL  # noqa: F401
from .backend import Backend, RtwError, HIT_DTYPE  # noqa: F401

import os as _os

# RTC_AMD_LIB lets a tuning run point at another build of the same HIP library (csrc/variants/); never a CPU library.
_LIB = _os.environ.get("RTC_AMD_LIB") or _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "csrc", "librtc_amd.so")
_backend = None


def hip_backend() -> Backend:
    """The one product backend (librtc_amd.so: flatten -> HIP kernels).  No CPU fallback exists."""
    global _backend
    if _backend is None:
        # PyTorch-ROCm ships its own libamdhip64; whichever copy is loaded first serves the whole process, and torch
        # cannot see the GPU through the system copy.  Load torch's first so both share one HIP runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _backend = Backend(_LIB)
        if _backend.name != "hip":
            raise RtwError("%s is not the HIP backend (reports %r)" % (_LIB, _backend.name))
    return _backend
