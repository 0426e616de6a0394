"""Device-resident entry points of include/rtc.h for harnesses that keep pixels in HBM (bench.py, multi-GPU).

PyTorch is used here only as plumbing: it owns the output buffer (a CUDA/HIP tensor whose ``data_ptr()`` is handed to
``rtc_render_rows_device``) and, in bench.py, the RCCL process group.  All rendering is done by librtc_amd.so.
"""
from __future__ import annotations

import ctypes as C

from .backend import Backend, CameraC, NativeWorld, RtwError
from .scene import Camera


class RtcCameraC(C.Structure):
    _fields_ = [("hsize", C.c_uint64), ("vsize", C.c_uint64), ("half_width", C.c_double), ("half_height", C.c_double),
                ("pixel_size", C.c_double), ("transform_inv", C.c_double * 16)]


class RtcStatsC(C.Structure):
    _COUNTERS = ("pixels", "rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "rays_container",
                 "accel_nodes", "group_tests", "tri_tests", "analytic_tests", "nan_ts")
    _fields_ = [(n, C.c_uint64) for n in _COUNTERS] + \
               [("kernel_ms", C.c_double), ("n_launches", C.c_uint32), ("_pad", C.c_uint32),
                ("accel_nodes_kernarg", C.c_uint64), ("analytic_tests_kernarg", C.c_uint64)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n in self._COUNTERS + ("accel_nodes_kernarg", "analytic_tests_kernarg")}
        d["kernel_ms"] = float(self.kernel_ms)
        d["n_launches"] = int(self.n_launches)
        d["unique_rays"] = d["rays_primary"] + d["rays_shadow"] + d["rays_reflect"] + d["rays_refract"]
        return d


# algorithmic bytes per unit of work
BYTES_RAY = 64 + 32          # ray in + hit out
BYTES_NODE = 128             # 4-wide accelerator node (a reference group box test is charged the same)
BYTES_TRI = 72               # p1, e1, e2 (f64)
BYTES_ANALYTIC = 112         # 3x4 f64 matrix + params
BYTES_PIXEL = 24             # framebuffer write (3 x f64)


def algorithmic_bytes(st: dict) -> int:
    """Bytes one launch must move by the accounting of SURVEY.md §8(d), from the kernel's own work counters."""
    rays = st["unique_rays"]
    return (BYTES_RAY * rays + BYTES_NODE * (st["accel_nodes"] + st["group_tests"]) + BYTES_TRI * st["tri_tests"]
            + BYTES_ANALYTIC * st["analytic_tests"] + BYTES_PIXEL * st["pixels"])


class DeviceRenderer:
    """One flattened+uploaded scene on one GPU plus a camera; renders interleaved rows into caller-owned HBM."""

    def __init__(self, backend: Backend, world: NativeWorld, camera: Camera, device: int = 0):
        if backend.name != "hip":
            raise RtwError("DeviceRenderer needs the HIP backend, got %r" % backend.name)
        lib = backend.lib
        lib.rtw_world_scene.restype = C.c_void_p
        lib.rtw_world_scene.argtypes = [C.c_void_p, C.c_int]
        lib.rtw_make_camera.restype = C.c_int
        lib.rtw_make_camera.argtypes = [C.POINTER(CameraC), C.POINTER(RtcCameraC)]
        lib.rtc_render_rows_device.restype = C.c_int
        lib.rtc_render_rows_device.argtypes = [C.c_void_p, C.POINTER(RtcCameraC), C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.c_void_p, C.POINTER(RtcStatsC), C.c_int, C.c_int]
        lib.rtc_scene_sync.restype = C.c_int
        lib.rtc_scene_sync.argtypes = [C.c_void_p]
        for name in ("rtc_scene_record", "rtc_scene_wait"):
            getattr(lib, name).restype = C.c_int
            getattr(lib, name).argtypes = [C.c_void_p, C.c_int]
        lib.rtc_scene_check.restype = C.c_int
        lib.rtc_scene_check.argtypes = [C.c_void_p]
        lib.rtc_scene_elapsed_ms.restype = C.c_int
        lib.rtc_scene_elapsed_ms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        lib.rtc_last_error.restype = C.c_char_p
        lib.rtc_scene_device_bytes.restype = C.c_uint64
        lib.rtc_scene_device_bytes.argtypes = [C.c_void_p]
        lib.rtc_scene_accel_info.restype = None
        lib.rtc_scene_accel_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 4
        self.backend, self.world, self.camera = backend, world, camera
        self.scene = lib.rtw_world_scene(world.handle, int(device))
        if not self.scene:
            raise RtwError("scene upload failed: %s" % backend._err())
        self.cam = RtcCameraC()
        cc = backend.camera_c(camera)
        if lib.rtw_make_camera(C.byref(cc), C.byref(self.cam)) != 0:
            raise RtwError("camera: %s" % backend._err())

    def render_rows(self, fuel: int, row_first: int, row_step: int, n_rows: int, out_tensor, count: bool = False, sync: bool = True,
                    want_stats: bool = True) -> dict:
        """Rows row_first, row_first+row_step, ... (n_rows) -> out_tensor (float64, n_rows*hsize*3, on this scene's GPU)."""
        need = n_rows * self.camera.hsize * 3
        if out_tensor.numel() < need or out_tensor.element_size() != 8 or not out_tensor.is_cuda:
            raise RtwError("output tensor must be a float64 device tensor with >= %d elements" % need)
        st = RtcStatsC()
        rc = self.backend.lib.rtc_render_rows_device(self.scene, C.byref(self.cam), int(fuel), int(row_first), int(row_step), int(n_rows),
                                                     C.c_void_p(out_tensor.data_ptr()), C.byref(st) if want_stats else None, 1 if count else 0, 1 if sync else 0)
        if rc != 0:
            raise RtwError("rtc_render_rows_device: %s" % (self.backend.lib.rtc_last_error() or b"").decode())
        return st.as_dict() if want_stats else {}

    def _rc(self, rc, what):
        if rc != 0:
            raise RtwError("%s: %s" % (what, (self.backend.lib.rtc_last_error() or b"").decode()))

    def render_rows_async(self, fuel: int, row_first: int, row_step: int, n_rows: int, out_tensor):
        """Queue a render on the scene's stream and return immediately (errors are reported by check())."""
        need = n_rows * self.camera.hsize * 3
        if out_tensor.numel() < need or out_tensor.element_size() != 8 or not out_tensor.is_cuda:
            raise RtwError("output tensor must be a float64 device tensor with >= %d elements" % need)
        self._rc(self.backend.lib.rtc_render_rows_device(self.scene, C.byref(self.cam), int(fuel), int(row_first), int(row_step), int(n_rows),
                                                        C.c_void_p(out_tensor.data_ptr()), None, 0, 0), "rtc_render_rows_device")

    def record(self, slot: int):
        self._rc(self.backend.lib.rtc_scene_record(self.scene, slot), "rtc_scene_record")

    def wait(self, slot: int):
        self._rc(self.backend.lib.rtc_scene_wait(self.scene, slot), "rtc_scene_wait")

    def elapsed_ms(self, slot_from: int, slot_to: int) -> float:
        ms = C.c_double(0.0)
        self._rc(self.backend.lib.rtc_scene_elapsed_ms(self.scene, slot_from, slot_to, C.byref(ms)), "rtc_scene_elapsed_ms")
        return ms.value

    def check(self):
        self._rc(self.backend.lib.rtc_scene_check(self.scene), "rtc_scene_check")

    def sync(self):
        self.backend.lib.rtc_scene_sync(self.scene)

    def tune(self, fuel: int, row_first: int, row_step: int, n_rows: int, out_tensor) -> dict:
        """Lets the library measure both device paths for this launch shape (four synchronous renders into out_tensor, two
        per path) so that later launches, including unsynchronised ones, take the faster; returns path_info()."""
        for _ in range(4):
            self.render_rows(fuel, row_first, row_step, n_rows, out_tensor, count=False, sync=True, want_stats=False)
        return self.path_info()

    def path_info(self) -> dict:
        lib = self.backend.lib
        lib.rtc_scene_path_info.restype = None
        lib.rtc_scene_path_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        ch, a, b = C.c_int32(0), C.c_double(-1), C.c_double(-1)
        lib.rtc_scene_path_info(self.scene, C.byref(ch), C.byref(a), C.byref(b))
        return {"path": {0: "undecided", 1: "one kernel", 4: "wavefront"}.get(ch.value, str(ch.value)),
                "one_kernel_ms": a.value, "wavefront_ms": b.value}

    def info(self) -> dict:
        a = [C.c_uint32(0) for _ in range(4)]
        self.backend.lib.rtc_scene_accel_info(self.scene, *[C.byref(x) for x in a])
        return {"scene_device_bytes": int(self.backend.lib.rtc_scene_device_bytes(self.scene)), "program_ops": a[0].value,
                "bvh_nodes": a[1].value, "mesh_triangles": a[2].value, "bvh_depth": a[3].value}
