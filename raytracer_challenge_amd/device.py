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
                ("accel_nodes_kernarg", C.c_uint64), ("analytic_tests_kernarg", C.c_uint64), ("light_grid_cells", C.c_uint64),
                ("group_tests_uniform", C.c_uint64)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n in self._COUNTERS + ("accel_nodes_kernarg", "analytic_tests_kernarg", "light_grid_cells", "group_tests_uniform")}
        d["kernel_ms"] = float(self.kernel_ms)
        d["n_launches"] = int(self.n_launches)
        d["unique_rays"] = d["rays_primary"] + d["rays_shadow"] + d["rays_reflect"] + d["rays_refract"]
        return d


# Algorithmic bytes per unit of work (with the record sizes this build really uses).  Only
# records that the kernels fetch or store with vector memory instructions count; what travels in the kernel arguments (scalar
# loads: plane records, each BVH's root node) is reported separately and counted as 0 bytes.
UNIT_BYTES = {
    "ray": 96,        # wavefront path only: 64 B queue record in + 32 B hit record out per unique ray (the one-kernel path keeps rays in registers)
    "node": 128,      # one 4-wide accelerator node (DBvhNode4)
    "group_box": 48,  # one reference group box (BoundingBox::intersects operands)
    "triangle": 72,   # p1, e1, e2 (f64)
    "analytic": 128,  # one intersection record (DPrimI: rows 0-2 of transform_inv + limits + tags)
    "pixel": 24,      # framebuffer write (3 x f64)
    "light_cell": 16, # one light-grid lookup of a shadow ray: two 4-B cell offsets + on average two 4-B candidate references
}
KERNARG_BYTES = {"node": 128, "plane": 48, "group_box": 48}


def algorithmic_bytes(st: dict, path: str, n_prims: int = 0, lds_tables: bool = False) -> dict:
    """Bytes one launch must move through the memory system by the accounting above, from the kernel's own work counters
    (rtc_stats of the counting variant).  `memory` is what roofline.achieved uses: never more than 4x the ideal
    one-descent figure (guards against a bad accelerator).  lds_tables: the wavefront traversal kernel keeps this
    scene's nodes / records / mesh triangles in LDS (rtc_scene_wavefront_lds_bytes > 0): those fetches move no bytes through the
    memory system either and are reported under `on_chip` beside the kernel-argument-resident ones."""
    import math
    wavefront = "wavefront" in path
    lds = bool(lds_tables) and wavefront
    node_b = UNIT_BYTES["node"] * (st["accel_nodes"] - st["accel_nodes_kernarg"])
    tri_b = UNIT_BYTES["triangle"] * st["tri_tests"]
    ana_b = UNIT_BYTES["analytic"] * (st["analytic_tests"] - st["analytic_tests_kernarg"])
    by_unit = {
        "ray": UNIT_BYTES["ray"] * st["unique_rays"] if wavefront else 0,
        "node": 0 if lds else node_b,
        # gates named by a kernel-argument program test ONE box per wave (scalar loads of a uniform address): kernarg class, 0 bytes
        "group_box": UNIT_BYTES["group_box"] * (st["group_tests"] - st.get("group_tests_uniform", 0)),
        "triangle": 0 if lds else tri_b,
        "analytic": 0 if lds else ana_b,
        "pixel": UNIT_BYTES["pixel"] * st["pixels"],
        "light_cell": UNIT_BYTES["light_cell"] * st.get("light_grid_cells", 0),
    }
    counted = sum(by_unit.values())
    ideal = 96 + 64 * math.ceil(math.log2(max(2, n_prims))) + 72 * 4
    cap = 4 * ideal * st["unique_rays"] + UNIT_BYTES["pixel"] * st["pixels"]
    return {"memory": min(counted, cap), "counted": counted, "cap_4x_ideal": cap,
            "kernarg": KERNARG_BYTES["node"] * st["accel_nodes_kernarg"] + KERNARG_BYTES["plane"] * st["analytic_tests_kernarg"]
                       + KERNARG_BYTES["group_box"] * st.get("group_tests_uniform", 0),
            "lds": (node_b + tri_b + ana_b) if lds else 0,
            "by_unit": by_unit, "unit_bytes": UNIT_BYTES}


class DeviceRenderer:
    """One flattened+uploaded scene on one GPU plus a camera; renders interleaved rows into caller-owned HBM."""

    def __init__(self, backend: Backend, world: NativeWorld, camera: Camera, device: int = 0, _cpu_standin: bool = False):
        # _cpu_standin: tests only (tests/test_bench_cpu.py): the CPU emulator of the kernel source behind the same rtc_* symbols,
        # host tensors as "device" buffers, to rehearse the multi-rank plumbing without a GPU.  Never set by the product.
        self._cpu_standin = bool(_cpu_standin) and backend.name == "hip-emu-cpu"
        if backend.name != "hip" and not self._cpu_standin:
            raise RtwError("DeviceRenderer needs the HIP backend, got %r" % backend.name)
        lib = backend.lib
        lib.rtw_world_scene.restype = C.c_void_p
        lib.rtw_world_scene.argtypes = [C.c_void_p, C.c_int]
        lib.rtw_make_camera.restype = C.c_int
        lib.rtw_make_camera.argtypes = [C.POINTER(CameraC), C.POINTER(RtcCameraC)]
        lib.rtc_render_bands_device.restype = C.c_int
        lib.rtc_render_bands_device.argtypes = [C.c_void_p, C.POINTER(RtcCameraC), C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
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
        lib.rtc_scene_wavefront_lds_bytes.restype = C.c_uint32
        lib.rtc_scene_wavefront_lds_bytes.argtypes = [C.c_void_p]
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
                    want_stats: bool = True, band_rows: int = 1) -> dict:
        """The first n_rows rows of the dense tile of part `row_first` of `row_step` -> out_tensor (float64, n_rows*hsize*3, on this
        scene's GPU).  band_rows = 1: image rows row_first, row_first+row_step, ...; band_rows = B: bands of B rows dealt out the
        same way (include/rtc.h rtc_render_bands_device; parallel.BAND_ROWS = 8 is the multi-GPU partition)."""
        need = n_rows * self.camera.hsize * 3
        if out_tensor.numel() < need or out_tensor.element_size() != 8 or not (out_tensor.is_cuda or self._cpu_standin):
            raise RtwError("output tensor must be a float64 device tensor with >= %d elements" % need)
        st = RtcStatsC()
        rc = self.backend.lib.rtc_render_bands_device(self.scene, C.byref(self.cam), int(fuel), int(band_rows), int(row_first), int(row_step), int(n_rows),
                                                      C.c_void_p(out_tensor.data_ptr()), C.byref(st) if want_stats else None, 1 if count else 0, 1 if sync else 0)
        if rc != 0:
            raise RtwError("rtc_render_bands_device: %s" % (self.backend.lib.rtc_last_error() or b"").decode())
        return st.as_dict() if want_stats else {}

    def _rc(self, rc, what):
        if rc != 0:
            raise RtwError("%s: %s" % (what, (self.backend.lib.rtc_last_error() or b"").decode()))

    def render_rows_async(self, fuel: int, row_first: int, row_step: int, n_rows: int, out_tensor, band_rows: int = 1):
        """Queue a render on the scene's stream and return immediately (errors are reported by check())."""
        need = n_rows * self.camera.hsize * 3
        if out_tensor.numel() < need or out_tensor.element_size() != 8 or not (out_tensor.is_cuda or self._cpu_standin):
            raise RtwError("output tensor must be a float64 device tensor with >= %d elements" % need)
        self._rc(self.backend.lib.rtc_render_bands_device(self.scene, C.byref(self.cam), int(fuel), int(band_rows), int(row_first), int(row_step), int(n_rows),
                                                         C.c_void_p(out_tensor.data_ptr()), None, 0, 0), "rtc_render_bands_device")

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

    def tune(self, fuel: int, row_first: int, row_step: int, n_rows: int, out_tensor, band_rows: int = 1) -> dict:
        """Lets the library measure both device paths for this launch shape (four synchronous renders into out_tensor, two
        per path) so that later launches, including unsynchronised ones, take the faster; returns path_info()."""
        for _ in range(4):
            self.render_rows(fuel, row_first, row_step, n_rows, out_tensor, count=False, sync=True, want_stats=False, band_rows=band_rows)
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
                "bvh_nodes": a[1].value, "mesh_triangles": a[2].value, "bvh_depth": a[3].value,
                "wavefront_lds_bytes_per_block": int(self.backend.lib.rtc_scene_wavefront_lds_bytes(self.scene))}
