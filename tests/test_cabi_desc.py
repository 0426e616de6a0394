"""The `rtc_scene_desc` ABI (include/rtc.h) filled by a FOREIGN flattener — tests/foreign_flattener.py: the reference's
construction semantics and the INTEGRATION.md §3 walk restated in Python/ctypes, sharing no code with the product's C++
host mirror — must give a scene the library accepts and renders identically, bit for bit, to the rtw_* path.
CPU: through the emulator of the kernel source (same scene_build.hpp validation + accelerator); GPU: librtc_amd.so, including
the N-device entry points (rtc_multi_*; two and three replicas on the one GPU of the test box)."""
import ctypes as C
import math

import numpy as np
import pytest

import cases
import foreign_flattener as ff
from raytracer_challenge_amd import scenes
from raytracer_challenge_amd.backend import HIT_DTYPE
from raytracer_challenge_amd.scene import (Camera, Color, Element, GroupKind, Material, Matrix, Pattern, PointLight, ShapeArgs, Vector, World)


def smooth_triangle_scene():
    """A fan of smooth triangles under a transformed, material-carrying aggregation group + a flat triangle + a floor."""
    mat = Material(pattern=Pattern.plain(Color.new(0.9, 0.5, 0.3)), reflective=0.2, shininess=50.0)
    tris = []
    n = 10
    for i in range(n):
        a0, a1 = 2 * math.pi * i / n, 2 * math.pi * (i + 1) / n
        p1, p2, p3 = Vector.point(0, 1.2, 0), Vector.point(math.cos(a0), 0.2, math.sin(a0)), Vector.point(math.cos(a1), 0.2, math.sin(a1))
        n1, n2, n3 = Vector.vector(0, 1, 0), Vector.vector(math.cos(a0), 0.4, math.sin(a0)), Vector.vector(math.cos(a1), 0.4, math.sin(a1))
        tris.append(Element.smooth_triangle(ShapeArgs(), p1, p2, p3, n1, n2, n3))
    fan = Element.composite(Matrix.translation(0.2, 0.0, 0.5) * Matrix.rotation_y(0.3) * Matrix.scaling(1.5, 1.2, 1.5), mat, GroupKind.Aggregation, tris)
    flat = Element.triangle(ShapeArgs(material=Material(pattern=Pattern.plain(Color.new(0.2, 0.6, 0.9)))), Vector.point(-3, 0, 2), Vector.point(-1, 2.5, 2.5), Vector.point(-2.5, 0.1, 0.2))
    floor = Element.plane(ShapeArgs(material=Material(pattern=Pattern.checkers(Matrix.id(), Pattern.plain(Color.white()), Pattern.plain(Color.new(0.3, 0.3, 0.3))), reflective=0.3)))
    world = World([PointLight(Color.white(), Vector.point(-5, 8, -6)), PointLight(Color.new(0.3, 0.3, 0.4), Vector.point(6, 5, -2))], [floor, fan, flat])
    cam = Camera.new(96, 54, 1.0, Camera.transform(Vector.point(0.5, 2.5, -6), Vector.point(0, 0.6, 0), Vector.vector(0, 1, 0)))
    return cam, world


DESC_CASES = {
    "default_world": cases.SMALL_CASES["default_world"],
    "nested_groups": cases.SMALL_CASES["nested_groups"],
    "synthetic_cones_grouped": cases.SMALL_CASES["synthetic_cones_grouped"],
    "smooth_triangles": smooth_triangle_scene,
    "patterns_and_noise": cases.SMALL_CASES["patterns_and_noise"],
    "csg_scene": cases.SMALL_CASES["csg_scene"],
    "all_primitives": cases.SMALL_CASES["all_primitives"],
}


def bind(lib):
    vp = C.c_void_p
    lib.rtc_scene_create.restype = C.c_int
    lib.rtc_scene_create.argtypes = [C.POINTER(ff.RtcSceneDesc), C.c_int, C.POINTER(vp)]
    lib.rtc_scene_destroy.restype = None
    lib.rtc_scene_destroy.argtypes = [vp]
    lib.rtc_render.restype = C.c_int
    lib.rtc_render.argtypes = [vp, C.POINTER(ff.RtcCamera), C.c_int32, vp, C.c_uint64, C.c_uint64, vp, vp, vp]
    lib.rtc_last_error.restype = C.c_char_p
    return lib


def render_desc(lib, world, cam, fuel=5):
    """Foreign flattener -> rtc_scene_create -> rtc_render (all pixels, with the primary-hit channel)."""
    bind(lib)
    flat = ff.flatten(world)
    desc = flat.desc()
    scene = C.c_void_p()
    assert lib.rtc_scene_create(C.byref(desc), 0, C.byref(scene)) == 0, lib.rtc_last_error()
    rc = ff.make_camera(cam)
    n = cam.hsize * cam.vsize
    rgb = np.empty((n, 3), dtype=np.float64)
    hits = np.empty(n, dtype=HIT_DTYPE)
    assert lib.rtc_render(scene, C.byref(rc), fuel, None, 0, n, rgb.ctypes.data, hits.ctypes.data, None) == 0, lib.rtc_last_error()
    lib.rtc_scene_destroy(scene)
    return rgb, hits, flat


@pytest.mark.parametrize("name", sorted(DESC_CASES))
def test_foreign_descriptor_renders_identically_on_the_emulated_kernels(name):
    from emu_lib import emu
    be = emu()
    cam, world = DESC_CASES[name]()
    rgb, hits, flat = render_desc(be.lib, world, cam)
    want_rgb, want_hits = be.render(be.build_world(world), cam, 5)
    assert np.array_equal(hits, want_hits), name
    assert np.array_equal(rgb, want_rgb), name
    # and the foreign walk agrees with the product's flattener on the array sizes a Rust shim would hand over
    counts = (C.c_uint32 * 8)()
    be.lib.rtw_world_flatten_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    nw = be.build_world(world)
    assert be.lib.rtw_world_flatten_counts(nw.handle, counts) == 0
    assert (counts[0], counts[1], counts[3], counts[4], counts[7]) == (len(flat.nodes), len(flat.prims), len(flat.limits) // 2, len(flat.tri_geo) // 9, len(flat.lights))


def test_foreign_descriptor_matches_the_oracle(orc):
    """... and the oracle (which has its own scene construction) on the smooth-triangle scene: hits bit-exact, colours <= 1e-5."""
    from emu_lib import emu
    cam, world = smooth_triangle_scene()
    rgb, hits, _ = render_desc(emu().lib, world, cam)
    ref_rgb, ref_hits = orc.render(orc.build_world(world), cam, 5)
    assert np.array_equal(hits, ref_hits)
    assert np.abs(rgb - ref_rgb).max() <= 1e-5


def test_malformed_descriptors_are_refused():
    """rtc_scene_create validates what a foreign flattener can get wrong: indices out of range, a `skip` that leaves the array."""
    from emu_lib import emu
    lib = bind(emu().lib)
    cam, world = cases.SMALL_CASES["nested_groups"]()
    for breaker in ("material", "xform", "skip", "pattern"):
        flat = ff.flatten(world)
        if breaker == "material":
            flat.prims[0].material = len(flat.materials)
        elif breaker == "xform":
            flat.prims[1].xform = -2
        elif breaker == "skip":
            flat.nodes[0].skip = len(flat.nodes) + 5
        else:
            flat.materials[0].pattern = len(flat.pats) + 1
        desc = flat.desc()
        scene = C.c_void_p()
        assert lib.rtc_scene_create(C.byref(desc), 0, C.byref(scene)) == 1, breaker   # RTC_ERR_INVALID
        assert lib.rtc_last_error(), breaker


# ---- on the GPU ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(DESC_CASES))
def test_hip_foreign_descriptor_renders_identically(hip, name):
    cam, world = DESC_CASES[name]()
    rgb, hits, _ = render_desc(hip.lib, world, cam)
    want_rgb, want_hits = hip.render(hip.build_world(world), cam, 5)
    assert np.array_equal(hits, want_hits), name
    assert np.array_equal(rgb, want_rgb), name


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [(0,), (0, 0), (0, 0, 0)])
def test_hip_render_multi_equals_one_device(hip, devices):
    """rtc_multi_*: replicas on the listed devices (the test box has one GPU, so one device is listed several times: every code
    path but the cross-device copy itself runs), rows interleaved by replica, gather + de-interleave on the first device."""
    lib = bind(hip.lib)
    vp = C.c_void_p
    lib.rtc_multi_create.restype = C.c_int
    lib.rtc_multi_create.argtypes = [C.POINTER(ff.RtcSceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.rtc_multi_destroy.restype = None
    lib.rtc_multi_destroy.argtypes = [vp]
    lib.rtc_render_multi.restype = C.c_int
    from raytracer_challenge_amd.device import RtcStatsC
    lib.rtc_render_multi.argtypes = [vp, C.POINTER(ff.RtcCamera), C.c_int32, vp, C.POINTER(RtcStatsC)]
    lib.rtc_render_multi_device.restype = C.c_int
    lib.rtc_render_multi_device.argtypes = [vp, C.POINTER(ff.RtcCamera), C.c_int32, vp, C.c_int]
    lib.rtc_multi_sync.restype = C.c_int
    lib.rtc_multi_sync.argtypes = [vp]
    lib.rtc_multi_device_count.restype = C.c_int
    lib.rtc_multi_device_count.argtypes = [vp]
    import torch
    for name, (hs, vs) in (("synthetic_cones_grouped", (None, None)), ("smooth_triangles", (None, None)), ("nested_glass", (61, 37))):   # 37 rows: uneven split
        cam, world = (DESC_CASES.get(name) or cases.SMALL_CASES[name])()
        if hs:
            cam = Camera.new(hs, vs, cam.field_of_view, cam.transform_matrix)
        want_rgb, _, flat = render_desc(lib, world, cam)
        desc = flat.desc()
        m = vp()
        devs = (C.c_int * len(devices))(*devices)
        assert lib.rtc_multi_create(C.byref(desc), devs, len(devices), C.byref(m)) == 0, lib.rtc_last_error()
        assert lib.rtc_multi_device_count(m) == len(devices)
        rc = ff.make_camera(cam)
        n = cam.hsize * cam.vsize
        rgb = np.full((n, 3), np.nan)
        st = RtcStatsC()
        assert lib.rtc_render_multi(m, C.byref(rc), 5, rgb.ctypes.data, C.byref(st)) == 0, lib.rtc_last_error()
        assert np.array_equal(rgb, want_rgb), (name, devices)
        assert st.pixels == n and st.rays_primary == n
        rgb2 = np.full((n, 3), np.nan)
        assert lib.rtc_render_multi(m, C.byref(rc), 5, rgb2.ctypes.data, None) == 0, lib.rtc_last_error()   # the timed form: no counters
        # the quantised form (each replica quantises its rows, 3 B/px gathered): Color::clamp of the f64 image, byte for byte
        lib.rtc_render_multi_rgb8.restype = C.c_int
        lib.rtc_render_multi_rgb8.argtypes = [vp, C.POINTER(ff.RtcCamera), C.c_int32, vp, C.POINTER(RtcStatsC)]
        rgb8 = np.zeros(rgb.size, dtype=np.uint8)
        assert lib.rtc_render_multi_rgb8(m, C.byref(rc), 5, rgb8.ctypes.data, None) == 0, lib.rtc_last_error()
        c = np.where(np.isnan(rgb.reshape(-1)), 1.0, np.minimum(rgb.reshape(-1), 1.0))
        want8 = np.floor(np.maximum(c, 0.0) * 255.0 + 0.5).astype(np.uint8)   # round half away from zero on non-negative values
        assert np.array_equal(rgb8, want8), (devices, int((rgb8 != want8).sum()))
        assert np.array_equal(rgb2, want_rgb), (name, devices)
        out = torch.full((n * 3,), float("nan"), dtype=torch.float64, device="cuda:0")
        for _ in range(3):                                                                                     # frames queued back to back
            assert lib.rtc_render_multi_device(m, C.byref(rc), 5, C.c_void_p(out.data_ptr()), 0) == 0, lib.rtc_last_error()
        assert lib.rtc_multi_sync(m) == 0, lib.rtc_last_error()
        assert np.array_equal(out.cpu().numpy().reshape(-1, 3), want_rgb), (name, devices)
        # the counters of the replicas' (side by side, asynchronous) counting launches add up to the one-device launch's
        lib.rtc_render.restype = C.c_int
        one = RtcStatsC()
        scene = vp()
        assert lib.rtc_scene_create(C.byref(desc), 0, C.byref(scene)) == 0
        rgb3 = np.empty((n, 3))
        assert lib.rtc_render(scene, C.byref(rc), 5, None, 0, n, rgb3.ctypes.data, None, C.byref(one)) == 0, lib.rtc_last_error()
        lib.rtc_scene_destroy(scene)
        for f in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "rays_container", "analytic_tests_kernarg"):
            assert getattr(st, f) == getattr(one, f), (f, name, devices)
        # other band heights (default 8): single rows interleaved, and a height that leaves a short last band
        lib.rtc_multi_set_band_rows.restype = C.c_int
        lib.rtc_multi_set_band_rows.argtypes = [vp, C.c_uint32]
        for band in (1, 5):
            assert lib.rtc_multi_set_band_rows(m, band) == 0
            rgb4 = np.full((n, 3), np.nan)
            assert lib.rtc_render_multi(m, C.byref(rc), 5, rgb4.ctypes.data, None) == 0, lib.rtc_last_error()
            assert np.array_equal(rgb4, want_rgb), (name, devices, band)
        assert lib.rtc_multi_set_band_rows(m, 0) != 0
        lib.rtc_multi_destroy(m)


@pytest.mark.gpu
def test_hip_render_rgb8_is_color_clamp_of_the_f64_frame(hip):
    """rtc_render_rgb8: the frame quantised on the device (src/color.rs:42-46), 3 bytes per pixel to the host."""
    lib = bind(hip.lib)
    vp = C.c_void_p
    from raytracer_challenge_amd.device import RtcStatsC
    lib.rtc_render_rgb8.restype = C.c_int
    lib.rtc_render_rgb8.argtypes = [vp, C.POINTER(ff.RtcCamera), C.c_int32, vp, C.POINTER(RtcStatsC)]
    for name in ("synthetic_cones_grouped", "nested_glass"):
        cam, world = (DESC_CASES.get(name) or cases.SMALL_CASES[name])()
        rgb, _, flat = render_desc(lib, world, cam)
        desc = flat.desc()
        scene = vp()
        assert lib.rtc_scene_create(C.byref(desc), 0, C.byref(scene)) == 0
        rc = ff.make_camera(cam)
        rgb8 = np.zeros(rgb.size, dtype=np.uint8)
        st = RtcStatsC()
        assert lib.rtc_render_rgb8(scene, C.byref(rc), 5, rgb8.ctypes.data, C.byref(st)) == 0, lib.rtc_last_error()
        lib.rtc_scene_destroy(scene)
        c = np.where(np.isnan(rgb.reshape(-1)), 1.0, np.minimum(rgb.reshape(-1), 1.0))
        want8 = np.floor(np.maximum(c, 0.0) * 255.0 + 0.5).astype(np.uint8)
        assert np.array_equal(rgb8, want8), name
        assert st.pixels == cam.hsize * cam.vsize
        assert lib.rtc_render_rgb8(scene if False else None, C.byref(rc), 5, rgb8.ctypes.data, None) != 0   # NULL scene is refused


@pytest.mark.gpu
def test_hip_render_multi_random_partitions(hip):
    """rtc_render_multi over 1-4 replicas with random image sizes (rows not a multiple of the band, fewer bands than replicas, one-row
    images) and band heights: always the one-device frame, bit for bit; the quantised form too."""
    lib = bind(hip.lib)
    vp = C.c_void_p
    from raytracer_challenge_amd.device import RtcStatsC
    lib.rtc_multi_create.restype = C.c_int
    lib.rtc_multi_create.argtypes = [C.POINTER(ff.RtcSceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.rtc_multi_destroy.restype = None
    lib.rtc_multi_destroy.argtypes = [vp]
    lib.rtc_render_multi.restype = C.c_int
    lib.rtc_render_multi.argtypes = [vp, C.POINTER(ff.RtcCamera), C.c_int32, vp, C.POINTER(RtcStatsC)]
    lib.rtc_render_multi_rgb8.restype = C.c_int
    lib.rtc_render_multi_rgb8.argtypes = [vp, C.POINTER(ff.RtcCamera), C.c_int32, vp, C.POINTER(RtcStatsC)]
    lib.rtc_multi_set_band_rows.restype = C.c_int
    lib.rtc_multi_set_band_rows.argtypes = [vp, C.c_uint32]
    rng = np.random.default_rng(77)
    cam0, world = cases.SMALL_CASES["nested_glass"]()
    flat = ff.flatten(world)
    desc = flat.desc()
    multis = {}
    for n in (1, 2, 3, 4):
        m = vp()
        devs = (C.c_int * n)(*([0] * n))
        assert lib.rtc_multi_create(C.byref(desc), devs, n, C.byref(m)) == 0, lib.rtc_last_error()
        multis[n] = m
    scene = vp()
    assert lib.rtc_scene_create(C.byref(desc), 0, C.byref(scene)) == 0
    for _ in range(40):
        hs, vs = int(rng.integers(1, 70)), int(rng.integers(1, 45))
        n, band = int(rng.integers(1, 5)), int(rng.choice([1, 2, 3, 8, 8, 8, 16, 64]))
        cam = Camera.new(hs, vs, cam0.field_of_view, cam0.transform_matrix)
        rc = ff.make_camera(cam)
        px = hs * vs
        want = np.empty((px, 3))
        assert lib.rtc_render(scene, C.byref(rc), 4, None, 0, px, want.ctypes.data, None, None) == 0, lib.rtc_last_error()
        m = multis[n]
        assert lib.rtc_multi_set_band_rows(m, band) == 0
        got = np.full((px, 3), np.nan)
        assert lib.rtc_render_multi(m, C.byref(rc), 4, got.ctypes.data, None) == 0, lib.rtc_last_error()
        assert np.array_equal(got, want), (hs, vs, n, band)
        got8 = np.zeros(3 * px, dtype=np.uint8)
        assert lib.rtc_render_multi_rgb8(m, C.byref(rc), 4, got8.ctypes.data, None) == 0, lib.rtc_last_error()
        c = np.minimum(want.reshape(-1), 1.0)
        assert np.array_equal(got8, np.floor(np.maximum(c, 0.0) * 255.0 + 0.5).astype(np.uint8)), (hs, vs, n, band)
    lib.rtc_scene_destroy(scene)
    for m in multis.values():
        lib.rtc_multi_destroy(m)
