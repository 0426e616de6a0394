"""CPU tests of the product's host logic and kernel *source* (no GPU):
  * the kernel source compiled for the CPU (tests/cpu_emu) against the oracle — logic parity, every case of tests/cases.py;
  * flatten/validation/error behaviour of the host mirror;
  * librtc_amd.so loads and exports every symbol include/*.h declares; compute entry points fail loudly without a device.
The GPU parity tests proper are in test_parity_gpu.py (-m gpu)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import raytracer_challenge_amd as rt
from raytracer_challenge_amd import scenes
from raytracer_challenge_amd.scene import Color, Element, GroupKind, Material, Matrix, Pattern, PointLight, ShapeArgs, Vector, World, Camera
import cases
from parity import assert_parity, assert_ray_parity, assert_ray_parity_with_panics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    from emu_lib import emu as _emu
    return _emu()


@pytest.mark.parametrize("name", sorted(cases.SMALL_CASES))
def test_emulated_kernel_matches_oracle(emu, orc, name):
    cam, world = cases.SMALL_CASES[name]()
    assert_parity(emu, orc, world, cam, 5, label=name)


@pytest.mark.parametrize("version", ["4"])
def test_emulated_other_kernel_versions(emu, orc, version, monkeypatch):
    """RTC_KERNEL=4 (wavefront: per-level trace / shade / shadow / reduce kernels over ray queues) through the same emulator."""
    monkeypatch.setenv("RTC_KERNEL", version)
    for name in ("synthetic_cones_grouped", "teapot_low", "nested_glass", "cube_lattice", "synthetic_mesh_small", "csg_scene", "patterns_and_noise"):
        cam, world = cases.SMALL_CASES[name]()
        assert_parity(emu, orc, world, cam, 5, label="kernel v%s %s" % (version, name))


def test_parallel_bvh_build_equals_serial(emu, orc, monkeypatch, tmp_path):
    """The accelerator build splits large ranges over threads on private vectors and appends them in serial order: same
    nodes, same leaf order, same pixels as the one-thread build (RTC_BUILD_THREADS)."""
    import ctypes as C
    obj = str(tmp_path / "hf.obj")
    cam, world = scenes.synthetic_mesh(obj, nx=160, nz=160, hsize=64, vsize=36)   # 50 562 triangles: two levels of threads
    idx = np.arange(0, cam.hsize * cam.vsize, 7, dtype=np.uint64)
    lib = emu.lib
    lib.rtw_world_scene.restype = C.c_void_p
    lib.rtw_world_scene.argtypes = [C.c_void_p, C.c_int]
    out = {}
    for threads in ("1", "4"):
        monkeypatch.setenv("RTC_BUILD_THREADS", threads)
        nw = emu.build_world(world)
        scene = lib.rtw_world_scene(nw.handle, 0)
        info = [C.c_uint32(0) for _ in range(4)]
        lib.rtc_scene_accel_info(C.c_void_p(scene), *[C.byref(x) for x in info])
        rgb, hits = emu.render(nw, cam, 3, pixel_indices=idx)
        out[threads] = ([x.value for x in info], rgb.copy(), hits.copy())
    assert out["1"][0] == out["4"][0]
    assert np.array_equal(out["1"][1], out["4"][1]) and np.array_equal(out["1"][2], out["4"][2])


def test_wavefront_path_really_runs_in_the_emulator(emu, monkeypatch):
    """rtc_stats.n_launches tells the paths apart: 1 for the one-kernel path, 2 fuel + 4 for the wavefront path (also after
    its queues were grown: the glass scene at fuel 8 needs that)."""
    import ctypes as C
    from raytracer_challenge_amd.device import RtcCameraC, RtcStatsC
    monkeypatch.setenv("RTC_KERNEL", "4")
    lib = emu.lib
    lib.rtw_world_scene.restype = C.c_void_p
    lib.rtw_world_scene.argtypes = [C.c_void_p, C.c_int]
    lib.rtc_render_rows_device.restype = C.c_int
    lib.rtc_render_rows_device.argtypes = [C.c_void_p, C.POINTER(RtcCameraC), C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(RtcStatsC),
                                           C.c_int, C.c_int]
    for (cam, world), fuel in ((scenes.synthetic_analytic(hsize=96, vsize=54), 5), (cases.nested_glass(), 8)):
        nw = emu.build_world(world)
        scene = lib.rtw_world_scene(nw.handle, 0)
        assert scene
        cc, rc = emu.camera_c(cam), RtcCameraC()
        assert lib.rtw_make_camera(C.byref(cc), C.byref(rc)) == 0
        out = np.empty(cam.hsize * cam.vsize * 3)
        st = RtcStatsC()
        assert lib.rtc_render_rows_device(scene, C.byref(rc), fuel, 0, 1, cam.vsize, out.ctypes.data, C.byref(st), 1, 1) == 0
        assert st.as_dict()["n_launches"] == 2 * fuel + 4, st.as_dict()


def test_wavefront_and_one_kernel_paths_agree_bitwise(emu, monkeypatch):
    """The wavefront path adds a pixel's contributions in the one-kernel path's order: same bits, whatever the path."""
    for name in ("nested_glass", "synthetic_cones_grouped", "patterns_and_noise", "csg_scene"):
        cam, world = cases.SMALL_CASES[name]()
        out = {}
        for version in ("1", "4"):
            monkeypatch.setenv("RTC_KERNEL", version)
            nw = emu.build_world(world)
            rgb, hits = emu.render(nw, cam, 5)
            out[version] = (rgb.copy(), hits.copy())
        assert np.array_equal(out["1"][0], out["4"][0]), name
        assert np.array_equal(out["1"][1], out["4"][1]), name


@pytest.mark.parametrize("version", ["1", "4"])
def test_simt_emulation(orc, version, monkeypatch):
    """One thread per lane: lanes share the block's stack array (1); wave-aggregated queue pushes with real ballots /
    shuffles (4)."""
    import subprocess
    from emu_lib import EMU_DIR
    from raytracer_challenge_amd.backend import Backend
    subprocess.run(["make", "-s", "-C", EMU_DIR, "simt"], check=True)
    monkeypatch.setenv("RTC_KERNEL", version)
    simt = Backend(os.path.join(EMU_DIR, "_build", "librtc_emu_simt.so"))
    for name in ("teapot_low", "nested_glass"):
        cam, world = cases.SMALL_CASES[name]()
        assert_parity(simt, orc, world, cam, 5, label="simt " + name)


@pytest.mark.parametrize("fuel", [0, 1, 8])
def test_emulated_kernel_fuel(emu, orc, fuel):
    cam, world = cases.nested_glass()
    assert_parity(emu, orc, world, cam, fuel, label="nested_glass fuel=%d" % fuel)


def test_emulated_kernel_edge_rays(emu, orc):
    for name in ("all_primitives", "nested_glass", "nested_groups", "cube_lattice", "synthetic_cones_grouped", "csg_scene"):
        _, world = cases.SMALL_CASES[name]()
        assert_ray_parity(emu, orc, world, cases.edge_rays(2048), 5, label=name)


def test_emulated_kernel_pixel_subset_and_empty(emu, orc):
    cam, world = scenes.chapter11_title(64, 36)
    idx = np.array([0, 5, 63, 64, 1000, 64 * 36 - 1], dtype=np.uint64)
    assert_parity(emu, orc, world, cam, 5, idx, label="subset")
    rgb, hits = emu.render(emu.build_world(world), cam, 5, np.zeros(0, dtype=np.uint64))
    assert rgb.shape == (0, 3) and hits.shape == (0,)
    empty_world = World([PointLight(Color.white(), Vector.point(0, 5, 0))], [])
    rgb, hits = emu.render(emu.build_world(empty_world), cam, 5)
    assert not rgb.any() and (hits["prim"] == -1).all()


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s_[a-z_0-9]+)\s*\(" % prefix, text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(ROOT, "raytracer_challenge_amd", "csrc", "librtc_amd.so"))
    names = declared("rtc.h", "rtc") + declared("rtw.h", "rtw")
    assert len(names) >= 28, names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.rtw_backend.restype = C.c_char_p
    assert lib.rtw_backend() == b"hip"


def test_host_errors_and_flatten(emu):
    b = rt.Backend(os.path.join(ROOT, "raytracer_challenge_amd", "csrc", "librtc_amd.so"))
    singular = Matrix([[1, 0, 0, 0], [0, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    with pytest.raises(rt.RtwError, match="singular"):
        b.build_world(World([], [Element.sphere(ShapeArgs(transform=singular))]))       # reference asserts det != 0
    with pytest.raises(rt.RtwError, match="two children"):
        b.build_world(World([], [Element.composite(Matrix.id(), None, GroupKind.Union, [Element.sphere()])]))  # src/shape.rs:82
    with pytest.raises(rt.RtwError, match="cannot open"):
        b.build_world(World([], [Element.obj("/nonexistent.obj", Matrix.id(), Material())]))
    # flatten sizes of the teapot scene: 3 planes + 240 triangles, one shared xform for the whole OBJ group
    cam, world = scenes.chapter15_teapot("teapot_low.obj", 16, 9)
    nw = b.build_world(world)
    assert nw.primitive_count == 243
    counts = (C.c_uint32 * 8)()
    b.lib.rtw_world_flatten_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    assert b.lib.rtw_world_flatten_counts(nw.handle, counts) == 0
    n_nodes, n_prims, n_xforms, n_limits, n_tris, n_materials, n_pats, n_lights = list(counts)
    assert (n_nodes, n_prims, n_xforms, n_tris, n_lights) == (244, 243, 4, 240, 2) and n_materials == 2


def test_csg_subtrees_beyond_the_per_lane_buffer(emu, orc):
    """The reference's CSG lists are unbounded (src/shape.rs:248-269).  A subtree that can produce more intersections than the
    per-lane buffer holds (32) is rendered through a slab in device memory, csg_max_hits rows per thread of the launch: 20 overlapping
    spheres minus a sphere (42 possible intersections), on both device paths, against the oracle."""
    many = Element.composite(Matrix.id(), None, GroupKind.Aggregation, [Element.sphere(ShapeArgs(transform=Matrix.translation(0.1 * i - 1.0, 0, 0))) for i in range(20)])
    cut = Element.sphere(ShapeArgs(transform=Matrix.translation(0.3, 0.4, -0.6), material=Material(pattern=Pattern.plain(Color.new(0.9, 0.3, 0.2)), transparency=0.5, refractive_index=1.3)))
    world = World([PointLight(Color.white(), Vector.point(0, 5, -5))], [Element.composite(Matrix.id(), None, GroupKind.Difference, [many, cut]), Element.plane(ShapeArgs(transform=Matrix.translation(0, -1.5, 0)))])
    cam = Camera.new(48, 27, 0.9, Camera.transform(Vector.point(0, 1.0, -5), Vector.point(0, 0, 0), Vector.vector(0, 1, 0)))
    assert_parity(emu, orc, world, cam, 5, label="CSG, 42 possible intersections")
    assert_ray_parity(emu, orc, world, cases.edge_rays(512), 5, label="CSG beyond the buffer, edge rays")


def test_no_device_fails_loudly():
    """Without a GPU the product must refuse to compute (no CPU fallback).  Skipped when a device is present."""
    b = rt.Backend(os.path.join(ROOT, "raytracer_challenge_amd", "csrc", "librtc_amd.so"))
    b.lib.rtc_device_count.restype = C.c_int
    if b.lib.rtc_device_count() > 0:
        pytest.skip("a HIP device is present")
    cam, world = scenes.default_world()
    with pytest.raises(rt.RtwError, match="no HIP device"):
        b.render(b.build_world(world), cam, 5)


def test_cabi_argument_validation_without_a_device():
    """Entry points reject bad arguments with RTC_ERR_INVALID (1) and a message before they touch a device: NULL scenes,
    marker slots outside 0..7, NULL descriptors."""
    lib = C.CDLL(os.path.join(ROOT, "raytracer_challenge_amd", "csrc", "librtc_amd.so"))
    lib.rtc_last_error.restype = C.c_char_p
    for fn, args in (("rtc_scene_record", (None, 0)), ("rtc_scene_wait", (None, 3)), ("rtc_scene_check", (None,)), ("rtc_scene_sync", (None,)),
                     ("rtc_scene_create", (None, 0, None)), ("rtc_render", (None, None, 5, None, 0, 0, None, None, None))):
        f = getattr(lib, fn)
        f.restype = C.c_int
        assert f(*args) == 1, fn
        assert lib.rtc_last_error(), fn
    ms = C.c_double(0)
    lib.rtc_scene_elapsed_ms.restype = C.c_int
    assert lib.rtc_scene_elapsed_ms(None, 0, 9, C.byref(ms)) == 1


def test_ppm_writer_matches_oracle_byte_for_byte(orc):
    """Image::ppm (src/image.rs:93-112) from the product's host formatter vs the oracle's, incl. the reference's two layout tests."""
    from raytracer_challenge_amd.image import ppm_text
    rng = np.random.default_rng(3)
    for (h, v) in ((5, 3), (9, 2), (10, 2), (1, 1), (7, 5), (64, 3)):
        rgb = rng.uniform(-0.3, 1.3, (h * v, 3))
        q = orc.quantize(rgb)
        assert ppm_text(h, v, q) == orc.ppm(h, v, rgb)
    rgb = np.tile(np.array([[1.0, 0.8, 0.6]]), (18, 1))     # src/image.rs:170-195: 9x2 image splits lines at 5 pixels
    row5, row4 = "255 204 153 " * 4 + "255 204 153\n", "255 204 153 " * 3 + "255 204 153\n"
    assert ppm_text(9, 2, orc.quantize(rgb)) == "P3\n9 2\n255\n" + row5 + row4 + row5 + row4


def test_emulated_kernel_far_rays(emu, orc):
    for name in ("all_primitives", "cube_lattice", "synthetic_mesh_small", "synthetic_cones_grouped"):
        _, world = cases.SMALL_CASES[name]()
        assert_ray_parity(emu, orc, world, cases.far_rays(512), 5, label="far " + name)



def test_pattern_trees_of_any_depth(emu, orc):
    """The reference's pattern tree is an unbounded Box tree (src/material.rs:60-65).  The device's walk keeps a frame only at nodes
    that need both children's colours or post-process a child's colour, so the limit is on THOSE per path (RTC_MAX_PATTERN_DEPTH = 8),
    not on the tree's depth: 24 nested checkers render like the oracle's recursion, 8 nested blends too, 9 are refused loudly."""
    cam, world = cases.pattern_world(cases.nested_pattern("checkers", 24))
    assert_parity(emu, orc, world, cam, 3, label="24 nested checkers")
    cam, world = cases.pattern_world(cases.nested_pattern("blend", 8))
    assert_parity(emu, orc, world, cam, 3, label="8 nested blends")
    cam, world = cases.pattern_world(cases.nested_pattern("blend", 9))
    with pytest.raises(rt.backend.RtwError, match="colour frames"):
        emu.render(emu.build_world(world), cam, 3)


def test_csg_nested_twelve_deep(emu, orc):
    """Union / Difference groups nested 12 deep (round 2 refused more than 8; the reference's recursion has no limit,
    src/shape.rs:248-269): one sub-program, post-order filters, against the oracle."""
    cam, world = cases.csg_nested(12)
    assert_parity(emu, orc, world, cam, 3, label="CSG nested 12 deep")


def test_nan_t_is_an_error_only_where_the_reference_sort_compares_it(emu, orc):
    """Found by the edge-ray fuzz (seed 5125): a hit at a cone's apex has a NaN normal, so NaN shadow rays; the reference panics
    on a NaN t only when its sort compares it (a list of two or more, src/intersection.rs:123-125).  One plane: the apex pixel is
    its ambient term on both sides.  Two planes: the oracle reports the panic and the device RTC_ERR_NAN."""
    world, rays = cases.cone_apex_world(1)
    assert_ray_parity(emu, orc, world, rays, 3, label="cone apex, one plane")
    rgb, hits = emu.color_at(emu.build_world(world), rays[:1], 3)
    assert hits["prim"][0] == 0 and hits["t"][0] == 5.0 and np.allclose(rgb[0], 0.25 * np.array([0.2, 0.6, 0.3]) * 1.5)
    world, rays = cases.cone_apex_world(2)
    with pytest.raises(rt.backend.RtwError, match="NaN"):
        orc.color_at(orc.build_world(world), rays, 3)
    with pytest.raises(rt.backend.RtwError, match="NaN"):
        emu.color_at(emu.build_world(world), rays, 3)
    assert_ray_parity(emu, orc, world, rays[1:], 3, label="cone apex world, ordinary rays")
    # the two planes under an aggregation group: the group's box test fails for the NaN shadow rays, the planes are never asked, no
    # panic (found by the lattice scenes: planes whose record travels in the kernel arguments skipped their group's gate)
    world, rays = cases.cone_apex_world(2, planes_in_group=True)
    assert_ray_parity(emu, orc, world, rays, 3, label="cone apex, two planes in a group")


def test_nan_reflectance_makes_the_pixel_nan_as_in_the_reference(emu, orc):
    """Found by rays aimed at the primitives' special points (cases.special_rays): on a reflective AND transparent cone the apex hit has a
    NaN Schlick reflectance, and the reference's `reflected * reflectance + refracted * (1 - reflectance)` (src/world.rs:70-78) is NaN
    even at fuel 0 where both colours are black.  The device used to return the surface colour; it now returns the reference's NaN."""
    world, rays = cases.cone_apex_world(1, glass_mirror=True)
    for fuel in (0, 1, 3):
        rgb, hits = emu.color_at(emu.build_world(world), rays, fuel)
        assert np.isnan(rgb[0]).all() and np.isfinite(rgb[1:]).all()
        assert_ray_parity(emu, orc, world, rays, fuel, label="glass-mirror cone apex, fuel %d" % fuel)
    # the same pixel through a camera, both device paths of the emulator (digest included)
    from raytracer_challenge_amd.scene import Camera
    cam = Camera.new(9, 9, 0.5, Camera.transform(Vector.point(0, 0, -5), Vector.point(0, 0, 0), Vector.vector(0, 1, 0)))
    assert_parity(emu, orc, world, cam, 3, label="glass-mirror cone apex through a camera")


def test_refractive_index_must_be_a_positive_finite_number(emu):
    """The reference computes with any refractive index; the device's fuel-0 rule for a NaN reflectance (blend_reflectance) needs
    Schlick's r0 finite, so rtc_scene_create refuses indices that are not positive finite numbers (RTC_ERR_UNSUPPORTED)."""
    world, rays = cases.cone_apex_world(1)
    for bad in (0.0, -1.5, float("inf"), float("nan")):
        els = list(world.elements)
        els[1] = Element.plane(ShapeArgs(transform=els[1].args.transform, material=Material(refractive_index=bad)))
        with pytest.raises(rt.backend.RtwError, match="refractive_index"):
            emu.color_at(emu.build_world(World(lights=world.lights, elements=els)), rays, 1)   # (the scene is created at the first render)


@pytest.mark.parametrize("name", ["all_primitives", "nested_glass", "nested_groups", "cube_lattice", "synthetic_cones_grouped", "csg_scene"])
def test_emulated_kernel_special_point_rays(emu, orc, name):
    """World::color_at on rays aimed at the points where the primitives' formulas change branch (cases.special_rays: a cone's apex, cap
    rims, cube corners and edges, poles, triangle vertices and edges; from / to lights; along object-space axes).  Rays on which the
    reference panics (a NaN t in a sorted list of two or more) must be refused one by one; all others match hit for hit."""
    _, world = cases.SMALL_CASES[name]()
    assert_ray_parity_with_panics(emu, orc, world, cases.special_rays(world, 3072), 5, label="special rays " + name)


def test_sixty_four_lights_and_one_more(emu, orc, monkeypatch):
    """64 lights is the device's limit (one shadow bit per light in the wavefront path's shade records); the reference re-traces every
    secondary subtree once per light, so fuel 1 keeps the oracle at 64^2 traces per pixel.  65 lights are refused loudly."""
    cam, world = cases.SMALL_CASES["all_primitives"]()
    rng = np.random.default_rng(5)
    lights = [PointLight(Color.new(*rng.uniform(0.01, 0.05, 3)), Vector.point(*rng.uniform(-10, 10, 3))) for _ in range(64)]
    small = Camera.new(24, 14, cam.field_of_view, cam.transform_matrix)
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        assert_parity(emu, orc, World(lights, world.elements), small, 1, label="64 lights, path " + path)
    with pytest.raises(rt.backend.RtwError, match="64 lights"):
        emu.render(emu.build_world(World(lights + lights[:1], world.elements)), small, 1)
    with pytest.raises(rt.backend.RtwError, match="RTC_MAX_FUEL"):
        emu.render(emu.build_world(world), small, 17)


def test_random_pixel_lists_with_repeats(emu, monkeypatch):
    """rtc_render / rtc_render_hit_digest on unordered pixel lists with repeated indices (list lengths around the wave size): every
    entry equals that pixel of the full frame, on both emulated paths."""
    cam, world = scenes.chapter11_title(64, 36)
    nw = emu.build_world(world)
    full, full_hits = emu.render(nw, cam, 5)
    full_dig = emu.render_digest(nw, cam, 5)
    rng = np.random.default_rng(3)
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        for n in (1, 7, 64, 65, 500, 3000):
            idx = rng.integers(0, 64 * 36, n).astype(np.uint64)
            rgb, hits = emu.render(nw, cam, 5, idx)
            assert np.array_equal(rgb, full[idx]) and np.array_equal(hits, full_hits[idx]), (path, n)
            assert np.array_equal(emu.render_digest(nw, cam, 5, idx), full_dig[idx]), (path, n)
