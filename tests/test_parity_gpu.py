"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the CPU oracle
on the same scene descriptions; against the reference's committed renders (tests/golden); and, at BASELINE sizes,
through size-independent properties."""
import json
import os

import numpy as np
import pytest

from raytracer_challenge_amd import scenes
import cases
from parity import RGB_TOL, assert_parity, assert_ray_parity, assert_ray_parity_with_panics
from test_oracle_pins import SCENES as GOLDEN_SCENES, load_samples

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(cases.SMALL_CASES))
def test_hip_matches_oracle(hip, orc, name):
    cam, world = cases.SMALL_CASES[name]()
    assert_parity(hip, orc, world, cam, 5, label=name)


@pytest.mark.parametrize("version", ["1", "4"])
def test_hip_every_kernel_version(hip, orc, version, monkeypatch):
    """RTC_KERNEL pins the device path at scene creation: 1 = one kernel per frame, 4 = wavefront kernels.  Both must be
    bit-exact in hits on analytic, mesh, grouped, glass and CSG scenes."""
    monkeypatch.setenv("RTC_KERNEL", version)
    for name in ("synthetic_cones_grouped", "teapot_low", "nested_glass", "cube_lattice", "synthetic_mesh_small", "patterns_and_noise", "csg_scene"):
        cam, world = cases.SMALL_CASES[name]()
        assert_parity(hip, orc, world, cam, 5, label="kernel v%s %s" % (version, name))


@pytest.mark.parametrize("fuel", [0, 1, 8])
def test_hip_fuel(hip, orc, fuel):
    cam, world = cases.nested_glass()
    assert_parity(hip, orc, world, cam, fuel, label="nested_glass fuel=%d" % fuel)
    cam, world = scenes.synthetic_analytic(hsize=64, vsize=36)
    assert_parity(hip, orc, world, cam, fuel, label="synthetic fuel=%d" % fuel)


def test_hip_edge_rays(hip, orc):
    for name in ("all_primitives", "nested_glass", "nested_groups", "synthetic_cones_grouped", "cube_lattice", "csg_scene"):
        _, world = cases.SMALL_CASES[name]()
        assert_ray_parity(hip, orc, world, cases.edge_rays(4096), 5, label=name)


@pytest.mark.parametrize("path", ["1", "4"])
@pytest.mark.parametrize("name", sorted(GOLDEN_SCENES))
def test_hip_matches_reference_render(hip, orc, name, path, monkeypatch):
    """Sampled pixels of the reference's own 4096-wide renders, exact after the reference's 8-bit quantiser — on BOTH device paths
    (RTC_KERNEL pins the path for pixel-list launches too: 1 = one kernel per launch, 4 = the wavefront kernels)."""
    monkeypatch.setenv("RTC_KERNEL", path)
    doc, s = load_samples(name)
    cam, world = GOLDEN_SCENES[name]()
    idx = (s[:, 1] * cam.hsize + s[:, 0]).astype(np.uint64)
    rgb, _ = hip.render(hip.build_world(world), cam, 5, idx)
    q = orc.quantize(rgb).astype(np.int64)   # the quantiser (src/color.rs:42-46) is the checker's, applied to HIP output
    bad = np.flatnonzero(np.abs(q - s[:, 2:5]).max(1) > 0)
    assert bad.size == 0, "%s (path %s): %d/%d sampled pixels differ from the reference PNG" % (name, path, bad.size, len(s))


def hits_equal(a, b):
    return bool(np.array_equal(a["prim"], b["prim"]) and np.array_equal(a["push_idx"], b["push_idx"]) and np.array_equal(a["t"].view(np.uint64), b["t"].view(np.uint64)))


@pytest.mark.parametrize("label", ["config2", "config3", "config2_cones"])
def test_hip_full_size_properties(hip, orc, label, monkeypatch):
    """BASELINE config 2 / 3 (and SURVEY C2's grouped variant with cones: kernel variant 5) at 1920x1080, fuel 5, on BOTH device paths: (i) >= 100 000 pixels of the full frame against the oracle
    (primary hits and the hit-tree digest of every ray tree bit-exact, colours <= 1e-5); (ii) rendering by explicit index list == the same pixels of the full-range render
    (idempotence / order independence); (iii) the two paths give the same bits; (iv) every primary hit record is self-consistent."""
    cam, world = (scenes.synthetic_analytic() if label == "config2" else scenes.synthetic_analytic(cones=True, grouped=True) if label == "config2_cones"
                  else scenes.chapter15_teapot("teapot_low.obj", 1920, 1080))
    idx = np.arange(0, 1920 * 1080, 19, dtype=np.uint64)          # 109 137 pixels
    assert idx.size >= 100000
    ref_rgb, ref_hits, ref_dig = orc.render_with_digest(orc.build_world(world), cam, 5, idx)
    frames, digests = {}, {}
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        nw = hip.build_world(world)
        rgb, hits = hip.render(nw, cam, 5)
        assert rgb.shape == (1920 * 1080, 3) and np.isfinite(rgb).all()
        assert (hits["t"][hits["prim"] >= 0] >= 0).all() and (hits["prim"] < nw.primitive_count).all()
        sub = np.arange(0, 1920 * 1080, 997, dtype=np.uint64)
        rgb2, hits2 = hip.render(nw, cam, 5, sub)
        assert np.array_equal(rgb2, rgb[sub.astype(np.int64)]) and np.array_equal(hits2, hits[sub.astype(np.int64)])
        ii = idx.astype(np.int64)
        assert hits_equal(hits[ii], ref_hits), "%s path %s: primary hits differ from the oracle" % (label, path)
        err = float(np.abs(rgb[ii] - ref_rgb).max())
        assert err <= RGB_TOL, "%s path %s: max |dRGB| = %.3e" % (label, path, err)
        # every closest hit of every ray tree: the whole frame's hit-tree digests (whole-row launch, the timed kernels' counting
        # variants), the oracle's on the sample
        dig = hip.render_digest(nw, cam, 5)
        assert dig.shape == (1920 * 1080,)
        bad = dig[ii] != ref_dig
        assert not bad.any(), "%s path %s: hit-tree digests of %d / %d sampled pixels differ from the oracle" % (label, path, int(bad.sum()), bad.size)
        frames[path] = (rgb, hits)
        digests[path] = dig
    assert np.array_equal(frames["1"][0], frames["4"][0]) and np.array_equal(frames["1"][1], frames["4"][1])
    assert np.array_equal(digests["1"], digests["4"])      # all 2 073 600 ray trees, path against path


def test_hip_rows_device_matches_host_path(hip):
    """The tile-interleaved device entry point (multi-GPU partition) must produce the same pixels as rtc_render."""
    import torch
    from raytracer_challenge_amd.device import DeviceRenderer
    cam, world = scenes.chapter11_title(128, 72)
    nw = hip.build_world(world)
    full, _ = hip.render(nw, cam, 5)
    dr = DeviceRenderer(hip, nw, cam, device=0)
    for step, first in ((1, 0), (2, 1), (4, 3), (8, 5)):
        rows = list(range(first, cam.vsize, step))
        out = torch.empty(len(rows) * cam.hsize * 3, dtype=torch.float64, device="cuda:0")
        st = dr.render_rows(5, first, step, len(rows), out, count=True)
        assert st["pixels"] == len(rows) * cam.hsize and st["rays_primary"] == st["pixels"]
        got = out.cpu().numpy().reshape(-1, 3)
        want = full.reshape(cam.vsize, cam.hsize, 3)[rows].reshape(-1, 3)
        assert np.array_equal(got, want), (step, first)


def test_hip_both_paths_give_the_same_bits(hip, monkeypatch):
    """One-kernel path (RTC_KERNEL=1) and wavefront path (RTC_KERNEL=4): identical pixels and hit records, bit for bit
    (the wavefront path adds a pixel's contributions in the one-kernel path's order)."""
    for name in ("nested_glass", "synthetic_cones_grouped", "patterns_and_noise", "csg_scene", "teapot_low"):
        cam, world = cases.SMALL_CASES[name]()
        out = {}
        for version in ("1", "4"):
            monkeypatch.setenv("RTC_KERNEL", version)
            nw = hip.build_world(world)
            out[version] = hip.render(nw, cam, 5)
        assert np.array_equal(out["1"][0], out["4"][0]), name
        assert np.array_equal(out["1"][1], out["4"][1]), name


def test_hip_wavefront_queue_growth_and_fallback(hip, orc, monkeypatch):
    """A glass scene at fuel 8 outgrows the initial queues (2 x work ids per level): with the default memory budget the
    queues grow and the wavefront path finishes; with a budget too small for any growth the launch falls back to the
    one-kernel path.  Both must match the oracle."""
    cam, world = cases.nested_glass()
    monkeypatch.setenv("RTC_KERNEL", "4")
    assert_parity(hip, orc, world, cam, 8, label="nested_glass fuel 8, queues grown")
    monkeypatch.setenv("RTC_WF_MAX_BYTES", "4000000")
    assert_parity(hip, orc, world, cam, 8, label="nested_glass fuel 8, fallback")


def test_hip_measured_path_choice(hip):
    """DeviceRenderer.tune(): four synchronous launches measure both paths twice, later launches (also unsynchronised ones) take
    the faster; whatever is chosen, the pixels are those of a plain synchronous render."""
    import torch
    from raytracer_challenge_amd.device import DeviceRenderer
    cam, world = scenes.synthetic_analytic(hsize=256, vsize=144)
    nw = hip.build_world(world)
    full, _ = hip.render(nw, cam, 5)
    dr = DeviceRenderer(hip, nw, cam, device=0)
    out = torch.empty(cam.vsize * cam.hsize * 3, dtype=torch.float64, device="cuda:0")
    assert dr.path_info()["path"] == "undecided"
    info = dr.tune(5, 0, 1, cam.vsize, out)
    assert info["path"] in ("one kernel", "wavefront") and info["one_kernel_ms"] > 0 and info["wavefront_ms"] > 0
    out.zero_()
    dr.render_rows_async(5, 0, 1, cam.vsize, out)
    dr.sync()
    dr.check()
    assert np.array_equal(out.cpu().numpy().reshape(-1, 3), full)


def test_hip_error_state_is_sticky_across_async_launches(hip):
    """A bad asynchronous launch (NaN camera -> NaN intersection t's) followed by a good one: rtc_scene_check() still reports
    RTC_ERR_NAN (the flags accumulate until read), and a second check is clean."""
    import torch
    from raytracer_challenge_amd import RtwError
    from raytracer_challenge_amd.device import DeviceRenderer
    cam, world = scenes.chapter15_teapot("teapot_low.obj", 32, 18)   # planes are tested outside any accelerator: NaN rays reach them
    nw = hip.build_world(world)
    for version in ("1", "4"):
        os.environ["RTC_KERNEL"] = version
        try:
            dr = DeviceRenderer(hip, nw, cam, device=0)
        finally:
            os.environ.pop("RTC_KERNEL")
        out = torch.empty(cam.vsize * cam.hsize * 3, dtype=torch.float64, device="cuda:0")
        good = dr.cam.transform_inv[0]
        dr.cam.transform_inv[0] = float("nan")
        dr.render_rows_async(5, 0, 1, cam.vsize, out)
        dr.cam.transform_inv[0] = good
        dr.render_rows_async(5, 0, 1, cam.vsize, out)
        with pytest.raises(RtwError, match="NaN"):
            dr.check()
        dr.check()
        # markers: waiting on / measuring a slot that was never recorded is an error, not an undefined wait
        dr.record(1)
        dr.wait(1)
        with pytest.raises(RtwError, match="never recorded"):
            dr.wait(5)
        with pytest.raises(RtwError, match="never recorded"):
            dr.elapsed_ms(1, 6)
        with pytest.raises(RtwError, match="out of range"):
            dr.record(8)


def test_hip_wavefront_allocation_failure_falls_back(hip, orc, monkeypatch):
    """The device refuses the wavefront queues (forced: RTC_WF_FAIL_ALLOC): the launch is rendered by the one-kernel path,
    same bits, no error."""
    cam, world = cases.nested_glass()
    monkeypatch.setenv("RTC_KERNEL", "4")
    ref = hip.render(hip.build_world(world), cam, 5)
    monkeypatch.setenv("RTC_WF_FAIL_ALLOC", "1")
    got = hip.render(hip.build_world(world), cam, 5)
    assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
    monkeypatch.delenv("RTC_KERNEL")
    got = hip.render(hip.build_world(world), cam, 5)     # measured choice: the wavefront candidate is dropped, not an error
    assert np.array_equal(ref[0], got[0])


@pytest.mark.parametrize("path", ["1", "4"])
def test_hip_csg_subtrees_beyond_the_per_lane_buffer(hip, orc, path, monkeypatch):
    """CSG subtrees that can produce more than 32 intersections run through a slab in device memory (include/rtc.h): 20 overlapping
    spheres minus a sphere, both device paths, against the oracle."""
    from raytracer_challenge_amd.scene import Camera, Color, Element, GroupKind, Material, Matrix, Pattern, PointLight, ShapeArgs, Vector, World
    monkeypatch.setenv("RTC_KERNEL", path)
    many = Element.composite(Matrix.id(), None, GroupKind.Aggregation, [Element.sphere(ShapeArgs(transform=Matrix.translation(0.1 * i - 1.0, 0, 0))) for i in range(20)])
    cut = Element.sphere(ShapeArgs(transform=Matrix.translation(0.3, 0.4, -0.6), material=Material(pattern=Pattern.plain(Color.new(0.9, 0.3, 0.2)), transparency=0.5, refractive_index=1.3)))
    world = World([PointLight(Color.white(), Vector.point(0, 5, -5))], [Element.composite(Matrix.id(), None, GroupKind.Difference, [many, cut]), Element.plane(ShapeArgs(transform=Matrix.translation(0, -1.5, 0)))])
    cam = Camera.new(160, 90, 0.9, Camera.transform(Vector.point(0, 1.0, -5), Vector.point(0, 0, 0), Vector.vector(0, 1, 0)))
    assert_parity(hip, orc, world, cam, 5, label="CSG, 42 possible intersections, path " + path)
    assert_ray_parity(hip, orc, world, cases.edge_rays(2048), 5, label="CSG beyond the buffer, edge rays")


@pytest.mark.parametrize("path", ["1", "4"])
def test_hip_structures_beyond_round_two_limits(hip, orc, path, monkeypatch):
    """Reference-unbounded structures (src/material.rs:60-65 pattern Box trees, src/shape.rs:248-269 nested CSG): 24 nested checkers,
    8 nested blends and CSG groups nested 12 deep against the oracle on both device paths; 9 nested blends are refused loudly."""
    from raytracer_challenge_amd.backend import RtwError
    monkeypatch.setenv("RTC_KERNEL", path)
    for label, (cam, world) in (("24 nested checkers", cases.pattern_world(cases.nested_pattern("checkers", 24))),
                                ("8 nested blends", cases.pattern_world(cases.nested_pattern("blend", 8))),
                                ("CSG nested 12 deep", cases.csg_nested(12))):
        assert_parity(hip, orc, world, cam, 3, label=label + ", path " + path)
    cam, world = cases.pattern_world(cases.nested_pattern("blend", 9))
    with pytest.raises(RtwError, match="colour frames"):
        hip.render(hip.build_world(world), cam, 3)


def test_hip_config4_teapot_high_4k_fuel8(hip, orc):
    """BASELINE configs[3] on one GPU: teapot_high.obj (6 320 smooth triangles), 3840x2160, fuel 8 — full frame on the HIP path,
    a strided sample against the oracle (which tests every triangle of the flat group, src/shape.rs:254-256)."""
    cam, world = scenes.chapter15_teapot("teapot_high.obj", 3840, 2160)
    nw = hip.build_world(world)
    rgb, hits = hip.render(nw, cam, 8)
    assert np.isfinite(rgb).all() and (hits["prim"] < nw.primitive_count).all()
    on_teapot = np.flatnonzero(hits["prim"] >= 3)
    assert on_teapot.size > 100000
    idx = np.concatenate([np.arange(0, 3840 * 2160, 2203), on_teapot[::150]]).astype(np.uint64)   # ~3 800 + ~3 000 pixels
    assert idx.size >= 5000
    assert_parity(hip, orc, world, cam, 8, idx, label="config4 sample (%d px)" % idx.size)


def test_hip_config5_million_triangles_noise(hip, orc, tmp_path):
    """BASELINE configs[4] on one GPU: ~10^6-triangle smooth mesh (one OBJ group) with Fractal/Simplex procedural textures,
    3840x2160, fuel 8.  Full frame on the HIP path; oracle parity on a small pixel sample AT FUEL 8 (the oracle tests
    10^6 triangles per ray that enters the group's box: seconds per pixel); full-size property: index-list render == full-range render."""
    path = str(tmp_path / "heightfield_708.obj")
    ntri = scenes.write_heightfield_obj(path, 708, 708, 12345)
    assert ntri == 999698
    cam, world = scenes.synthetic_mesh(path)
    nw = hip.build_world(world)
    assert nw.primitive_count == ntri + 3
    rgb, hits = hip.render(nw, cam, 8)
    assert np.isfinite(rgb).all()
    on_mesh = np.flatnonzero((hits["prim"] >= 1) & (hits["prim"] <= ntri))
    assert on_mesh.size > 1000000
    idx = np.concatenate([np.arange(1000, 3840 * 2160, 3840 * 2160 // 24), on_mesh[:: on_mesh.size // 24]]).astype(np.uint64)
    rgb2, hits2 = hip.render(nw, cam, 8, idx)
    assert np.array_equal(rgb2, rgb[idx.astype(np.int64)]) and np.array_equal(hits2, hits[idx.astype(np.int64)])
    # (the oracle's flat triangle list costs ~10 s per on-mesh pixel and thread here: 48 pixels; the 10^5-triangle cut below takes 640)
    assert_parity(hip, orc, world, cam, 8, idx, label="config5 sample at fuel 8 (%d px)" % idx.size)


def test_hip_config5_cut_100k_triangles_4k_fuel8(hip, orc, tmp_path):
    """The same scene with a 10^5-triangle cut of the mesh (224 x 224 heightfield: 99 458 smooth triangles, the same noise
    patterns) at the full 3840x2160, fuel 8: 640 oracle pixels — half of them on the mesh — with primary hits, the hit-tree digest
    of every pixel and colours, on both device paths."""
    from parity import oracle_reference
    path = str(tmp_path / "heightfield_224.obj")
    ntri = scenes.write_heightfield_obj(path, 224, 224, 12345)
    assert ntri == 99458
    cam, world = scenes.synthetic_mesh(path, nx=224, nz=224)
    nw = hip.build_world(world)
    rgb, hits = hip.render(nw, cam, 8)
    on_mesh = np.flatnonzero((hits["prim"] >= 1) & (hits["prim"] <= ntri))
    assert on_mesh.size > 1000000
    idx = np.concatenate([np.arange(777, 3840 * 2160, 3840 * 2160 // 320), on_mesh[:: on_mesh.size // 320]]).astype(np.uint64)
    assert idx.size >= 640
    ref = oracle_reference(orc, world, cam, 8, idx, threads=16)
    import os
    for kernel in ("1", "4"):
        os.environ["RTC_KERNEL"] = kernel
        try:
            assert_parity(hip, orc, world, cam, 8, idx, label="config5 cut (99 458 triangles), 4K fuel 8, %d px, path %s" % (idx.size, kernel), ref=ref)
        finally:
            del os.environ["RTC_KERNEL"]


def test_hip_quantiser_and_ppm(hip, orc):
    """Color::clamp on the device and Image::ppm, byte-exact against the oracle; edge values incl. NaN,
    +-inf, exact .5 boundaries (round half away from zero)."""
    from raytracer_challenge_amd.image import Image
    cam, world = scenes.chapter11_title(80, 45)
    img = Image.par_render(cam, world)
    ref_rgb, _ = orc.render(orc.build_world(world), cam, 5)
    assert np.abs(img.pixels - ref_rgb).max() <= RGB_TOL
    assert np.array_equal(img.quantized(), orc.quantize(img.pixels))
    assert img.ppm() == orc.ppm(80, 45, img.pixels)
    c = img.read(40, 22)
    assert (c.r, c.g, c.b) == tuple(img.pixels[22 * 80 + 40])
    edge = np.array([[np.nan, np.inf, -np.inf], [0.5 / 255, 1.5 / 255, 2.5 / 255], [-0.0, 1.0, 0.999999999], [254.5 / 255, 0.49999 / 255, 1e-300]])
    e = Image(4, 1, edge, img._native)
    assert np.array_equal(e.quantized(), orc.quantize(edge))


@pytest.mark.parametrize("path", ["1", "4"])
def test_hip_nan_t_is_an_error_only_where_the_reference_sort_compares_it(hip, orc, monkeypatch, path):
    """tests/test_kernel_logic_cpu.py's case of the same name on the device, both paths: a hit at a cone's apex (NaN normal, NaN
    shadow rays) over one plane is the ambient term, over two planes the reference's sort panics = RTC_ERR_NAN."""
    from raytracer_challenge_amd import RtwError
    monkeypatch.setenv("RTC_KERNEL", path)
    world, rays = cases.cone_apex_world(1)
    assert_ray_parity(hip, orc, world, rays, 3, label="cone apex, one plane, path " + path)
    world, rays = cases.cone_apex_world(2)
    with pytest.raises(RtwError, match="NaN"):
        hip.color_at(hip.build_world(world), rays, 3)
    assert_ray_parity(hip, orc, world, rays[1:], 3, label="cone apex world, ordinary rays, path " + path)
    world, rays = cases.cone_apex_world(2, planes_in_group=True)   # the group's box test fails for NaN rays: the planes are never asked
    assert_ray_parity(hip, orc, world, rays, 3, label="cone apex, two planes in a group, path " + path)


@pytest.mark.parametrize("path", ["1", "4"])
def test_hip_nan_reflectance_makes_the_pixel_nan_as_in_the_reference(hip, orc, monkeypatch, path):
    """A reflective AND transparent cone hit at its apex: NaN Schlick reflectance, and the reference's blend of two black colours with it
    is NaN at every fuel (src/world.rs:70-78).  Rays and a 9x9 camera whose centre pixel is that ray, both device paths."""
    from raytracer_challenge_amd.scene import Camera, Vector
    monkeypatch.setenv("RTC_KERNEL", path)
    world, rays = cases.cone_apex_world(1, glass_mirror=True)
    for fuel in (0, 1, 3):
        rgb, _ = hip.color_at(hip.build_world(world), rays, fuel)
        assert np.isnan(rgb[0]).all() and np.isfinite(rgb[1:]).all()
        assert_ray_parity(hip, orc, world, rays, fuel, label="glass-mirror cone apex, fuel %d, path %s" % (fuel, path))
    cam = Camera.new(9, 9, 0.5, Camera.transform(Vector.point(0, 0, -5), Vector.point(0, 0, 0), Vector.vector(0, 1, 0)))
    assert_parity(hip, orc, world, cam, 3, label="glass-mirror cone apex through a camera, path " + path)


def test_hip_special_point_rays(hip, orc, monkeypatch):
    """cases.special_rays (aimed at apexes, rims, corners, edges, poles, vertices; from / to lights; along object-space axes) on the small
    scenes, both device paths; rays the reference panics on must be refused one by one (tests/parity.py)."""
    for name in ("all_primitives", "nested_glass", "nested_groups", "cube_lattice", "synthetic_cones_grouped", "csg_scene", "synthetic_mesh_small"):
        _, world = cases.SMALL_CASES[name]()
        rays = cases.special_rays(world, 1024 if name == "synthetic_mesh_small" else 6144)   # (the oracle tests every triangle for every ray)
        for path in ("1", "4"):
            monkeypatch.setenv("RTC_KERNEL", path)
            assert_ray_parity_with_panics(hip, orc, world, rays, 5, label="special rays %s path %s" % (name, path))


def test_hip_far_rays(hip, orc):
    for name in ("all_primitives", "cube_lattice", "synthetic_mesh_small", "synthetic_cones_grouped"):
        _, world = cases.SMALL_CASES[name]()
        assert_ray_parity(hip, orc, world, cases.far_rays(2048), 5, label="far " + name)


def test_hip_random_pixel_lists_with_repeats(hip, monkeypatch):
    """As tests/test_kernel_logic_cpu.py: unordered pixel lists with repeated indices against the full frame, both device paths."""
    cam, world = scenes.chapter11_title(256, 144)
    nw = hip.build_world(world)
    full, full_hits = hip.render(nw, cam, 5)
    full_dig = hip.render_digest(nw, cam, 5)
    rng = np.random.default_rng(3)
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        for n in (1, 63, 64, 65, 4097, 100000):
            idx = rng.integers(0, 256 * 144, n).astype(np.uint64)
            rgb, hits = hip.render(nw, cam, 5, idx)
            assert np.array_equal(rgb, full[idx]) and np.array_equal(hits, full_hits[idx]), (path, n)
            assert np.array_equal(hip.render_digest(nw, cam, 5, idx), full_dig[idx]), (path, n)
