"""Light grids (scene_build.hpp build_light_grids): a shadow ray takes its candidates from its light's direction grid
instead of walking the analytic BVH.  Results-neutral by construction, so a scene rendered with the grids (default) and without
(RTC_LIGHT_GRID=0) must give the same pixels and hit records bit for bit, on both device paths — and the counters must show that
the grids really stood in for walks."""
import numpy as np
import pytest

from parity import assert_parity
from raytracer_challenge_amd import scenes
from raytracer_challenge_amd.device import DeviceRenderer


def both_ways(be, world, cam, fuel, monkeypatch, idx=None, kernels=("1", "4")):
    out = {}
    for k in kernels:
        monkeypatch.setenv("RTC_KERNEL", k)
        for flag in ("0", "1"):
            monkeypatch.setenv("RTC_LIGHT_GRID", flag)
            out[k, flag] = be.render(be.build_world(world), cam, fuel, idx)
    ref = out[kernels[0], "0"]
    for key, (rgb, hits) in out.items():
        assert np.array_equal(hits, ref[1]), key
        assert np.array_equal(rgb, ref[0]), key
    monkeypatch.delenv("RTC_LIGHT_GRID")
    monkeypatch.delenv("RTC_KERNEL")


@pytest.fixture(scope="module")
def emu():
    from emu_lib import emu as _emu
    return _emu()


@pytest.mark.parametrize("cones,grouped", [(False, False), (True, True)])
def test_light_grids_are_results_neutral_in_the_emulator(emu, orc, monkeypatch, cones, grouped):
    cam, world = scenes.synthetic_analytic(n_primitives=96, seed=7, cones=cones, grouped=grouped, hsize=96, vsize=54)
    both_ways(emu, world, cam, 3, monkeypatch)
    assert_parity(emu, orc, world, cam, 3, label="synthetic analytic, light grids on")


def test_light_grid_counters_in_the_emulator(emu, monkeypatch):
    """With grids: one cell lookup per shadow ray (minus the rays of over-full cells) and far fewer accelerator nodes."""
    import torch
    cam, world = scenes.synthetic_analytic(n_primitives=96, seed=7, hsize=96, vsize=54)
    st = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RTC_LIGHT_GRID", flag)
        dr = DeviceRenderer(emu, emu.build_world(world), cam, 0, _cpu_standin=True)
        out = torch.empty(cam.vsize * cam.hsize * 3, dtype=torch.float64)
        st[flag] = dr.render_rows(3, 0, 1, cam.vsize, out, count=True, sync=True)
        st[flag]["img"] = out.clone()
    assert torch.equal(st["0"]["img"], st["1"]["img"])
    assert st["0"]["light_grid_cells"] == 0
    assert 0.9 * st["1"]["rays_shadow"] <= st["1"]["light_grid_cells"] <= st["1"]["rays_shadow"]
    assert st["1"]["accel_nodes"] < 0.6 * st["0"]["accel_nodes"]
    for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract"):
        assert st["0"][k] == st["1"][k], k


def test_light_inside_a_primitive_and_degenerate_shadow_rays(emu, orc, monkeypatch):
    """A light inside primitives' bounds (those are candidates of every cell) and a light ON a surface (zero-length shadow rays: the
    direction is NaN, the cell function declines, the ray walks the BVH)."""
    from raytracer_challenge_amd.scene import Color, PointLight, Vector
    cam, world = scenes.synthetic_analytic(n_primitives=64, seed=3, hsize=64, vsize=36)
    world.lights.append(PointLight(Color.new(0.5, 0.5, 0.5), Vector.point(0.0, 1.0, 0.0)))
    world.lights.append(PointLight(Color.new(0.2, 0.2, 0.2), Vector.point(0.3, 0.0, 0.2)))   # on the floor plane
    both_ways(emu, world, cam, 2, monkeypatch)
    assert_parity(emu, orc, world, cam, 2, label="lights inside bounds / on a surface")


def test_overlapping_glass_in_the_emulator(emu, orc, monkeypatch):
    """cases.glass_cluster: container lists several shapes deep, a light INSIDE the cluster (so inside many primitives' bounds),
    light grids on / off, both paths, against the oracle."""
    import cases
    cam, world = cases.glass_cluster()
    both_ways(emu, world, cam, 4, monkeypatch)
    for k in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", k)
        assert_parity(emu, orc, world, cam, 4, label="glass cluster, path " + k)


@pytest.mark.gpu
def test_hip_overlapping_glass(hip, orc, monkeypatch):
    import cases
    cam, world = cases.glass_cluster()
    both_ways(hip, world, cam, 6, monkeypatch)
    for k in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", k)
        assert_parity(hip, orc, world, cam, 6, label="glass cluster, path " + k)


@pytest.mark.gpu
@pytest.mark.parametrize("cones,grouped", [(False, False), (True, True)])
def test_hip_light_grids_are_results_neutral(hip, orc, monkeypatch, cones, grouped):
    cam, world = scenes.synthetic_analytic(n_primitives=512, seed=12345, cones=cones, grouped=grouped, hsize=480, vsize=270)
    both_ways(hip, world, cam, 5, monkeypatch)
    assert_parity(hip, orc, world, cam, 5, np.arange(0, 480 * 270, 7, dtype=np.uint64), label="config-2 scene at 480x270, light grids on")


def test_light_grid_build_has_a_work_budget(emu, orc):
    """A light in the middle of a cloud of large boxes: every box projects onto every cell of the light's direction grid, the build
    would cost boxes x cells and throw the lists away as over-full.  It stops at an average of 64 entries per cell and builds no
    grids (scene_build.hpp build_light_grids): shadow rays walk the BVH, results unchanged."""
    import torch
    from raytracer_challenge_amd.scene import Camera, Color, Element, Material, Matrix, Pattern, PointLight, ShapeArgs, Vector, World
    rng = np.random.default_rng(5)
    els = [Element.plane(ShapeArgs(transform=Matrix.translation(0, -12, 0)))]
    for i in range(120):   # spheres of radius 6 around the origin: all of them contain the light at the origin
        t = Matrix.translation(*rng.uniform(-1.5, 1.5, 3)) * Matrix.scaling(6, 6, 6)
        els.append(Element.sphere(ShapeArgs(transform=t, material=Material(pattern=Pattern.plain(Color.new(*rng.uniform(0.2, 0.9, 3))), transparency=0.2 if i % 7 == 0 else 0.0))))
    world = World([PointLight(Color.white(), Vector.point(0.0, 0.0, 0.0)), PointLight(Color.new(0.3, 0.3, 0.3), Vector.point(0, 20, -20))], els)
    cam = Camera.new(48, 27, 1.0, Camera.transform(Vector.point(0, 5, -22), Vector.point(0, 0, 0), Vector.vector(0, 1, 0)))
    dr = DeviceRenderer(emu, emu.build_world(world), cam, 0, _cpu_standin=True)
    out = torch.empty(cam.vsize * cam.hsize * 3, dtype=torch.float64)
    st = dr.render_rows(2, 0, 1, cam.vsize, out, count=True, sync=True)
    assert st["rays_shadow"] > 0 and st["light_grid_cells"] == 0
    assert_parity(emu, orc, world, cam, 2, label="light inside a cloud of large boxes (no grids)")


def test_lights_behind_the_surface_are_answered_without_a_shadow_traversal(emu, orc, monkeypatch):
    """One-kernel path (rtc_device.hpp light_is_behind): where `light . normal < 0` Shape::lighting ignores the shadow test's answer
    (src/shape.rs:448-459), and a finite shadow ray cannot make the NaN t the reference would panic on, so the traversal is skipped: the
    same image bit for bit, the same ray counts (the ray is answered, not dropped), fewer tests.  Scenes with unbounded coordinates or
    space-flattening matrices keep tracing every shadow ray."""
    import torch
    from raytracer_challenge_amd.scene import Element, Matrix, ShapeArgs, World
    cam, world = scenes.chapter15_teapot("teapot_low.obj", 96, 54)
    monkeypatch.setenv("RTC_KERNEL", "1")
    st = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RTC_BACKFACE_SKIP", flag)
        dr = DeviceRenderer(emu, emu.build_world(world), cam, 0, _cpu_standin=True)
        out = torch.empty(cam.vsize * cam.hsize * 3, dtype=torch.float64)
        st[flag] = dr.render_rows(5, 0, 1, cam.vsize, out, count=True, sync=True)
        st[flag]["img"] = out.clone()
    assert torch.equal(st["0"]["img"], st["1"]["img"])
    for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract"):
        assert st["0"][k] == st["1"][k], k
    assert st["1"]["tri_tests"] < 0.9 * st["0"]["tri_tests"] and st["1"]["accel_nodes"] < st["0"]["accel_nodes"]   # (the teapot's far side)
    monkeypatch.delenv("RTC_BACKFACE_SKIP")
    assert_parity(emu, orc, world, cam, 5, label="teapot, lights behind surfaces skipped")
    # a sphere scaled by 1e40 (its world -> object matrix shrinks unit vectors to 1e-40: a = |d|^2 could underflow elsewhere): no skipping
    far = World(world.lights, list(world.elements) + [Element.sphere(ShapeArgs(transform=Matrix.translation(0.0, -1e41, 0.0) * Matrix.scaling(1e40, 1e40, 1e40)))])
    a = DeviceRenderer(emu, emu.build_world(far), cam, 0, _cpu_standin=True)
    out = torch.empty(cam.vsize * cam.hsize * 3, dtype=torch.float64)
    monkeypatch.setenv("RTC_BACKFACE_SKIP", "0")
    off = DeviceRenderer(emu, emu.build_world(far), cam, 0, _cpu_standin=True).render_rows(5, 0, 1, cam.vsize, out, count=True, sync=True)
    monkeypatch.delenv("RTC_BACKFACE_SKIP")
    on = a.render_rows(5, 0, 1, cam.vsize, out, count=True, sync=True)
    assert on["tri_tests"] == off["tri_tests"] and on["analytic_tests"] == off["analytic_tests"]
