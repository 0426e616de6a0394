"""SURVEY §8f rank 2, second half: the mesh accelerator's tree built ON THE DEVICE (RTC_DEVICE_BVH=1: linear BVH, csrc/bvh_device.hip)
with the host's binned-SAH build as the checker.  The accelerator is results-neutral, so the same scene rendered through either tree
must give the same pixels and primary-hit records, bit for bit — and both must match the oracle."""
import ctypes as C

import numpy as np
import pytest

import cases
from parity import assert_parity
from raytracer_challenge_amd import scenes

pytestmark = pytest.mark.gpu


def built_on_device(hip, nw):
    lib = hip.lib
    lib.rtw_world_scene.restype = C.c_void_p
    lib.rtw_world_scene.argtypes = [C.c_void_p, C.c_int]
    lib.rtc_scene_bvh_built_on_device.restype = C.c_int
    lib.rtc_scene_bvh_built_on_device.argtypes = [C.c_void_p]
    return lib.rtc_scene_bvh_built_on_device(lib.rtw_world_scene(nw.handle, 0))


@pytest.mark.parametrize("name", ["teapot_low", "teapot_high", "heightfield_50k"])
def test_device_built_tree_renders_the_same_bits(hip, orc, name, monkeypatch, tmp_path):
    if name == "heightfield_50k":
        cam, world = scenes.synthetic_mesh(str(tmp_path / "hf.obj"), nx=160, nz=160, hsize=256, vsize=144)   # 50 562 triangles
    else:
        cam, world = scenes.chapter15_teapot(name + ".obj", 256, 144)
    monkeypatch.setenv("RTC_DEVICE_BVH_MIN", "64")
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RTC_DEVICE_BVH", flag)
        nw = hip.build_world(world)
        out[flag] = hip.render(nw, cam, 5)
        assert (built_on_device(hip, nw) > 0) == (flag == "1"), (name, flag)
    assert np.array_equal(out["0"][1], out["1"][1]), name
    assert np.array_equal(out["0"][0], out["1"][0]), name
    if name == "teapot_low":   # and against the oracle (flat triangle list, no accelerator at all)
        assert_parity(hip, orc, world, cam, 5, label="teapot_low, tree built on the device")


def test_device_build_of_a_million_triangles(hip, tmp_path, monkeypatch):
    """Config 5's mesh (999 698 triangles): both builds, same sampled pixels at 4K fuel 8; build times on stderr with RTC_TIMING=1."""
    path = str(tmp_path / "heightfield_708.obj")
    cam, world = scenes.synthetic_mesh(path)
    idx = np.arange(0, 3840 * 2160, 1013, dtype=np.uint64)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RTC_DEVICE_BVH", flag)
        nw = hip.build_world(world)
        out[flag] = hip.render(nw, cam, 8, idx)
        assert (built_on_device(hip, nw) > 0) == (flag == "1")
    assert np.array_equal(out["0"][1], out["1"][1]) and np.array_equal(out["0"][0], out["1"][0])
    # the default (RTC_DEVICE_BVH unset): a mesh of >= 100 000 triangles is built on the device, a small one on the host
    monkeypatch.delenv("RTC_DEVICE_BVH")
    nw = hip.build_world(world)
    got = hip.render(nw, cam, 8, idx)
    assert built_on_device(hip, nw) > 0
    assert np.array_equal(got[1], out["0"][1]) and np.array_equal(got[0], out["0"][0])
    cam2, world2 = scenes.chapter15_teapot("teapot_high.obj", 64, 36)
    nw2 = hip.build_world(world2)
    hip.render(nw2, cam2, 1)
    assert built_on_device(hip, nw2) == 0
