"""Randomised parity sweep: synthetic scenes with random primitive counts, kinds (cones on / off), grouping, lights (some inside the
cluster, some on the floor plane), fuels and resolutions, both device paths against the oracle — hit records bit-exact, colours within
1e-5.  Light grids, the container passes' point test and the group gates all decide per ray which exact tests run; a scene generator
that nobody tuned the kernels on is the cheapest way to catch a wrong decision.  RTC_FUZZ_SEEDS=<n> runs more seeds on the GPU."""
import os

import numpy as np
import pytest

from parity import assert_parity
from raytracer_challenge_amd import scenes
from raytracer_challenge_amd.scene import Color, PointLight, Vector


def random_case(seed, sizes=((64, 36), (96, 54), (128, 72)), counts=(17, 40, 96, 200, 512)):
    rng = np.random.default_rng(seed)
    n = int(rng.choice(counts))
    cones, grouped = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    h, v = (int(x) for x in sizes[int(rng.integers(0, len(sizes)))])
    cam, world = scenes.synthetic_analytic(n_primitives=n, seed=int(rng.integers(1, 1 << 30)), cones=cones, grouped=grouped, hsize=h, vsize=v)
    for _ in range(int(rng.integers(0, 3))):   # lights inside the cloud of primitives
        p = rng.uniform(-20, 20, 3) * np.array([1.0, 0.5, 1.0]) + np.array([0.0, 11.0, 5.0])
        world.lights.append(PointLight(Color.new(*rng.uniform(0.1, 0.6, 3)), Vector.point(*p)))
    if rng.random() < 0.3:                      # a light ON the floor plane
        world.lights.append(PointLight(Color.new(0.2, 0.2, 0.2), Vector.point(float(rng.uniform(-10, 10)), 0.0, float(rng.uniform(-10, 10)))))
    fuel = int(rng.choice([0, 1, 3, 5, 7]))
    label = "fuzz seed %d (n=%d cones=%s grouped=%s lights=%d fuel=%d %dx%d)" % (seed, n, cones, grouped, len(world.lights), fuel, h, v)
    return cam, world, fuel, label


@pytest.fixture(scope="module")
def emu():
    from emu_lib import emu as _emu
    return _emu()


@pytest.mark.parametrize("seed", [2001, 2002, 2003])
def test_random_scenes_in_the_emulator(emu, orc, seed, monkeypatch):
    cam, world, fuel, label = random_case(seed, sizes=((48, 27), (64, 36)), counts=(17, 40, 96))
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        assert_parity(emu, orc, world, cam, min(fuel, 3), label=label + " path " + path)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(1000, 1000 + int(os.environ.get("RTC_FUZZ_SEEDS", "12")))))
def test_hip_random_scenes(hip, orc, seed, monkeypatch):
    cam, world, fuel, label = random_case(seed)
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        assert_parity(hip, orc, world, cam, fuel, label=label + " path " + path)
