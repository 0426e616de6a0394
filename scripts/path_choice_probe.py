#!/usr/bin/env python3
"""GPU box: the measured device time of both paths for every scene the repo knows (the nine reference bins at 1920x1080 and the
BASELINE workloads) beside what the first-launch guess (rtc_scene.cpp first_guess) picks — the calibration table of the guess.
usage: python3 scripts/path_choice_probe.py > profiles/r3_path_choice.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracer_challenge_amd as rt  # noqa: E402
from raytracer_challenge_amd import scenes  # noqa: E402
from raytracer_challenge_amd.device import DeviceRenderer  # noqa: E402
import bench  # noqa: E402

hip = rt.hip_backend()
todo = [(n, lambda n=n: getattr(scenes, n)(1920, 1080) + ("reference bin at 1920x1080",), 5) for n in
        ("chapter11_glass_air_bubble", "chapter11_title", "chapter12_title", "chapter13_title", "chapter14_title", "cover", "chapter14_hexagon", "chapter14_benchmark")]
todo += [(w, lambda w=w: bench.make_workload(w), bench.default_fuel(w)) for w in ("config2", "config2_cones", "config3", "config3_high", "config4") + (("config5",) if "--config5" in sys.argv else ())]
print("%-28s %6s %9s %9s %8s %8s %7s" % ("scene", "prims", "1-kernel", "wavefront", "measured", "guess", "agree"))
for name, make, fuel in todo:
    cam, world, _ = make()
    nw = hip.build_world(world)
    dr = DeviceRenderer(hip, nw, cam, device=0)
    out = torch.empty(cam.vsize * cam.hsize * 3, dtype=torch.float64, device="cuda:0")
    dr.render_rows_async(fuel, 0, 1, cam.vsize, out)     # unmeasured + asynchronous: takes the guess
    st = dr.render_rows(fuel, 0, 1, cam.vsize, out, count=True, sync=True)
    dr2 = DeviceRenderer(hip, hip.build_world(world), cam, device=0)
    dr2.render_rows_async(fuel, 0, 1, cam.vsize, out)
    dr2.sync()
    # which path did the asynchronous first launch take?  the counting launch after it is synchronous and starts with the guess too
    guess = "wavefront" if st["n_launches"] > 1 else "one kernel"
    info = dr.tune(fuel, 0, 1, cam.vsize, out)
    print("%-28s %6d %9.3f %9.3f %10s %10s %5s" % (name, nw.primitive_count, info["one_kernel_ms"], info["wavefront_ms"], info["path"], guess, "yes" if guess == info["path"] else "NO"), flush=True)
    del dr, dr2, out
