#!/usr/bin/env python3
"""GPU box: host binned-SAH build vs device LBVH build (RTC_DEVICE_BVH=1) of a mesh accelerator: build seconds ([rtc-timing] lines on
stderr) and the frame time each tree gives.  usage: RTC_TIMING=1 scripts/device_bvh_probe.py [config5|config4|config3_high]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_challenge_amd as rt
from raytracer_challenge_amd.device import DeviceRenderer
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "config5"
cam, world, desc = bench.make_workload(wl)
fuel = bench.default_fuel(wl)
hip = rt.hip_backend()
out = torch.empty(cam.vsize * cam.hsize * 3, dtype=torch.float64, device="cuda:0")
for flag in ("0", "1"):
    os.environ["RTC_DEVICE_BVH"] = flag
    nw = hip.build_world(world)
    t0 = time.perf_counter()
    dr = DeviceRenderer(hip, nw, cam, 0)          # flatten + accelerator build + upload
    t_create = time.perf_counter() - t0
    dr.tune(fuel, 0, 1, cam.vsize, out)
    ms = [dr.render_rows(fuel, 0, 1, cam.vsize, out)["kernel_ms"] for _ in range(5)]
    print("%s RTC_DEVICE_BVH=%s: scene create %.3f s, frame %.3f ms (%s), accelerator %s" % (wl, flag, t_create, min(ms), dr.path_info()["path"], dr.info()), file=sys.stderr, flush=True)
