#!/usr/bin/env python3
"""GPU box: what one rank of an N-GPU run does per frame, measured on one GPU (rank 0's interleaved rows of config 2):
device time and wall time per frame for each device path, to see how much of an N-way strong-scaling step is host overhead.
usage: python3 scripts/partition_probe.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracer_challenge_amd as rt  # noqa: E402
from raytracer_challenge_amd import scenes  # noqa: E402
from raytracer_challenge_amd.device import DeviceRenderer  # noqa: E402

hip = rt.hip_backend()
cam, world = scenes.synthetic_analytic()
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    rows = len(range(0, cam.vsize, n))
    for kernel in ("1", "4", ""):
        if kernel:
            os.environ["RTC_KERNEL"] = kernel
        else:
            os.environ.pop("RTC_KERNEL", None)
        nw = hip.build_world(world)  # RTC_KERNEL is read when the scene is created
        dr = DeviceRenderer(hip, nw, cam, device=0)
        out = torch.empty(rows * cam.hsize * 3, dtype=torch.float64, device="cuda:0")
        info = dr.tune(5, 0, n, rows, out)
        for _ in range(3):
            dr.render_rows_async(5, 0, n, rows, out)
        dr.sync()
        k = 30
        dr.record(2)
        t0 = time.perf_counter()
        for _ in range(k):
            dr.render_rows_async(5, 0, n, rows, out)
        dr.record(3)
        dr.sync()
        wall = (time.perf_counter() - t0) / k * 1e3
        print("N=%d rows=%4d RTC_KERNEL=%-4s path=%-10s device %.3f ms/frame  wall %.3f ms/frame" % (n, rows, kernel or "auto", info["path"], dr.elapsed_ms(2, 3) / k, wall), flush=True)
