#!/usr/bin/env python3
"""GPU box: throughput with F frames in flight (F scenes = F streams + F sets of wavefront buffers, frames round-robin),
for rank 0's rows of an N-way partition of config 2.  usage: python3 scripts/inflight_probe.py"""
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
for n in (1, 8):
    rows = len(range(0, cam.vsize, n))
    for f in (1, 2, 3, 4):
        drs, outs = [], []
        for _ in range(f):
            nw = hip.build_world(world)
            dr = DeviceRenderer(hip, nw, cam, device=0)
            out = torch.empty(rows * cam.hsize * 3, dtype=torch.float64, device="cuda:0")
            dr.tune(5, 0, n, rows, out)
            drs.append((dr, nw))
            outs.append(out)
        k = 40
        for i in range(2 * f):
            drs[i % f][0].render_rows_async(5, 0, n, rows, outs[i % f])
        for dr, _ in drs:
            dr.sync()
        t0 = time.perf_counter()
        for i in range(k):
            drs[i % f][0].render_rows_async(5, 0, n, rows, outs[i % f])
        for dr, _ in drs:
            dr.sync()
        wall = (time.perf_counter() - t0) / k * 1e3
        print("N=%d frames in flight=%d: %.3f ms/frame (%s)" % (n, f, wall, drs[0][0].path_info()["path"]), flush=True)
        del drs, outs
