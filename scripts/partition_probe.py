#!/usr/bin/env python3
"""GPU box: what one rank of an N-GPU run does per frame, measured on one GPU.  For N in 1/2/4/8: part 0's
share of the frame under the two partitions — single rows interleaved by rank (band 1, round 2) and 8-row bands (round 3) — as
sequential launches (device time of one frame's kernels) and with F = 3 frames in flight on three scene copies (what bench.py
times).  The predicted N-GPU ratio is t(1) / t(N) of the pipelined column: the gather (RCCL, 50 MB / N per rank over xGMI) runs
behind the renders.
usage: python3 scripts/partition_probe.py [workload] [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracer_challenge_amd as rt  # noqa: E402
from raytracer_challenge_amd.device import DeviceRenderer  # noqa: E402
from raytracer_challenge_amd.parallel import rows_of  # noqa: E402
import bench  # noqa: E402

args = sys.argv[1:]
workload = args.pop(0) if args and not args[0].isdigit() else "config2"
hip = rt.hip_backend()
cam, world, desc = bench.make_workload(workload)
fuel = bench.default_fuel(workload)
print("# %s" % desc)
F = 3
base = {}
for n in [int(a) for a in args] or [1, 2, 4, 8]:
    for band in (1, 8):
        rows = len(rows_of(0, n, cam.vsize, band))
        drs = [DeviceRenderer(hip, hip.build_world(world), cam, device=0) for _ in range(F)]
        outs = [torch.empty(rows * cam.hsize * 3, dtype=torch.float64, device="cuda:0") for _ in range(F)]
        infos = [d.tune(fuel, 0, n, rows, o, band_rows=band) for d, o in zip(drs, outs)]
        dr, out = drs[0], outs[0]
        k = 30
        dr.record(2)
        for _ in range(k):
            dr.render_rows_async(fuel, 0, n, rows, out, band_rows=band)
        dr.record(3)
        dr.sync()
        seq = dr.elapsed_ms(2, 3) / k
        for i in range(2 * F):
            drs[i % F].render_rows_async(fuel, 0, n, rows, outs[i % F], band_rows=band)
        for d in drs:
            d.sync()
        k = 60
        t0 = time.perf_counter()
        for i in range(k):
            if i >= F:
                drs[i % F].wait(0)
            drs[i % F].render_rows_async(fuel, 0, n, rows, outs[i % F], band_rows=band)
            drs[i % F].record(0)
        for d in drs:
            d.sync()
        pipe = (time.perf_counter() - t0) / k * 1e3
        base.setdefault(band, pipe if n == 1 else None)
        print("N=%d band=%d rows=%4d path=%-10s sequential %.3f ms/frame   pipelined(F=3) %.3f ms/frame   predicted ratio vs N=1: %.2fx"
              % (n, band, rows, infos[0]["path"], seq, pipe, (base[band] or pipe) / pipe), flush=True)
        del drs, outs
