#!/usr/bin/env python3
"""GPU box, RTC_DIAG variant (RTC_AMD_LIB=...diag.so RTC_DIAG_DUMP=1): one synchronous frame per device path -> [rtc-diag] lines on stderr.
usage: scripts/diag_run.py [workload] [kernel: 1|4] [fuel]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTC_KERNEL"] = sys.argv[2] if len(sys.argv) > 2 else "4"
import torch
import raytracer_challenge_amd as rt
from raytracer_challenge_amd.device import DeviceRenderer
import bench
cam, world, desc = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else "config2")
hip = rt.hip_backend()
nw = hip.build_world(world)
dr = DeviceRenderer(hip, nw, cam, 0)
out = torch.empty(cam.vsize * cam.hsize * 3, dtype=torch.float64, device="cuda:0")
fuel = int(sys.argv[3]) if len(sys.argv) > 3 else bench.default_fuel(sys.argv[1] if len(sys.argv) > 1 else "config2")
dr.render_rows(fuel, 0, 1, cam.vsize, out, count=False, sync=True)
st = dr.render_rows(fuel, 0, 1, cam.vsize, out, count=False, sync=True)
print("kernel_ms", st["kernel_ms"], file=sys.stderr)
