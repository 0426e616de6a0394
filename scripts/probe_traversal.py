#!/usr/bin/env python3
"""Experiment (GPU box, RTC_PROBE variant build): closest-hit traversal alone vs register budget / occupancy.
usage: RTC_AMD_LIB=raytracer_challenge_amd/csrc/variants/librtc_amd_probe.so python3 scripts/probe_traversal.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_challenge_amd as rt  # noqa: E402
from raytracer_challenge_amd import scenes  # noqa: E402
from raytracer_challenge_amd.backend import HIT_DTYPE  # noqa: E402
from raytracer_challenge_amd.device import DeviceRenderer  # noqa: E402

hip = rt.hip_backend()
cam, world = scenes.synthetic_analytic(hsize=1920, vsize=1080)
nw = hip.build_world(world)
dev = DeviceRenderer(hip, nw, cam, 0)
lib = hip.lib
lib.rtc_probe_closest.restype = C.c_double
lib.rtc_probe_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p]

# primary rays through pixel centres (same construction as Camera::ray_for_pixel, numpy; only used as probe input)
inv = np.linalg.inv(np.array(cam.transform_matrix.m, dtype=np.float64).reshape(4, 4))
half_view = np.tan(cam.field_of_view / 2)
aspect = cam.hsize / cam.vsize
hw, hh = (half_view, half_view / aspect) if aspect >= 1 else (half_view * aspect, half_view)
ps = hw * 2 / cam.hsize
ys, xs = np.mgrid[0:cam.vsize, 0:cam.hsize]
# 8x8 tile order, as the render kernel assigns lanes
ty, tx = ys // 8, xs // 8
order = np.lexsort(((xs % 8).ravel(), (ys % 8).ravel(), tx.ravel(), ty.ravel()))
wx = hw - (xs.ravel()[order] + 0.5) * ps
wy = hh - (ys.ravel()[order] + 0.5) * ps
pix = np.stack([wx, wy, -np.ones_like(wx), np.ones_like(wx)], 1) @ inv.T
org = (inv @ np.array([0, 0, 0, 1.0]))[:3]
d = pix[:, :3] - org
d /= np.linalg.norm(d, axis=1, keepdims=True)
primary = np.ascontiguousarray(np.concatenate([np.broadcast_to(org, d.shape), d], 1))
n = primary.shape[0]


def run(rays, w, reps=5, want=False):
    hits = np.empty(rays.shape[0], dtype=HIT_DTYPE) if want else None
    ms = lib.rtc_probe_closest(dev.scene, rays.ctypes.data, rays.shape[0], w, reps, hits.ctypes.data if want else None)
    return ms, hits


ms, hits = run(primary, 2, 1, True)
t = hits["t"]
hit = hits["prim"] >= 0
rng = np.random.default_rng(7)
rd = rng.normal(size=(n, 3))
rd /= np.linalg.norm(rd, axis=1, keepdims=True)
p = primary[:, :3] + primary[:, 3:] * np.where(hit, t, 1.0)[:, None]
nrm_off = -primary[:, 3:] * 1e-3  # step back off the surface
secondary = np.ascontiguousarray(np.concatenate([p + nrm_off, rd], 1))
# mirror-ish: keep the primary direction's horizontal part, flip y (coherent secondary rays)
md = primary[:, 3:] * np.array([1.0, -1.0, 1.0])
mirror = np.ascontiguousarray(np.concatenate([p + nrm_off, md], 1))
for name, rays in (("primary (tile order)", primary), ("from hit points, mirrored dirs", mirror), ("from hit points, random dirs", secondary)):
    for w in (2, 6):
        ms, _ = run(rays, w)
        print("%-34s budget w=%d: %7.3f ms  %8.1f Mrays/s" % (name, w, ms, rays.shape[0] / ms / 1e3), flush=True)
