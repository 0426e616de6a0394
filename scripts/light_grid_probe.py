#!/usr/bin/env python3
"""Light grids on / off on both device paths: pixel and hit-record differences (none expected).  usage: light_grid_probe.py [hip|emu|simt] [n_prims] [hsize] [vsize]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_challenge_amd as rt
from raytracer_challenge_amd import scenes
which = sys.argv[1] if len(sys.argv) > 1 else "hip"
if which == "hip":
    be = rt.hip_backend()
else:
    from emu_lib import EMU_DIR
    from raytracer_challenge_amd.backend import Backend
    be = Backend(os.path.join(EMU_DIR, "_build", "librtc_emu_simt.so" if which == "simt" else "librtc_emu.so"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
h, v = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (480, 270)
cam, world = scenes.synthetic_analytic(n_primitives=n, seed=12345, hsize=h, vsize=v)
out = {}
for k in ("1", "4"):
    os.environ["RTC_KERNEL"] = k
    for flag in ("0", "1"):
        os.environ["RTC_LIGHT_GRID"] = flag
        out[k, flag] = be.render(be.build_world(world), cam, 5)
ref = out["1", "0"]
for key, (rgb, hits) in out.items():
    bad = np.nonzero((rgb != ref[0]).any(axis=1))[0]
    print(key, "hits equal", np.array_equal(hits, ref[1]), "rgb mismatches", bad.size, bad[:8], "max |d|", np.abs(rgb - ref[0]).max())
    for b in bad[:3]:
        print("   pixel", b, divmod(int(b), h), rgb[b], ref[0][b])
