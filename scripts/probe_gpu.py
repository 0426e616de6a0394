import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import torch
import raytracer_challenge_amd as rt
from raytracer_challenge_amd.device import DeviceRenderer
from oracle_lib import oracle
import cases
hip = rt.hip_backend(); o = oracle()
os.environ["RTC_KERNEL"] = "1"
cam, w = cases.SMALL_CASES["all_primitives"]()
ro, ho = o.render(o.build_world(w), cam, 5)
nw = hip.build_world(w)
rgb, _ = hip.render(nw, cam, 5, want_hits=False)
print("mode0 no-hits: max dRGB %.3e nonzero %d" % (np.abs(rgb-ro).max(), np.count_nonzero(rgb)))
rgb, hits = hip.render(nw, cam, 5, want_hits=True)
print("mode0 hits:    max dRGB %.3e nonzero %d" % (np.abs(rgb-ro).max(), np.count_nonzero(rgb)))
idx = np.arange(cam.hsize*cam.vsize, dtype=np.uint64)
rgb, hits = hip.render(nw, cam, 5, idx)
print("mode1 list:    max dRGB %.3e hit mismatch %d" % (np.abs(rgb-ro).max(), int((hits["prim"]!=ho["prim"]).sum())))
dr = DeviceRenderer(hip, nw, cam, 0)
out = torch.zeros(cam.hsize*cam.vsize*3, dtype=torch.float64, device="cuda:0")
st = dr.render_rows(5, 0, 1, cam.vsize, out, count=True)
print("mode2 rows:    max dRGB %.3e" % np.abs(out.cpu().numpy().reshape(-1,3)-ro).max(), st["pixels"], st["rays_primary"])
st = dr.render_rows(5, 0, 1, cam.vsize, out, count=False)
print("mode2 rows nc: max dRGB %.3e" % np.abs(out.cpu().numpy().reshape(-1,3)-ro).max())
