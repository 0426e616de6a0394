#!/usr/bin/env python3
"""Generates tests/golden/*.samples.json from the reference's committed 8-bit renders (image/*.png).

Run in the build container only (needs /root/reference and PIL); the JSON fixtures it writes are committed
and are what travels to the GPU box.  A fixture is data: for each image, the native camera size and N (+M from the
lit region when the image is mostly background) sampled pixels [x, y, r, g, b] (gray PNGs have r = g = b).  Sampling is deterministic (numpy PCG64, seed
= 20241004 + index) so the files can be regenerated bit-for-bit.
"""
import json
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference/image"
OUT = os.path.dirname(os.path.abspath(__file__))
# new images are appended so that the per-image seeds (20241004 + index) of the existing fixtures do not change
IMAGES = ["chapter11_glass_air_bubble", "chapter11_title", "chapter14_benchmark", "chapter14_hexagon", "chapter15_teapot",
          "chapter12_title", "chapter13_title", "chapter14_title", "cover"]
N = 1500
M = 1000


def main():
    for k, name in enumerate(IMAGES):
        im = Image.open(os.path.join(REF, name + ".png"))
        w, h = im.size
        a = np.asarray(im.convert("RGB"))
        rng = np.random.Generator(np.random.PCG64(20241004 + k))
        xs = rng.integers(0, w, N)
        ys = rng.integers(0, h, N)
        # images that are mostly background: add M more samples drawn from the non-black pixels only
        lit = np.flatnonzero(a.reshape(-1, 3).max(1) > 0)
        if lit.size < 0.5 * w * h:
            pick = lit[rng.integers(0, lit.size, M)]
            xs = np.concatenate([xs, pick % w])
            ys = np.concatenate([ys, pick // w])
        samples = [[int(x), int(y)] + [int(c) for c in a[y, x]] for x, y in zip(xs, ys)]
        doc = {"source": "image/%s.png" % name, "hsize": w, "vsize": h, "mode": im.mode, "samples": samples}
        with open(os.path.join(OUT, name + ".samples.json"), "w") as f:
            json.dump(doc, f, separators=(",", ":"))
        print(name, w, h, im.mode, len(samples))


if __name__ == "__main__":
    sys.exit(main())
