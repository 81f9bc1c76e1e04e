#!/usr/bin/env python3
"""Derives tests/golden/example_image_silhouette.json from the reference's images/example.png (the only rendered
output the reference ships: a qualitative pin for scene geometry and the inferred camera model).

Segments the magenta clear-coat sphere (objects 3+4 of src/codelets/codelets.cpp:115-116,140-141: the only strongly
magenta region, G << R and B comparable to R) and records its silhouette extents.  Only these measurements are
committed, not the image."""
import json
import os
import sys

import numpy as np
from PIL import Image
from scipy import ndimage

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/images/example.png"
im = np.asarray(Image.open(src).convert("RGB")).astype(np.float32)
H, W, _ = im.shape
R, G, B = im[..., 0], im[..., 1], im[..., 2]
mask = (G < 0.33 * R) & (B > 0.45 * R) & (R > 15)
lab, n = ndimage.label(mask)
sizes = ndimage.sum(mask, lab, range(1, n + 1))
blob = ndimage.binary_fill_holes(lab == (int(np.argmax(sizes)) + 1))
ys, xs = np.nonzero(blob)
# the right flank carries bright specular highlights that fail the hue test: keep left, top and bottom extents only
out = {"image": "images/example.png", "width": W, "height": H, "area": int(blob.sum()),
       "left": int(xs.min()), "top": int(ys.min()), "bottom": int(ys.max()),
       "note": "silhouette extents (pixels) of the magenta clear-coat sphere; right edge unreliable (highlights)"}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "example_image_silhouette.json")
json.dump(out, open(path, "w"), indent=1)
print(out)
