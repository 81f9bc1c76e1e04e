#!/usr/bin/env python3
"""Derives tests/golden/example_image_edges.npz from the reference's images/example.png (the only rendered output the
reference ships): positions of the strong image edges around each of the four spheres and along the visible left part of
the disc's rim.  Only these measurements (a few thousand pixel coordinates) are committed, not the image.

Method (no oracle, no scene knowledge beyond where to look): Gaussian-smoothed RGB (sigma 1.5), Sobel gradients, non-maximum
suppression along the gradient direction; for each object the edge pixels above the 80th percentile of gradient magnitude
inside a generous search region -- an annulus 0.7..1.3 r0 around a centre and radius r0 EYEBALLED from the picture, or a box
for the disc rim.  tests/test_oracle_example_image.py aligns the oracle's silhouettes to these points.

Round 5 adds what the picture shows INSIDE its mirror and glass spheres (the only reference output in which light::reflect and
light::refract are at work): (a) the strong edges inside a disc of 0.93 r0 around each of the two spheres (coordinates, as
above) and (b) a one-bit "floor-like" classification of the pixels in each sphere's bounding box -- the disc floor is the only
dark red-brown surface of the scene (R > 1.5 G and R > 1.5 B on the sigma-1 smoothed picture; sky is blue-purple, lights are
yellow) -- bit-packed.  The test compares both with where the oracle's deterministic specular walk (orc_specular_ids) puts the
reflected / refracted floor and spheres."""
import os
import sys

import numpy as np
from PIL import Image
from scipy import ndimage

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/images/example.png"
im = np.asarray(Image.open(src).convert("RGB")).astype(np.float32)
H, W, _ = im.shape
sm = ndimage.gaussian_filter(im, (1.5, 1.5, 0))
gx = np.stack([ndimage.sobel(sm[..., c], axis=1) for c in range(3)], -1)
gy = np.stack([ndimage.sobel(sm[..., c], axis=0) for c in range(3)], -1)
g = np.sqrt((gx ** 2 + gy ** 2).sum(-1))
dom = np.argmax(np.abs(gx) + np.abs(gy), axis=-1)[..., None]
ang = np.arctan2(np.take_along_axis(gy, dom, -1)[..., 0], np.take_along_axis(gx, dom, -1)[..., 0])
dxn, dyn = np.round(np.cos(ang)).astype(int), np.round(np.sin(ang)).astype(int)
yy, xx = np.mgrid[0:H, 0:W]


def at(y, x):
    return g[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)]


nms = (g >= at(yy + dyn, xx + dxn)) & (g >= at(yy - dyn, xx - dxn))
# object: (index in the scene's declaration order, codelets.cpp:137-143; eyeballed centre; eyeballed radius)
SPHERES = {"tan_diffuse": (0, (168, 505), 97), "mirror": (1, (535, 400), 128), "glass": (2, (765, 537), 93),
           "magenta_coat": (4, (403, 585), 80)}
DISC_BOX = (0, 95, 480, 640)     # x0, x1, y0, y1: where the disc's rim is seen against the ground
out = {"width": np.int32(W), "height": np.int32(H), "disc_box": np.array(DISC_BOX, np.int32)}
for name, (index, (cx, cy), r0) in SPHERES.items():
    rr = np.hypot(xx - cx, yy - cy)
    region = (rr > 0.7 * r0) & (rr < 1.3 * r0)
    e = nms & region & (g > np.percentile(g[region], 80))
    ey, ex = np.nonzero(e)
    out[name] = np.stack([ex, ey], 1).astype(np.int16)
    out[name + "_index"] = np.int32(index)
x0, x1, y0, y1 = DISC_BOX
region = (xx >= x0) & (xx < x1) & (yy > y0) & (yy < y1)
e = nms & region & (g > np.percentile(g[region], 80))
ey, ex = np.nonzero(e)
out["disc_rim"] = np.stack([ex, ey], 1).astype(np.int16)
out["disc_rim_index"] = np.int32(5)
# ---- round 5: the interiors of the mirror and the glass sphere
sm1 = ndimage.gaussian_filter(im, (1.0, 1.0, 0))
floor_like = (sm1[..., 0] > 1.5 * sm1[..., 1]) & (sm1[..., 0] > 1.5 * sm1[..., 2]) & (sm1[..., 0] > 20)
for name in ("mirror", "glass"):
    _, (cx, cy), r0 = SPHERES[name]
    rr = np.hypot(xx - cx, yy - cy)
    region = rr < 0.93 * r0
    e = nms & region & (g > np.percentile(g[region], 70))
    ey, ex = np.nonzero(e)
    out[name + "_interior"] = np.stack([ex, ey], 1).astype(np.int16)
    x0, x1, y0, y1 = max(cx - int(1.4 * r0), 0), min(cx + int(1.4 * r0), W), max(cy - int(1.4 * r0), 0), min(cy + int(1.4 * r0), H)
    out[name + "_floor_box"] = np.array([x0, x1, y0, y1], np.int32)
    out[name + "_floor_bits"] = np.packbits(floor_like[y0:y1, x0:x1])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "example_image_edges.npz")
np.savez_compressed(path, **out)
print({k: (v.shape if hasattr(v, "shape") and v.ndim else int(v)) for k, v in out.items()}, os.path.getsize(path), "bytes")
