"""Qualitative pin against the one rendered image the reference ships (images/example.png, rendered with an unknown
--fov through the remote UI): the silhouette of the magenta clear-coat sphere constrains the INFERRED camera model."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_image_silhouette.json")


def _tangent_slopes(c_axis, c_z, radius):
    """Slopes tan(angle) of the two planes through the camera axis that touch a sphere (exact image-plane extents)."""
    rho = np.hypot(c_axis, c_z)
    a = np.arctan2(c_axis, -c_z)
    b = np.arcsin(radius / rho)
    return np.tan(a - b), np.tan(a + b)


def test_magenta_sphere_silhouette_supports_square_pixel_camera(oracle):
    m = json.load(open(GOLD))
    W, H = m["width"], m["height"]
    import ctypes as C
    centre = np.zeros(3, np.float32); col = np.zeros(3, np.float32); rad = C.c_float(); ty_ = C.c_int32()
    oracle.lib().orc_scene_object(4, centre.ctypes.data, C.byref(rad), col.ctypes.data, C.byref(ty_))
    cx, cy, cz = [float(v) for v in centre]
    x_lo, _ = _tangent_slopes(cx, cz, rad.value)          # left edge of the silhouette (x / -z)
    y_lo, y_hi = _tangent_slopes(-cy, cz, rad.value)      # top and bottom edges (image y grows downwards)
    # pixel = centre + slope / t * half-size  ->  t = slope * half / (pixel - centre)
    tx = x_lo * (W / 2) / (m["left"] + 0.5 - W / 2)
    ty_top = y_lo * (H / 2) / (m["top"] + 0.5 - H / 2)
    ty_bot = y_hi * (H / 2) / (m["bottom"] + 0.5 - H / 2)
    ty = 0.5 * (ty_top + ty_bot)
    assert abs(ty_top - ty_bot) / ty < 0.06                # top and bottom agree on one vertical scale
    assert 0.7 < tx < 1.0                                  # a horizontal FOV of 70-90 degrees
    ratio = ty / tx
    square = H / W                                         # ty = (h/w) tx          (oracle's pixelToRay)
    smallpaint = np.tan(square * np.arctan(tx)) / tx       # ty = tan((h/w) fov/2)  (the alternative)
    assert abs(ratio - square) < abs(ratio - smallpaint)
    assert abs(ratio - square) / square < 0.04
    # and the oracle's pixelToRay reproduces those extents at the fitted fov
    fov = float(np.float32(2 * np.arctan(tx)))
    out = np.zeros(3, np.float32)
    oracle.lib().orc_pixel_to_ray(float(m["left"]) + 0.5, float(m["top"]) + 0.5, W, H, fov, out.ctypes.data)
    assert abs(out[0] - x_lo) < 0.01 and abs(-out[1] - y_lo) < 0.03
