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


EDGES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_image_edges.npz")
OBJECTS = ("tan_diffuse", "mirror", "glass", "magenta_coat", "disc_rim")


def _boundary(ids, index, box=None):
    """Boundary pixels of an object's VISIBLE silhouette in the oracle's object-id image (for the disc: its rim against the
    environment inside the search box; the clear-coat shell 4 and the sphere 3 inside it are one silhouette)."""
    from scipy import ndimage
    mask = (ids == index) | ((index == 4) & (ids == 3))
    bd = mask & ~ndimage.binary_erosion(mask)
    if box is not None:
        x0, x1, y0, y1 = box
        yy, xx = np.mgrid[0:ids.shape[0], 0:ids.shape[1]]
        bd &= ndimage.binary_dilation(ids == -1) & (xx >= x0) & (xx < x1) & (yy > y0) & (yy < y1)
    bd[:2] = 0; bd[-2:] = 0; bd[:, :2] = 0; bd[:, -2:] = 0
    return bd


def _scorer(ids, fixture, name):
    """score(dx, dy) = image edge points within 1.5 px of the oracle's silhouette boundary shifted by (dx, dy), per boundary pixel."""
    from scipy import ndimage
    bd = _boundary(ids, int(fixture[name + "_index"]), fixture["disc_box"] if name == "disc_rim" else None)
    if not bd.any():
        return lambda dx=0, dy=0: 0.0
    near = ndimage.distance_transform_edt(~bd) < 1.5
    H, W = ids.shape
    ex, ey = fixture[name][:, 0].astype(int), fixture[name][:, 1].astype(int)
    n = float(bd.sum())
    return lambda dx=0, dy=0: float(near[np.clip(ey - dy, 0, H - 1), np.clip(ex - dx, 0, W - 1)].sum()) / n


def test_all_five_silhouettes_align_with_the_reference_image(oracle):
    """Pins Scene::intersect (five spheres + disc, declaration order, occlusion: codelets.cpp:111-144,183), the scene
    constants and the INFERRED pixelToRay (horizontal FOV, square pixels, z = -1: codelets.cpp:73-75) against the one image
    the reference ships.  The image's --fov is unknown (it was rendered through the remote UI), so it is the single fitted
    parameter: ONE fov has to put all five silhouettes onto the image's edges at once.

    Measured when the fixture was made: joint optimum at 81.0 degrees; at that fov every object's best alignment is within
    (dx, dy) = (-1,-1), (0,0), (-1,-1), (0,0), (2,-3) pixels of the oracle's silhouette in the 882 x 720 image.  A camera with
    the vertical scale of the alternative model (ty = tan((h/w) fov/2), 7 % smaller) would displace the lower spheres by
    ~15 pixels.  This pins geometry and camera; radiance values remain unpinned (different NIF, unknown exposure)."""
    f = np.load(EDGES)
    W, H = int(f["width"]), int(f["height"])
    fovs = np.arange(78.0, 86.01, 0.25)
    totals, per_object = [], []
    for deg in fovs:
        ids = oracle.object_ids(W, H, np.radians(deg))
        s = [_scorer(ids, f, n)() for n in OBJECTS]
        per_object.append(s)
        totals.append(sum(s))
    totals, per_object = np.array(totals), np.array(per_object)
    best = int(np.argmax(totals))
    assert 80.0 <= fovs[best] <= 82.0
    assert totals[best] > 2.5 * np.median(totals)                         # a sharp optimum, not a plateau
    # every object on its own prefers the same fov (within half a degree): one camera explains all of them
    for k, name in enumerate(OBJECTS):
        assert abs(fovs[int(np.argmax(per_object[:, k]))] - fovs[best]) <= 0.5, name
    ids = oracle.object_ids(W, H, np.radians(fovs[best]))
    assert set(np.unique(ids)) == {-1, 0, 1, 2, 3, 4, 5} or set(np.unique(ids)) == {-1, 0, 1, 2, 4, 5}
    for name in OBJECTS:
        score = _scorer(ids, f, name)
        scores = {(dx, dy): score(dx, dy) for dy in range(-8, 9) for dx in range(-8, 9)}
        (dx, dy), peak = max(scores.items(), key=lambda kv: kv[1])
        assert abs(dx) <= 3 and abs(dy) <= 3, (name, dx, dy)
        assert scores[(0, 0)] >= 0.8 * peak, name                         # unshifted is (nearly) as good as the best shift
        assert peak > 1.8 * np.median(list(scores.values())), name
    # occlusion order as declared: the magenta sphere hides part of the mirror sphere, the mirror sphere hides the disc's far rim
    assert (ids[560:600, 400:410] != 1).all() and (ids[380:420, 520:560] == 1).all()
