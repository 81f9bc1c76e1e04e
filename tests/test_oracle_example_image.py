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


def _floor_like(f, name):
    """The fixture's one-bit "floor-like" classification of the picture inside a sphere's bounding box, as a full-size mask."""
    x0, x1, y0, y1 = [int(v) for v in f[name + "_floor_box"]]
    H, W = int(f["height"]), int(f["width"])
    bits = np.unpackbits(f[name + "_floor_bits"])[: (y1 - y0) * (x1 - x0)].reshape(y1 - y0, x1 - x0).astype(bool)
    full = np.zeros((H, W), dtype=bool)
    full[y0:y1, x0:x1] = bits
    return full


def _interior_scores(oracle, f, name, index, fov, **variant):
    """(IoU of the predicted and the pictured floor region, fraction of the predicted region boundaries that lie within
    3 px of a strong picture edge) inside the silhouette of sphere `index`, for the oracle's deterministic specular walk."""
    from scipy import ndimage
    W, H = int(f["width"]), int(f["height"])
    inside = ndimage.binary_erosion(oracle.object_ids(W, H, fov) == index, iterations=6)
    ids, _ = oracle.specular_ids(W, H, fov, **variant)
    predicted, pictured = (ids == 5) & inside, _floor_like(f, name) & inside
    iou = float((predicted & pictured).sum()) / float((predicted | pictured).sum())
    bd = np.zeros_like(inside)
    bd[:-1, :] |= ids[:-1, :] != ids[1:, :]
    bd[:, :-1] |= ids[:, :-1] != ids[:, 1:]
    bd &= inside
    edges = np.zeros_like(inside)
    pts = f[name + "_interior"]
    edges[pts[:, 1].astype(int), pts[:, 0].astype(int)] = True
    near_edge = ndimage.distance_transform_edt(~edges) <= 3.0
    return iou, (float(near_edge[bd].mean()) if bd.any() else 0.0)


def test_reflect_and_refract_against_the_inside_of_the_mirror_and_glass_spheres(oracle):
    """Pins the INFERRED light::reflect and light::refract (codelets.cpp:205-213) against the one piece of reference output
    that shows them at work: what images/example.png shows INSIDE its mirror sphere (the reflected tan, magenta and glass
    spheres and the reflected floor with its rim) and inside its glass sphere (the floor seen upside down through two
    refractions).  The oracle follows every pixel's central ray deterministically through the specular objects
    (orc_specular_ids, at the 81 degrees the silhouettes fitted) and must put the floor where the picture has its only
    red-brown surface, and its region boundaries on the picture's edges.  Negative controls, scores measured when the fixture
    was made (IoU of the floor region / predicted boundaries on picture edges):

        mirror sphere   reflect as restated   0.54 / 0.61      reflection mirrored about the surface   0.07 / 0.13
        glass sphere    eta = 1/1.5 entering  0.66             eta INVERTED                            0.16
        glass sphere    n = 1.5 (the default) 0.66             n = 1.33                                0.71   (NOT discriminated)

    So the picture pins the reflection formula and which way Snell's ratio goes, and it does NOT pin the refractive index:
    the floor region prefers 1.33-1.45 over the 1.5 default by a margin (0.04 IoU) smaller than the classifier's noise, and the
    edge score cannot arbitrate inside the glass (0.23 at 1.5, 0.59 at 1.33, but 0.53 for the INVERTED eta: the refracted
    skyline's own edges decide it, not the floor's rim) -- the picture was rendered with options nobody recorded (882 x 720, another NIF; --refractive-index is a command-line option,
    PathTracerApp.cpp:804).  Still unpinned after this test: the roulette comparison, the diffuse weight, the Fresnel
    probability (what decides WHEN a ray refracts), the 1.15 gain's side -- radiance, not geometry."""
    f = np.load(EDGES)
    fov = np.radians(81.0)
    good_iou, good_edges = _interior_scores(oracle, f, "mirror", 1, fov)
    bad_iou, bad_edges = _interior_scores(oracle, f, "mirror", 1, fov, reflect_variant=1)
    assert good_iou > 0.45 and good_edges > 0.5, (good_iou, good_edges)
    assert bad_iou < 0.2 and bad_edges < 0.25, (bad_iou, bad_edges)           # the wrong reflection FAILS
    assert good_iou > 4 * bad_iou and good_edges > 2 * bad_edges

    glass_iou, _ = _interior_scores(oracle, f, "glass", 2, fov)
    inverted_iou, _ = _interior_scores(oracle, f, "glass", 2, fov, refractive_index=1.0 / 1.5)
    assert glass_iou > 0.55, glass_iou
    assert inverted_iou < 0.3 and glass_iou > 3 * inverted_iou, (glass_iou, inverted_iou)   # eta inverted FAILS
    # the mirror does not care about the index; the glass sphere does, weakly: recorded, not asserted as a pin
    water_iou, _ = _interior_scores(oracle, f, "glass", 2, fov, refractive_index=1.33)
    assert abs(water_iou - glass_iou) < 0.1, (water_iou, glass_iou)           # 1.33 and 1.5 are NOT told apart by this picture
    dense_iou, _ = _interior_scores(oracle, f, "glass", 2, fov, refractive_index=2.0)
    assert dense_iou < glass_iou                                              # ... but the index is not free either


def test_oracle_refraction_against_an_independent_numpy_restatement(oracle):
    """orc_specular_ids' walk through the glass sphere (enter: refract with eta = 1/n, leave: refract with eta = n, Snell in
    vector form, total internal reflection reflects) restated in numpy in float64, independently of the oracle's C: the set of
    pixels that see the floor THROUGH the glass sphere agrees for the scene's index and for another one.  (Differences are
    confined to the silhouettes of the small spheres the numpy version ignores and to boundary pixels.)"""
    from scipy import ndimage
    W, H, fov = 882, 720, np.radians(81.0)
    inside = ndimage.binary_erosion(oracle.object_ids(W, H, fov) == 2, iterations=6)
    ys, xs = np.nonzero(inside)
    tx = np.tan(fov / 2)
    ty = (H / W) * tx
    d = np.stack([((2 * xs - W) / W) * tx, -(((2 * ys - H) / H) * ty), -np.ones(xs.size)], 1)
    d /= np.linalg.norm(d, axis=1)[:, None]
    C, R = np.array([1.9929, -1.08666, -3.23]), 0.5                     # codelets.cpp:114

    def sphere_t(o, dirs, near):
        oc = o - C
        b = 2 * (oc * dirs).sum(1)
        disc = np.sqrt(np.maximum(b * b - 4 * ((oc * oc).sum(1) - R * R), 0))
        return (-b - disc) / 2 if near else (-b + disc) / 2

    def bend(dirs, n, eta):                                              # n faces against dirs
        c1 = -(n * dirs).sum(1)
        c2 = 1 - eta * eta * (1 - c1 * c1)
        out = np.where((c2 <= 0)[:, None], dirs + n * (2 * c1)[:, None], dirs * eta + n * (eta * c1 - np.sqrt(np.maximum(c2, 0)))[:, None])
        return out / np.linalg.norm(out, axis=1)[:, None]

    for index in (1.5, 1.33):
        p = d * sphere_t(np.zeros_like(d), d, True)[:, None]
        d1 = bend(d, (p - C) / R, 1 / index)
        p2 = p + d1 * sphere_t(p + d1 * 1e-6, d1, False)[:, None]
        d2 = bend(d1, -(p2 - C) / R, index)
        t = (-1.6 - p2[:, 1]) / d2[:, 1]                                 # the disc's plane (codelets.cpp:121)
        q = p2 + d2 * t[:, None]
        floor = np.zeros((H, W), dtype=bool)
        floor[ys, xs] = (t > 0) & (q[:, 0] ** 2 + (q[:, 2] + 5.22) ** 2 <= 3.5 ** 2)
        ids, _ = oracle.specular_ids(W, H, fov, refractive_index=index)
        agree = ((ids == 5) == floor)[inside].mean()
        assert agree > 0.985, (index, agree)
