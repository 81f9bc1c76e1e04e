"""Analytic known-answer tests for the restated light:: routines.

The reference has no tests and external/light is absent, so these closed-form checks are what pins
the INFERRED functions of oracle/pt_oracle.c.
"""
import ctypes as C

import numpy as np
import pytest

f32 = np.float32


_KEEP = []


def v3(*x):
    a = np.ascontiguousarray(x, dtype=np.float32)
    _KEEP.append(a)  # ctypes receives raw addresses: keep every argument array alive
    return a


def test_trace_record_wire_format(oracle):
    # src/codelets/TraceRecord.hpp:7-19
    d = oracle.TRACE_DTYPE
    assert d.itemsize == 20
    assert [d.fields[k][1] for k in ("u", "v", "r", "g", "b", "sampleCount", "pathLength")] == [0, 2, 4, 8, 12, 16, 18]


def test_scene_constants_match_codelets(oracle):
    L = oracle.lib()
    exp = [((-1.8575, -0.98714, -3.6), 0.6, (2.0, 1.78, 1.1), 0), ((0.74795, -0.55, -4.3816), 1.05, (1, 1, 1), 1),
           ((1.9929, -1.08666, -3.23), 0.5, (.75, .75, .75), 2), ((-0.19931, -1.183, -2.75), 0.4, (1.6, .12, .782), 0),
           ((-0.19931, -1.183, -2.75), 0.4001, (1, 1, 1), 2), ((0, -1.6, -5.22), 3.5, (1.96, 1.52, 1.32), 0x100)]
    for i, (c, r, col, t) in enumerate(exp):
        cc, colr, rad, ty = v3(0, 0, 0), v3(0, 0, 0), C.c_float(), C.c_int32()
        L.orc_scene_object(i, cc.ctypes.data, C.byref(rad), colr.ctypes.data, C.byref(ty))
        np.testing.assert_allclose(cc, c, rtol=1e-6)
        np.testing.assert_allclose(colr, col, rtol=1e-6)
        assert rad.value == pytest.approx(r, rel=1e-6) and ty.value == t


def test_ray_sphere_and_disc_intersection(oracle):
    L = oracle.lib()
    o, d, c = v3(0, 0, 0), v3(0, 0, -1), v3(0, 0, -5)
    assert L.orc_intersect_sphere(o.ctypes.data, d.ctypes.data, c.ctypes.data, 1.0) == pytest.approx(4.0, rel=1e-6)
    # from inside: far root
    assert L.orc_intersect_sphere(c.ctypes.data, d.ctypes.data, c.ctypes.data, 1.0) == pytest.approx(1.0, rel=1e-6)
    # miss and behind
    assert L.orc_intersect_sphere(o.ctypes.data, v3(0, 1, 0).ctypes.data, c.ctypes.data, 1.0) == 0.0
    assert L.orc_intersect_sphere(o.ctypes.data, v3(0, 0, 1).ctypes.data, c.ctypes.data, 1.0) == 0.0
    # tangent-ish offset ray: distance = 5 - sqrt(1 - 0.6^2)
    o2 = v3(0.6, 0, 0)
    assert L.orc_intersect_sphere(o2.ctypes.data, d.ctypes.data, c.ctypes.data, 1.0) == pytest.approx(5 - 0.8, rel=1e-6)
    n, dc = v3(0, 1, 0), v3(0, -1.6, -5.22)
    down = v3(0, -1, 0)
    assert L.orc_intersect_disc(v3(0, 0, -5.22).ctypes.data, down.ctypes.data, n.ctypes.data, dc.ctypes.data, 3.5) == pytest.approx(1.6, rel=1e-6)
    assert L.orc_intersect_disc(v3(3.6, 0, -5.22).ctypes.data, down.ctypes.data, n.ctypes.data, dc.ctypes.data, 3.5) == 0.0  # outside radius
    assert L.orc_intersect_disc(v3(0, 0, -5.22).ctypes.data, v3(1, 0, 0).ctypes.data, n.ctypes.data, dc.ctypes.data, 3.5) == 0.0  # parallel
    assert L.orc_intersect_disc(v3(0, 0, -5.22).ctypes.data, v3(0, 1, 0).ctypes.data, n.ctypes.data, dc.ctypes.data, 3.5) == 0.0  # behind


def test_scene_intersect_nearest_and_clear_coat(oracle):
    L = oracle.lib()
    t, hp, nrm = C.c_float(), v3(0, 0, 0), v3(0, 0, 0)
    c = v3(-0.19931, -1.183, -2.75)
    d = (c / np.linalg.norm(c)).astype(f32)
    # camera ray through the front sphere's centre hits the refractive coat (r=0.4001, index 4) first
    idx = L.orc_scene_intersect(v3(0, 0, 0).ctypes.data, d.ctypes.data, C.byref(t), hp.ctypes.data, nrm.ctypes.data)
    assert idx == 4 and t.value == pytest.approx(np.linalg.norm(c) - 0.4001, rel=1e-5)
    np.testing.assert_allclose(nrm, -d, atol=1e-4)
    # continuing inward from that hit point the diffuse core (index 3) is 1e-4 away: epsilon must resolve it
    idx2 = L.orc_scene_intersect(hp.ctypes.data, d.ctypes.data, C.byref(t), hp.ctypes.data, nrm.ctypes.data)
    assert idx2 == 3 and t.value == pytest.approx(1e-4, rel=0.05)
    # straight up: nothing
    assert L.orc_scene_intersect(v3(0, 0, 0).ctypes.data, v3(0, 1, 0).ctypes.data, C.byref(t), hp.ctypes.data, nrm.ctypes.data) == -1


def test_reflect_identity(oracle):
    L = oracle.lib()
    d = v3(1, -1, 0) / f32(np.sqrt(2))
    L.orc_reflect(d.ctypes.data, v3(0, 1, 0).ctypes.data)
    np.testing.assert_allclose(d, v3(1, 1, 0) / np.sqrt(2), atol=1e-6)
    rng = np.random.default_rng(0)
    for _ in range(100):
        n = rng.standard_normal(3); n /= np.linalg.norm(n)
        d0 = rng.standard_normal(3); d0 /= np.linalg.norm(d0)
        d1 = d0.astype(f32).copy()
        n32 = n.astype(f32)
        L.orc_reflect(d1.ctypes.data, n32.ctypes.data)
        np.testing.assert_allclose(d1, d0 - 2 * np.dot(d0, n) * n, atol=2e-6)
        L.orc_reflect(d1.ctypes.data, n32.ctypes.data)  # involution
        np.testing.assert_allclose(d1, d0, atol=4e-6)


def test_refract_snell_and_total_internal_reflection(oracle):
    L = oracle.lib()
    n = v3(0, 0, 1)
    d = v3(0, 0, -1)
    assert L.orc_refract(d.ctypes.data, n.ctypes.data, 1.5, 0.99) == 1  # normal incidence passes straight through
    np.testing.assert_allclose(d, (0, 0, -1), atol=1e-6)
    # Schlick at normal incidence: R0 = ((1-1.5)/(1+1.5))^2 = 0.04
    d = v3(0, 0, -1); assert L.orc_refract(d.ctypes.data, n.ctypes.data, 1.5, 0.039) == 0
    np.testing.assert_allclose(d, (0, 0, 1), atol=1e-6)
    d = v3(0, 0, -1); assert L.orc_refract(d.ctypes.data, n.ctypes.data, 1.5, 0.041) == 1
    # Snell: sin(t2) = sin(t1)/1.5 entering
    th = np.deg2rad(40.0)
    d = v3(np.sin(th), 0, -np.cos(th))
    assert L.orc_refract(d.ctypes.data, n.ctypes.data, 1.5, 0.999) == 1
    assert np.hypot(d[0], d[1]) == pytest.approx(np.sin(th) / 1.5, rel=1e-5)
    # leaving glass above the critical angle asin(1/1.5) = 41.81 deg: always reflects
    for deg, expect in ((41.0, 1), (42.5, 0)):
        th = np.deg2rad(deg)
        d = v3(np.sin(th), 0, np.cos(th))  # travelling along +z with outward normal +z => inside the medium
        r = L.orc_refract(d.ctypes.data, n.ctypes.data, 1.5, 0.999999)
        assert r == expect
        if not expect:
            np.testing.assert_allclose(d, (np.sin(th), 0, -np.cos(th)), atol=1e-6)
        assert np.linalg.norm(d) == pytest.approx(1.0, abs=1e-6)


def test_hemisphere_and_diffuse_frame(oracle):
    L = oracle.lib()
    out = v3(0, 0, 0)
    L.orc_hemisphere(0.25, 0.0, out.ctypes.data)
    np.testing.assert_allclose(out, (np.sqrt(1 - 0.0625), 0, 0.25), atol=1e-6)
    rng = np.random.default_rng(1)
    # uniform hemisphere: E[cos] = 1/2, E[cos^2] = 1/3; samples stay in the normal's hemisphere
    for n in (v3(0, 1, 0), v3(1, 0, 0), v3(0.6, 0.0, -0.8), v3(-0.3, 0.9, 0.31622776)):
        n = (n / np.linalg.norm(n)).astype(f32)
        cos = []
        for u1, u2 in rng.random((4000, 2)):
            L.orc_diffuse_dir(n.ctypes.data, float(u1), float(u2), out.ctypes.data)
            assert np.linalg.norm(out) == pytest.approx(1.0, abs=2e-6)
            cos.append(float(np.dot(out, n)))
            assert cos[-1] == pytest.approx(u1, abs=2e-6)  # weight cos(theta) equals the first sample
        assert np.mean(cos) == pytest.approx(0.5, abs=0.02) and np.mean(np.square(cos)) == pytest.approx(1 / 3, abs=0.02)


def test_roulette_is_unbiased(oracle):
    L = oracle.lib()
    fac = C.c_float()
    assert L.orc_roulette(0.29, 0.3, C.byref(fac)) == 1
    assert L.orc_roulette(0.31, 0.3, C.byref(fac)) == 0 and fac.value == pytest.approx(1 / 0.7, rel=1e-6)
    us = (np.arange(2048) / 2048.0)  # the half grid of primary samples
    p = float(np.float16(0.3))
    vals = []
    for u in us:
        stop = L.orc_roulette(float(u), p, C.byref(fac))
        vals.append(0.0 if stop else fac.value)
    assert np.mean(vals) == pytest.approx(1.0, abs=2e-3)


def test_pixel_to_ray_pinhole(oracle):
    L = oracle.lib()
    out = v3(0, 0, 0)
    fov = float(np.float32(np.pi / 2))
    L.orc_pixel_to_ray(128.0, 128.0, 256, 256, fov, out.ctypes.data)
    np.testing.assert_allclose(out, (0, 0, -1), atol=1e-6)
    L.orc_pixel_to_ray(256.0, 0.0, 256, 256, fov, out.ctypes.data)  # right/top corner at 90 deg horizontal FOV
    np.testing.assert_allclose(out, (1, 1, -1), atol=1e-6)
    L.orc_pixel_to_ray(0.0, 1000.0, 1104, 1000, fov, out.ctypes.data)  # square pixels: y extent = h/w
    np.testing.assert_allclose(out, (-1, -1000 / 1104, -1), atol=1e-6)


def test_equirect_uv_for_axis_directions(oracle):
    # codelets.cpp:333-347: u = acos(y)/pi (vertical), v = (atan2(z,x)+azimuth wrapped)/2pi
    L = oracle.lib()
    uv = np.zeros(2, dtype=f32)
    cases = {(0, 1, 0): (0.0, None), (0, -1, 0): (1.0, None), (1, 0, 0): (0.5, 0.0), (0, 0, 1): (0.5, 0.25),
             (-1, 0, 0): (0.5, 0.5), (0, 0, -1): (0.5, 0.75)}
    for d, (u, v) in cases.items():
        L.orc_dir_to_uv(v3(*d).ctypes.data, 0.0, uv.ctypes.data)
        assert uv[0] == pytest.approx(u, abs=1e-6)
        if v is not None:
            assert uv[1] == pytest.approx(v, abs=1e-6)
    L.orc_dir_to_uv(v3(1, 0, 0).ctypes.data, float(np.float32(np.pi)), uv.ctypes.data)
    assert uv[1] == pytest.approx(0.5, abs=1e-6)
    L.orc_dir_to_uv(v3(0, 0, -1).ctypes.data, float(np.float32(np.pi)), uv.ctypes.data)  # 1.5pi + pi wraps to 0.5pi
    assert uv[1] == pytest.approx(0.25, abs=1e-6)


def test_aa_noise_distributions(oracle):
    out = np.zeros(2, dtype=f32)
    L = oracle.lib()
    for kind, check in ((oracle.AA_NORMAL, lambda x: (abs(x.mean()) < 0.03, abs(x.std() - 1) < 0.03)),
                        (oracle.AA_UNIFORM, lambda x: (x.min() >= -1 and x.max() <= 1, abs(x.std() - 1 / np.sqrt(3)) < 0.02)),
                        (oracle.AA_TRUNCATED_NORMAL, lambda x: (np.abs(x).max() <= 3.0, abs(x.mean()) < 0.03))):
        cfg = oracle.make_config(aa_noise_type=kind, seed=5)
        xs = []
        for i in range(4000):
            L.orc_aa_noise(C.byref(cfg), i % 97, i // 97, i, out.ctypes.data)
            xs.extend(out.tolist())
        xs = np.array(xs)
        assert all(check(xs))
        assert np.array_equal(xs.astype(np.float16).astype(np.float64), xs)  # values are halves
