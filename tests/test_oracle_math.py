"""Pins the oracle's building blocks: Philox KAT (Random123), binary16 conversion, deterministic math."""
import ctypes as C

import numpy as np


def test_philox4x32_10_random123_known_answers(oracle):
    # kat_vectors of Random123 (philox4x32 10 rounds)
    assert list(oracle.philox([0, 0, 0, 0], [0, 0])) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(oracle.philox([0xffffffff] * 4, [0xffffffff] * 2)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert list(oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_half_conversion_matches_ieee(oracle):
    L = oracle.lib()
    allh = np.arange(65536, dtype=np.uint16)
    f = np.array([L.orc_h2f(int(h)) for h in allh], dtype=np.float32)
    ref = allh.view(np.float16).astype(np.float32)
    assert np.all((f == ref) | (np.isnan(f) & np.isnan(ref)))
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(30000) * 10.0 ** rng.integers(-9, 5, 30000)).astype(np.float32)
    x = np.concatenate([x, np.float32([0, -0.0, 65504, 65519.99, 65520, 1e-8, 5.96e-8, 2.98e-8, 2.9802325e-8, 6.1e-5])])
    h = np.array([L.orc_f2h(float(v)) for v in x], dtype=np.uint16)
    with np.errstate(over="ignore"):
        assert np.array_equal(h, x.astype(np.float16).view(np.uint16))
    # ties round to even: 1 + 2^-11 is halfway between 1 and 1+2^-10
    assert L.orc_f2h(1.0 + 2.0 ** -11) == 0x3c00 and L.orc_f2h(1.0 + 3 * 2.0 ** -11) == 0x3c02


def _ulps(got, ref):
    ref = np.float64(ref)
    return abs(np.float64(got) - ref) / np.spacing(np.float32(abs(ref)))


def test_deterministic_math_close_to_libm(oracle):
    """The reference calls libm (acosf/atan2: codelets.cpp:333-334); the +,-,*,/ versions stay within 3 ulp."""
    L = oracle.lib()
    for u in np.linspace(1e-7, 1, 5001, dtype=np.float32):
        assert _ulps(L.orc_dm_log(float(u)), np.log(np.float64(u))) <= 2.0 or abs(L.orc_dm_log(float(u))) < 1e-6
    s, c = C.c_float(), C.c_float()
    for u in np.linspace(0, 1, 5001, dtype=np.float32):
        L.orc_dm_sincos2pi(float(u), C.byref(s), C.byref(c))
        assert abs(s.value - np.sin(2 * np.pi * np.float64(u))) < 2e-7
        assert abs(c.value - np.cos(2 * np.pi * np.float64(u))) < 2e-7
    rng = np.random.default_rng(3)
    for y, x in rng.standard_normal((5000, 2)).astype(np.float32):
        assert _ulps(L.orc_dm_atan2(float(y), float(x)), np.arctan2(np.float64(y), np.float64(x))) <= 3.0
    for x in np.linspace(-1, 1, 5001, dtype=np.float32):
        assert abs(L.orc_dm_acos(float(x)) - np.arccos(np.float64(x))) < 4e-7
    assert L.orc_dm_atan2(0.0, 0.0) == 0.0
    assert L.orc_dm_acos(1.0) == 0.0 and abs(L.orc_dm_acos(-1.0) - np.pi) < 1e-6
