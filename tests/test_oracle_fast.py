"""The TIMING build of the oracle (oracle/libpt_oracle_fast.so: -O3 -march=native -fopenmp, contraction allowed) against the
strict build (the parity checker).  The timing build exists only for bench.py's cpu_baseline leg;
these tests show that what it times is the same computation: the NIF agrees within far less than the stated NIF tolerance
(the matmul keeps the strict build's k-ordered FMA chain per output), and a render agrees path for path except where a
contracted a*b+c moved a ray across an edge."""
import numpy as np
import pytest

from ipu_path_trace_amd import nif_assets
from oracle import pt_oracle as O


@pytest.fixture(scope="module")
def fast():
    try:
        return O.lib(fast=True)
    except Exception as e:   # noqa: BLE001 -- a host without AVX2 / F16C cannot build it
        pytest.skip("timing build unavailable on this host: %s" % e)


def test_build_info_names_the_flags(fast):
    info = fast.orc_build_info().decode()
    assert "-O3" in info and "-march=native" in info and "-fopenmp" in info
    assert O.lib().orc_build_info().decode().startswith("strict")


@pytest.mark.parametrize("kw", [dict(), dict(hidden=96, layer_count=4), dict(hidden=320, layer_count=6, dtype=np.float32),
                                dict(hidden=1024, layer_count=3)])
def test_nif_matches_the_strict_build(fast, kw):
    layers = nif_assets.synthetic_nif(**kw)
    meta, mean = nif_assets.URBAN_ALLEY_META, nif_assets.folded_mean()
    strict = O.Nif(layers, 12, meta["max"], mean)
    quick = O.Nif(layers, 12, meta["max"], mean, fast=True)
    rng = np.random.default_rng(3)
    n = 1000 + 37                                            # ragged against the 64-sample batch and the 6 / 12-row blocks
    u, v = rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32)
    a, b = strict.infer(u, v), quick.infer(u, v)
    # stated NIF tolerance of the GPU path is 2e-2 relative; the timing build differs from the strict one only by the
    # contraction of x * max + mean in the decode (one rounding instead of two in front of exp)
    np.testing.assert_allclose(b, a, rtol=2e-6, atol=1e-9)


def test_c1_render_matches_the_strict_build(fast):
    """BASELINE configs[0] at a reduced size: constant sky, depth 4."""
    W = H = 96
    cfg = O.make_config(width=W, height=H, max_path_length=4, env_mode=O.ENV_CONSTANT, env_rgb=(1.0, 1.0, 1.0))
    a, b = O.worklist(W, H), O.worklist(W, H)
    sa = O.render(cfg, None, a, 0, 16)
    sb = O.render(cfg, None, b, 0, 16, fast=True)
    assert sa.paths == sb.paths == W * H * 16
    same = a["pathLength"] == b["pathLength"]
    # a contracted a*b+c differs from the strict build in the last ulp; in front of a rounding to half (AA noise, camera ray)
    # or of a hit / miss decision that occasionally gives another path: < 0.1 % of the paths, i.e. < 2 % of the 16-sample sums
    assert same.mean() > 0.97
    assert abs(int(sa.segments) - int(sb.segments)) <= 0.001 * sa.segments
    for c in "rgb":
        close = np.isclose(b[c][same], a[c][same], rtol=1e-4, atol=1e-5)
        assert close.mean() > 0.99                           # same length but another path: rare
        assert abs(float(a[c].mean()) - float(b[c].mean())) < 2e-3 * float(a[c].mean())


def test_c2_shape_render_matches_the_strict_build(fast):
    layers = nif_assets.synthetic_nif()
    meta, mean = nif_assets.URBAN_ALLEY_META, nif_assets.folded_mean()
    cfg = O.make_config(width=1104, height=1000, max_path_length=8, env_mode=O.ENV_NIF)
    full = O.worklist(1104, 1000)
    pick = np.random.default_rng(0).choice(full.size, 3000, replace=False)
    a, b = full[pick].copy(), full[pick].copy()
    sa = O.render(cfg, O.Nif(layers, 12, meta["max"], mean), a, 0, 2)
    sb = O.render(cfg, O.Nif(layers, 12, meta["max"], mean, fast=True), b, 0, 2, fast=True)
    same = a["pathLength"] == b["pathLength"]
    assert same.mean() > 0.995 and abs(int(sa.escaped) - int(sb.escaped)) <= 0.002 * sa.escaped
    for c in "rgb":   # same paths -> same NIF inputs up to the last ulp of a contracted direction -> within the NIF tolerance
        close = np.isclose(b[c][same], a[c][same], rtol=2e-2, atol=1e-4)
        assert close.mean() > 0.995                          # (a pixel whose two paths differ but have equal lengths: rare)
        assert abs(float(a[c].mean()) - float(b[c].mean())) < 1e-2 * float(a[c].mean())


def test_the_avx2_block_of_the_timing_build(fast, tmp_path):
    """The timing build has two NIF blocks, chosen by the compiler's target: 12 x 16 on AVX-512, 6 x 16 on AVX2.  A host with
    AVX-512 (this one, the pool's boxes) only ever runs the first, so the second is built here with -mno-avx512f and checked
    the same way -- a CPU without AVX-512 would otherwise run an untested path in bench.py's cpu_baseline leg."""
    import ctypes as C
    import os
    import subprocess
    if "AVX-512" not in fast.orc_build_info().decode():
        pytest.skip("this host's own timing build already is the AVX2 one")
    here = os.path.dirname(os.path.abspath(O.__file__))
    so = str(tmp_path / "libpt_oracle_fast_avx2.so")
    subprocess.check_call(["gcc", "-O3", "-march=native", "-mno-avx512f", "-std=gnu11", "-fPIC", "-shared", "-fopenmp", "-fno-math-errno",
                           "-DORC_FAST_BUILD", "-o", so, os.path.join(here, "pt_oracle.c"), "-lm"])
    lib = O._bind(C.CDLL(so))
    assert "AVX2 6x16" in lib.orc_build_info().decode()
    layers = nif_assets.synthetic_nif(hidden=96, layer_count=4)
    meta, mean = nif_assets.URBAN_ALLEY_META, nif_assets.folded_mean()
    strict = O.Nif(layers, 12, meta["max"], mean)
    # a Nif bound to the AVX2 library by hand (O.Nif only knows the two builds of this host)
    keep, arr = [], (O.LayerAny * len(layers))()
    for i, (k, b, relu) in enumerate(layers):
        k = np.ascontiguousarray(k, dtype=np.float16); b = np.ascontiguousarray(b, dtype=np.float16)
        keep += [k, b]
        arr[i].rows, arr[i].cols = k.shape
        arr[i].kernel, arr[i].bias, arr[i].relu, arr[i].float32 = k.ctypes.data, b.ctypes.data, int(bool(relu)), 0
    m = (C.c_float * 3)(*[float(x) for x in mean])
    handle = lib.orc_nif_create_mixed(arr, len(layers), 12, float(meta["max"]), m, 1)
    rng = np.random.default_rng(3)
    n = 1000 + 37
    u, v = rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32)
    out = np.empty((n, 3), dtype=np.float32)
    assert lib.orc_nif_infer(handle, u.ctypes.data, v.ctypes.data, n, out.ctypes.data) == 0
    lib.orc_nif_destroy(handle)
    np.testing.assert_allclose(out, strict.infer(u, v), rtol=2e-6, atol=1e-9)
