"""GPU parity tests proper: HIP kernels (through the C-ABI) against the CPU oracle.

Trace stage: BIT-EXACT (integer path structure and every float of the path record / TraceRecord).
NIF stage: tolerance -- fp16 MFMA accumulation order differs from the oracle's sequential fp32 sum,
so a hidden activation can round to the neighbouring half.  Stated bound: decoded radiance within
2e-2 relative of the oracle, median below 2e-3.
"""
import numpy as np
import pytest

from ipu_path_trace_amd import nif_assets

pytestmark = pytest.mark.gpu

NIF_RTOL_MAX = 2e-2
NIF_RTOL_MEDIAN = 2e-3


def _oracle_paths(O, cfg, u, v, s):
    out = []
    for a, b, c in zip(u, v, s):
        out.append(O.trace_path(cfg, int(a), int(b), int(c)))
    return out


@pytest.mark.parametrize("depth,aa,prec,rot", [(4, 0, 0, 0.0), (8, 0, 0, 37.0), (16, 1, 1, 0.0), (10, 2, 0, 300.0)])
def test_trace_paths_bit_exact(oracle, ptmi_lib, depth, aa, prec, rot):
    O = oracle
    W, H = 1104, 1000
    rng = np.random.default_rng(1234 + depth)
    n = 6000
    u = rng.integers(0, W, n).astype(np.uint16)
    v = rng.integers(0, H, n).astype(np.uint16)
    # bias half of the pixels towards the spheres/floor where paths are long
    v[: n // 2] = rng.integers(H // 2, H, n // 2).astype(np.uint16)
    u[:8] = 65535  # worklist padding coordinates (LoadBalancer.cpp:66-71) are traced like real ones
    v[:8] = 65535
    s = rng.integers(0, 100000, n).astype(np.uint32)
    cfg = O.make_config(width=W, height=H, max_path_length=depth, aa_noise_type=aa, sample_precision=prec, seed=77,
                        env_rotation_degrees=rot)
    r = ptmi_lib.Renderer(W, H, max_work_items=1024, max_path_length=depth, aa_noise_type=aa, sample_precision=prec)
    r.set_constant_env((1, 1, 1))
    r.init_render_settings(seed=77, env_rotation_degrees=rot, samples_per_step=1)
    got = r.trace_paths(u, v, s)
    ref = _oracle_paths(O, cfg, u, v, s)
    lengths = np.array([p.length for p in ref], dtype=np.uint32)
    esc = np.array([p.escaped for p in ref], dtype=np.uint32)
    assert np.array_equal(got["length"], lengths)
    assert np.array_equal(got["escaped"], esc)
    for name in ("dir", "uv", "throughput", "cam"):
        refv = np.array([list(getattr(p, name)) for p in ref], dtype=np.float32)
        assert np.array_equal(got[name].view(np.uint32), refv.view(np.uint32)), name
    assert lengths.max() > 3 and esc.sum() > 0 and (esc == 0).sum() > 0
    r.close()


def test_render_constant_sky_bit_exact_config_c1(oracle, ptmi_lib):
    """BASELINE config C1: 256x256, 16 spp, depth 4, constant sky.  Every TraceRecord field is identical."""
    O = oracle
    W = H = 256
    cfg = O.make_config(width=W, height=H, max_path_length=4, env_rgb=(1.0, 0.9, 0.8), fold=O.FOLD_FORWARD)
    ref = O.worklist(W, H)
    st = O.render(cfg, None, ref, 0, 16)
    r = ptmi_lib.Renderer(W, H, max_path_length=4, iterations_per_batch=5)  # 16 = 5+5+5+1: exercises a ragged last batch
    r.set_constant_env((1.0, 0.9, 0.8))
    r.init_render_settings(seed=1, samples_per_step=16)
    got = ptmi_lib.worklist(W, H)
    r.setup(got)
    r.path_trace()
    gst = r.read_results(got)
    assert got.tobytes() == ref.tobytes()
    assert (gst.paths, gst.segments, gst.escaped) == (st.paths, st.segments, st.escaped)
    # a second step continues the sample sequence (samples 16..31) and keeps accumulating
    O.render(cfg, None, ref, 16, 16)
    r.path_trace()
    r.read_results(got)
    assert got.tobytes() == ref.tobytes()
    r.close()


@pytest.mark.parametrize("opts", [
    dict(max_path_length=1),                                                    # every path ends at its first hit or escapes
    dict(max_path_length=3, roulette_depth=1, stop_prob=0.6),                   # roulette from the first bounce on, most paths stop
    dict(max_path_length=12, roulette_depth=20, stop_prob=0.3),                 # roulette never starts: paths run to the depth limit
    dict(max_path_length=8, refractive_index=1.33, stop_prob=0.0),              # water instead of glass, roulette that never stops
    dict(max_path_length=6, aa_noise_type=1, aa_noise_scale=1.5, fov_degrees=40.0, env_rotation_degrees=123.0, seed=987654321987),
])
def test_render_bit_exact_under_non_default_options(oracle, ptmi_lib, opts):
    """The CLI options that shape the sampling loop (PathTracerApp.cpp:797-817: --max-path-length, --roulette-depth,
    --stop-prob, --refractive-index, --aa-noise-type/-scale, --fov, --env-map-rotation, --seed) away from their defaults:
    the three-phase trace kernel (camera rays / first shading / persistent loop) against the oracle, every TraceRecord
    field identical, on an image whose width is not a multiple of the wave size and over a ragged last batch."""
    O = oracle
    W, H, spp = 203, 77, 7
    create = {k: opts[k] for k in ("max_path_length", "roulette_depth", "stop_prob", "refractive_index", "aa_noise_type") if k in opts}
    settings = {k: opts[k] for k in ("seed", "aa_noise_scale", "fov_degrees", "env_rotation_degrees") if k in opts}
    cfg = O.make_config(width=W, height=H, env_rgb=(0.7, 1.1, 0.4), **opts)
    ref = O.worklist(W, H)
    st = O.render(cfg, None, ref, 0, spp)
    r = ptmi_lib.Renderer(W, H, iterations_per_batch=3, **create)
    r.set_constant_env((0.7, 1.1, 0.4))
    r.init_render_settings(samples_per_step=spp, **settings)
    got = ptmi_lib.worklist(W, H)
    r.setup(got)
    r.path_trace()
    gst = r.read_results(got)
    r.close()
    assert (gst.paths, gst.segments, gst.escaped) == (st.paths, st.segments, st.escaped)
    assert got.tobytes() == ref.tobytes()


EXTREME_OPTIONS = [
    dict(W=64, H=48, max_path_length=64, roulette_depth=64, stop_prob=0.05),             # the longest stack the ABI takes, no roulette
    dict(W=64, H=48, max_path_length=64, roulette_depth=1, stop_prob=0.99),              # roulette weight 1 / (1 - 0.99) on every bounce
    dict(W=97, H=31, max_path_length=8, fov_degrees=179.0),                              # tan(fov / 2) = 115: almost every ray leaves sideways
    dict(W=97, H=31, max_path_length=8, fov_degrees=1.0),                                # a pencil of rays at the image centre
    dict(W=80, H=60, max_path_length=10, refractive_index=1.0),                          # glass that does not bend (r0 = 0)
    dict(W=80, H=60, max_path_length=10, refractive_index=3.0),                          # total internal reflection almost everywhere
    dict(W=80, H=60, max_path_length=6, aa_noise_type=2, aa_noise_scale=3.0),            # truncated normal, 3 pixels wide
    dict(W=80, H=60, max_path_length=6, aa_noise_scale=0.0, sample_precision=1),         # no jitter; 24-bit primary samples
    dict(W=80, H=60, max_path_length=6, env_rotation_degrees=-725.0, seed=0),            # azimuth far outside one turn, seed 0
    dict(W=80, H=60, max_path_length=6, seed=2 ** 64 - 1),                               # all seed bits set
    dict(W=3, H=997, max_path_length=5),                                                 # a sliver of an image
    dict(W=1, H=1, max_path_length=5),                                                   # one pixel
]


@pytest.mark.parametrize("opts", EXTREME_OPTIONS, ids=lambda o: ",".join("%s=%s" % kv for kv in o.items() if kv[0] not in ("W", "H")) or "tiny")
def test_render_bit_exact_at_the_extremes_of_the_options(oracle, ptmi_lib, opts):
    """The same comparison at the ENDS of the option ranges the boundary accepts (pt_create / pt_set_render_settings: depth 1..64,
    stop-prob in [0, 1), fov in (0, 180) degrees, any refractive index, any seed, any image from 1 x 1): every TraceRecord field
    identical to the oracle's.  (Reference: the same options of PathTracerApp.cpp:797-817; it has no range checks of its own.)"""
    O = oracle
    opts = dict(opts)
    W, H, spp = opts.pop("W"), opts.pop("H"), 5
    create = {k: opts[k] for k in ("max_path_length", "roulette_depth", "stop_prob", "refractive_index", "aa_noise_type", "sample_precision") if k in opts}
    settings = {k: opts[k] for k in ("seed", "aa_noise_scale", "fov_degrees", "env_rotation_degrees") if k in opts}
    cfg = O.make_config(width=W, height=H, env_rgb=(0.9, 0.8, 1.2), **opts)
    ref = O.worklist(W, H)
    st = O.render(cfg, None, ref, 3, spp)                     # (from sample index 3: the cursor below is advanced to it)
    r = ptmi_lib.Renderer(W, H, iterations_per_batch=2, **create)
    r.set_constant_env((0.9, 0.8, 1.2))
    r.init_render_settings(samples_per_step=3, **settings)
    got = ptmi_lib.worklist(W, H)
    r.setup(got)
    r.path_trace()                                            # samples 0..2, thrown away: setup again keeps the sample cursor
    got = ptmi_lib.worklist(W, H)
    r.setup(got)
    r.init_render_settings(samples_per_step=spp, **settings)  # same seed: the sequence continues at sample 3
    r.path_trace()
    gst = r.read_results(got)
    r.close()
    assert gst.first_sample == 3
    assert (gst.paths, gst.segments, gst.escaped) == (st.paths, st.segments, st.escaped)
    assert got.tobytes() == ref.tobytes()


def test_render_backward_fold_matches_forward(oracle, ptmi_lib):
    """GPU (forward throughput) against the reference's backward fold (codelets.cpp:255-292): rounding only."""
    O = oracle
    W = H = 128
    cfg = O.make_config(width=W, height=H, max_path_length=8, fold=O.FOLD_BACKWARD)
    ref = O.worklist(W, H)
    O.render(cfg, None, ref, 0, 8)
    r = ptmi_lib.Renderer(W, H, max_path_length=8)
    r.set_constant_env((1, 1, 1))
    r.init_render_settings(samples_per_step=8)
    got = ptmi_lib.worklist(W, H)
    r.setup(got)
    r.path_trace()
    r.read_results(got)
    assert np.array_equal(got["pathLength"], ref["pathLength"])
    for c in "rgb":
        np.testing.assert_allclose(got[c], ref[c], rtol=2e-6, atol=1e-7)
    r.close()


@pytest.mark.parametrize("hidden,layers", [(64, 2), (128, 4), (320, 6), (64, 10), (256, 9), (512, 3), (1024, 8)])
def test_nif_infer_matches_oracle(oracle, ptmi_lib, hidden, layers):
    O = oracle
    L = nif_assets.synthetic_nif(hidden=hidden, layer_count=layers, seed=7 + hidden)
    meta = nif_assets.URBAN_ALLEY_META
    mean = nif_assets.folded_mean()
    onif = O.Nif(L, 12, meta["max"], mean)
    r = ptmi_lib.Renderer(64, 64)
    r.init_nif_weights(L, 12, meta["max"], mean)
    rng = np.random.default_rng(5)
    for n in (1, 63, 64, 65, 1000, 50001 if hidden <= 320 else 9001):  # ragged tiles
        u = rng.random(n, dtype=np.float32)
        v = rng.random(n, dtype=np.float32)
        got = r.nif_infer(u, v)
        ref = onif.infer(u, v)
        rel = np.abs(got - ref) / np.abs(ref)
        assert rel.max() < NIF_RTOL_MAX, (n, rel.max())
        assert np.median(rel) < NIF_RTOL_MEDIAN
    r.close()


def test_render_with_nif_matches_oracle(oracle, ptmi_lib):
    O = oracle
    W = H = 96
    L = nif_assets.synthetic_nif()
    meta = nif_assets.URBAN_ALLEY_META
    mean = nif_assets.folded_mean()
    onif = O.Nif(L, 12, meta["max"], mean)
    cfg = O.make_config(width=W, height=H, max_path_length=8, env_mode=O.ENV_NIF, env_rotation_degrees=20.0)
    ref = O.worklist(W, H)
    st = O.render(cfg, onif, ref, 0, 6)
    r = ptmi_lib.Renderer(W, H, max_path_length=8, iterations_per_batch=4)
    r.init_nif_weights(L, 12, meta["max"], mean)
    r.init_render_settings(env_rotation_degrees=20.0, samples_per_step=6)
    got = ptmi_lib.worklist(W, H)
    r.setup(got)
    r.path_trace()
    gst = r.read_results(got)
    assert np.array_equal(got["pathLength"], ref["pathLength"])
    assert np.array_equal(got["sampleCount"], ref["sampleCount"])
    assert (gst.paths, gst.segments, gst.escaped) == (st.paths, st.segments, st.escaped)
    assert gst.nif_flops_per_sample == 1089283
    for c in "rgb":
        np.testing.assert_allclose(got[c], ref[c], rtol=NIF_RTOL_MAX, atol=1e-6)
    r.close()


def test_wide_nif_spans_chunks(oracle, ptmi_lib):
    """Layer-by-layer path (pt_nif_gemm.h): a queue longer than one chunk of 4096 tiles, with a ragged last sample
    block, must give the same numbers as the oracle for every sample."""
    O = oracle
    L = nif_assets.synthetic_nif(hidden=512, layer_count=3, seed=77)
    meta = nif_assets.URBAN_ALLEY_META
    mean = nif_assets.folded_mean()
    onif = O.Nif(L, 12, meta["max"], mean)
    r = ptmi_lib.Renderer(64, 64)
    r.init_nif_weights(L, 12, meta["max"], mean)
    rng = np.random.default_rng(11)
    n = 4096 * 32 + 12345
    u = rng.random(n, dtype=np.float32)
    v = rng.random(n, dtype=np.float32)
    got = r.nif_infer(u, v)
    ref = onif.infer(u, v)
    rel = np.abs(got - ref) / np.abs(ref)
    assert rel.max() < NIF_RTOL_MAX, rel.max()
    assert np.median(rel) < NIF_RTOL_MEDIAN
    r.close()


def test_render_with_wide_nif_matches_oracle(oracle, ptmi_lib):
    """Whole step through the layer-by-layer NIF path: many queue regions with ragged counts."""
    O = oracle
    W = H = 96
    L = nif_assets.synthetic_nif(hidden=512, layer_count=3, seed=78)
    meta = nif_assets.URBAN_ALLEY_META
    mean = nif_assets.folded_mean()
    onif = O.Nif(L, 12, meta["max"], mean)
    cfg = O.make_config(width=W, height=H, max_path_length=8, env_mode=O.ENV_NIF, env_rotation_degrees=-35.0)
    ref = O.worklist(W, H)
    st = O.render(cfg, onif, ref, 0, 5)
    r = ptmi_lib.Renderer(W, H, max_path_length=8, iterations_per_batch=2)
    r.init_nif_weights(L, 12, meta["max"], mean)
    r.init_render_settings(env_rotation_degrees=-35.0, samples_per_step=5)
    got = ptmi_lib.worklist(W, H)
    r.setup(got)
    r.path_trace()
    gst = r.read_results(got)
    assert np.array_equal(got["pathLength"], ref["pathLength"])
    assert (gst.paths, gst.segments, gst.escaped) == (st.paths, st.segments, st.escaped)
    for c in "rgb":
        np.testing.assert_allclose(got[c], ref[c], rtol=NIF_RTOL_MAX, atol=1e-6)
    r.close()


def test_errors_are_reported(ptmi_lib):
    r = ptmi_lib.Renderer(32, 32)
    with pytest.raises(ptmi_lib.PtError):
        r.path_trace()  # no render settings yet
    r.init_render_settings(samples_per_step=1)
    with pytest.raises(ptmi_lib.PtError):
        r.path_trace()  # no environment
    bad = nif_assets.synthetic_nif(hidden=96, layer_count=2)[:-1]
    with pytest.raises(ptmi_lib.PtError):
        r.init_nif_weights(bad, 12, 1.0, [0, 0, 0])  # no 3-channel head
    with pytest.raises(ptmi_lib.PtError) as e:
        r.init_render_settings(samples_per_step=65536)   # TraceRecord::sampleCount is uint16 (TraceRecord.hpp:10)
    assert e.value.code == -1
    r.close()
