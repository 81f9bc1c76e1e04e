"""NIF shapes beyond the shipped 6x320 / embedding 12 (reference: NifModel.cpp:295-326 builds whatever Dense stack
the H5 describes; Hdf5Model.cpp:109-133 accepts float16 and float32 variables).  Same stated tolerance as
test_gpu_parity.py: decoded radiance within 2e-2 relative of the oracle, median below 2e-3.

Parity of these shapes is UNPINNED by the reference (it ships one metadata file and no weights): the checker is
the oracle's independent fp32 restatement of the same rounding points.
"""
import numpy as np
import pytest

from ipu_path_trace_amd import nif_assets

pytestmark = pytest.mark.gpu

NIF_RTOL_MAX = 2e-2
NIF_RTOL_MEDIAN = 2e-3
META = nif_assets.URBAN_ALLEY_META


def _check(O, ptmi, L, emb, sizes=(1, 65, 3000), seed=5):
    mean = nif_assets.folded_mean()
    onif = O.Nif([(np.asarray(k, dtype=np.float16), None if b is None else np.asarray(b, dtype=np.float16), relu)
                  for k, b, relu in L], emb, META["max"], mean)
    r = ptmi.Renderer(64, 64)
    r.init_nif_weights(L, emb, META["max"], mean)
    rng = np.random.default_rng(seed)
    for n in sizes:
        u = rng.random(n, dtype=np.float32)
        v = rng.random(n, dtype=np.float32)
        got = r.nif_infer(u, v)
        ref = onif.infer(u, v)
        assert np.isfinite(got).all()
        rel = np.abs(got - ref) / np.abs(ref)
        assert rel.max() < NIF_RTOL_MAX, (n, rel.max())
        if n > 100:
            assert np.median(rel) < NIF_RTOL_MEDIAN
    r.close()


@pytest.mark.parametrize("emb", [4, 8, 12, 16])
def test_every_register_resident_width(oracle, ptmi_lib, emb):
    """hidden = every multiple of 32 up to 320 x embedding {4, 8, 12, 16}: one nif_kernel_v3 instantiation each."""
    for hidden in range(32, 321, 32):
        L = nif_assets.synthetic_nif(hidden=hidden, layer_count=3 + (hidden // 32) % 3, embedding_dim=emb, seed=hidden + emb)
        _check(oracle, ptmi_lib, L, emb)


@pytest.mark.parametrize("hidden,emb", [(96, 8), (160, 16), (288, 4), (224, 12)])
def test_deep_networks_take_the_v2_ring(oracle, ptmi_lib, hidden, emb):
    """More than 8 dense layers: nif_kernel_v2 (layer 0 and all biases resident)."""
    L = nif_assets.synthetic_nif(hidden=hidden, layer_count=10, embedding_dim=emb, seed=3 * hidden + emb)
    _check(oracle, ptmi_lib, L, emb)


@pytest.mark.parametrize("widths,emb,skips", [
    ([100, 60, 200], 10, {2}),        # ragged widths, embedding not a multiple of 4 -> zero-padded to 224 / 12
    ([320, 17, 320], 12, {1, 3}),     # a bottleneck, two concat layers, the head takes concat(x, input)
    ([48], 3, set()),                 # layer 0 + head only
    ([400, 300], 12, {1}),            # wider than 320 -> layer-by-layer path padded to 512
    ([768, 768, 768], 5, {2}),        # 768 = 3 feature blocks of 256
])
def test_arbitrary_dense_stacks(oracle, ptmi_lib, widths, emb, skips):
    L = nif_assets.synthetic_nif(widths=widths, embedding_dim=emb, skips=skips, seed=99 + len(widths))
    assert ptmi_lib  # shapes as the reference would see them
    _check(oracle, ptmi_lib, L, emb, sizes=(1, 65, 2000))


def test_bias_free_and_linear_layers(oracle, ptmi_lib):
    """use_bias=False layers (Hdf5Model.cpp:77-82) and 'linear' hidden activations (NifModel.cpp:73-76)."""
    L = nif_assets.synthetic_nif(hidden=128, layer_count=4, embedding_dim=8, seed=21)
    L = [(k, None if i in (1, 4) else b, relu and i != 2) for i, (k, b, relu) in enumerate(L)]
    _check(oracle, ptmi_lib, L, 8)


@pytest.mark.parametrize("widths,emb,skips", [
    ([320] * 6, 12, None),                        # the shipped architecture with float32 variables
    ([96, 72, 40], 10, {2, 3}),                   # ragged widths, embedding not a multiple of 4, concat on a hidden layer and the head
    ([512, 512], 4, {1}),                         # wider than the register-resident fp16 kernels take
])
def test_float32_models_run_in_float(oracle, ptmi_lib, widths, emb, skips):
    """A Keras H5 with float32 variables loads (Hdf5Model.cpp:109-133) and the reference then gives every matmul its kernel's
    type (NifModel.cpp:314): the layers run in float.  Rounds 1-2 rounded such weights to binary16; now a model all of whose
    layers are float32 takes the float path (pt_nif_f32.h: v_mfma_f32_32x32x2_f32, an exact fp32 FMA chain in k order), checked
    against the oracle's float mode -- the same sequential fmaf sums, so the only differences left are the half-precision
    trig of the features (v_sin against libm, then rounded to half) and the final exp.  It must NOT equal the fp16 result."""
    L32 = nif_assets.synthetic_nif(widths=widths, embedding_dim=emb, seed=5, dtype=np.float32, skips=skips)
    assert all(k.dtype == np.float32 for k, _, _ in L32)
    mean = nif_assets.folded_mean()
    onif = oracle.Nif(L32, emb, META["max"], mean)
    assert onif.float32
    rng = np.random.default_rng(1)
    r = ptmi_lib.Renderer(64, 64)
    r.init_nif_weights(L32, emb, META["max"], mean)
    for n in (1, 33, 5000):
        u = rng.random(n, dtype=np.float32)
        v = rng.random(n, dtype=np.float32)
        got, ref = r.nif_infer(u, v), onif.infer(u, v)
        assert np.isfinite(got).all()
        rel = np.abs(got - ref) / np.abs(ref)
        assert rel.max() < NIF_RTOL_MAX, (n, rel.max())           # a feature one half-ulp off moves the output by this much at most
        if n > 100:
            assert np.median(rel) < 2e-5, np.median(rel)          # ... and most samples agree to fp32 rounding
    L16 = [(k.astype(np.float16), b.astype(np.float16), relu) for k, b, relu in L32]
    r.init_nif_weights(L16, emb, META["max"], mean)
    half = r.nif_infer(u, v)
    r.close()
    assert np.median(np.abs(half - got) / np.abs(got)) > 1e-4     # the float path is not the fp16 path


def test_float32_model_spans_chunks(oracle, ptmi_lib):
    """More samples than one chunk of the float path holds (4096 queue tiles = 131,072): the second chunk reuses the packed
    activation buffers, its last workgroup is partly filled, and a 32 (mod 64) wide layer takes the one-tile block."""
    L32 = nif_assets.synthetic_nif(widths=[96, 64], embedding_dim=8, seed=12, dtype=np.float32)
    mean = nif_assets.folded_mean()
    onif = oracle.Nif(L32, 8, META["max"], mean)
    rng = np.random.default_rng(4)
    n = 131072 + 8 * 32 + 777
    u = rng.random(n, dtype=np.float32)
    v = rng.random(n, dtype=np.float32)
    r = ptmi_lib.Renderer(64, 64)
    r.init_nif_weights(L32, 8, META["max"], mean)
    got, ref = r.nif_infer(u, v), onif.infer(u, v)
    r.close()
    rel = np.abs(got - ref) / np.abs(ref)
    assert np.isfinite(got).all() and rel.max() < NIF_RTOL_MAX and np.median(rel) < 2e-5, (rel.max(), np.median(rel))


def test_float32_model_renders_and_a_mixed_model_runs_each_layer_in_its_type(oracle, ptmi_lib):
    """The float path inside the whole step (queue, chunks, scatter, accumulate) against the oracle's float mode; and a
    model that MIXES float32 and float16 layers runs every layer in the type of its own kernel (NifModel.cpp:314: a matmul
    takes its kernel's type): the float16 layers round their sums to half, add their bias in half and read their input cast
    to half, the float32 layers do none of that -- against the oracle's mixed mode (orc_nif_create_mixed), for every
    placement of the float32 layers, the head included."""
    O = oracle
    W = H = 48
    L32 = nif_assets.synthetic_nif(hidden=64, layer_count=3, seed=8, dtype=np.float32)
    mean = nif_assets.folded_mean()
    r = ptmi_lib.Renderer(W, H, max_path_length=5)
    r.init_nif_weights(L32, 12, META["max"], mean)
    r.init_render_settings(samples_per_step=3)
    rec = ptmi_lib.worklist(W, H)
    r.setup(rec)
    r.path_trace()
    st = r.read_results(rec)
    cfg = O.make_config(width=W, height=H, max_path_length=5, env_mode=O.ENV_NIF)
    ref = O.worklist(W, H)
    ost = O.render(cfg, O.Nif(L32, 12, META["max"], mean), ref, 0, 3)
    assert np.array_equal(rec["pathLength"], ref["pathLength"]) and st.escaped == ost.escaped
    assert st.nif_flops_per_sample == nif_assets.flops_per_sample(L32)
    for c in "rgb":
        np.testing.assert_allclose(rec[c], ref[c], rtol=NIF_RTOL_MAX, atol=1e-6)

    L16 = [(k.astype(np.float16), b.astype(np.float16), relu) for k, b, relu in L32]
    rng = np.random.default_rng(9)
    u, v = rng.random(3000, dtype=np.float32), rng.random(3000, dtype=np.float32)
    r.init_nif_weights(L16, 12, META["max"], mean)
    all_half = r.nif_infer(u, v)
    n = len(L32)
    for f32_layers in ([1], [0], [n - 1], [0, n - 1], [1, 2], list(range(n - 1))):   # which layers keep float32 variables
        mixed = [L32[i] if i in f32_layers else L16[i] for i in range(n)]
        onif = O.Nif(mixed, 12, META["max"], mean)
        assert onif.mixed and not onif.float32
        r.init_nif_weights(mixed, 12, META["max"], mean)
        assert "nif32_layer_kernel" in (r.nif_infer(u[:1], v[:1]), r.nif_kernel_name())[1]   # a mixed model takes the float path
        got, want = r.nif_infer(u, v), onif.infer(u, v)
        assert np.isfinite(got).all()
        rel = np.abs(got - want) / np.abs(want)
        # every layer is the same k-ordered fp32 FMA chain on both sides; what differs is the half-precision trig of the
        # features (v_sin against libm), which a half rounding downstream can amplify to one half-ulp of a hidden activation
        assert rel.max() < NIF_RTOL_MAX and np.median(rel) < 2e-4, (f32_layers, rel.max(), np.median(rel))
        # ... and it is NOT the all-float16 result (what rounds 1-3 computed for such a model)
        assert np.median(np.abs(got - all_half) / np.abs(all_half)) > 2e-5, f32_layers
    r.close()


def test_shapes_the_reference_would_reject(ptmi_lib):
    r = ptmi_lib.Renderer(32, 32)
    good = nif_assets.synthetic_nif(hidden=64, layer_count=2)
    for bad in (
        good[:-1] + [(np.zeros((64, 4), np.float16), None, False)],            # head is not 3-wide
        [(np.zeros((40, 64), np.float16), None, True)] + good[1:],             # first layer does not take 4*E features
        good[:1] + [(np.zeros((70, 64), np.float16), None, True)] + good[2:],  # neither width nor width + 4*E
        good[:1],                                                              # a single layer
    ):
        with pytest.raises(ptmi_lib.PtError) as e:
            r.init_nif_weights(bad, 12, 1.0, [0, 0, 0])
        assert e.value.code == -4   # PT_ERR_UNSUPPORTED_MODEL
    with pytest.raises(ptmi_lib.PtError):
        r.init_nif_weights(good, 17, 1.0, [0, 0, 0])                            # embedding > 16
    r.close()
