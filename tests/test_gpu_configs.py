"""BASELINE.json configurations at full size on the GPU, checked through size-independent properties and through
the oracle on a random subset of pixels (the oracle cannot render the full sizes in test time)."""
import os

import numpy as np
import pytest

from ipu_path_trace_amd import nif_assets, partition

pytestmark = pytest.mark.gpu

NIF_RTOL = 2e-2   # stated NIF tolerance


def _subset_against_oracle(O, ptmi, r, W, H, depth, spp, layers, n_pixels, seed, rotation=0.0):
    """Render `n_pixels` random pixels with the GPU (as their own small worklist) and with the oracle."""
    rng = np.random.default_rng(seed)
    rec = np.zeros(n_pixels, dtype=ptmi.TRACE_DTYPE)
    rec["u"] = rng.integers(0, W, n_pixels)
    rec["v"] = rng.integers(H // 3, H, n_pixels)
    ref = rec.copy()
    r.setup(rec)
    r.path_trace()
    r.read_results(rec)
    meta = nif_assets.URBAN_ALLEY_META
    cfg = O.make_config(width=W, height=H, max_path_length=depth, env_mode=O.ENV_NIF, env_rotation_degrees=rotation)
    O.render(cfg, O.Nif(layers, 12, meta["max"], nif_assets.folded_mean()), ref, 0, spp)
    return rec, ref


def test_config_c2_full_size_properties_and_subset(oracle, ptmi_lib):
    """C2: 1104x1000, 6x320 NIF, depth 8; one 24-spp step over the full image + oracle on 1500 pixels."""
    W, H, depth, spp = 1104, 1000, 8, 24
    layers = nif_assets.synthetic_nif()
    meta = nif_assets.URBAN_ALLEY_META
    r = ptmi_lib.Renderer(W, H, max_path_length=depth)
    r.init_nif_weights(layers, 12, meta["max"], nif_assets.folded_mean())
    r.init_render_settings(samples_per_step=spp)
    work = partition.tile_order_worklist(W, H)
    r.setup(work)
    r.path_trace()
    st = r.read_results(work)
    assert st.paths == W * H * spp and st.nif_flops_per_sample == 1089283
    assert np.all(work["sampleCount"] == spp)
    assert work["pathLength"].min() >= spp and work["pathLength"].max() <= depth * spp
    assert int(work["pathLength"].astype(np.int64).sum()) == st.segments
    assert 0.90 < st.escaped / st.paths < 0.97                       # the scene is mostly sky
    img = np.stack([work["r"], work["g"], work["b"]], -1)
    assert np.all(np.isfinite(img)) and img.min() >= 0
    # the image does not depend on worklist order: row-major worklist gives identical per-pixel records
    work2 = ptmi_lib.worklist(W, H)
    r.init_render_settings(seed=2, samples_per_step=spp)              # new seed restarts the sample sequence ...
    r.init_render_settings(seed=1, samples_per_step=spp)              # ... and so does switching back
    r.setup(work2)
    r.path_trace()
    r.read_results(work2)
    key = work["v"].astype(np.int64) * W + work["u"]
    assert work2[key].tobytes() == work.tobytes()
    # subset against the oracle (fresh sample sequence)
    r.init_render_settings(seed=3, samples_per_step=spp)
    r.init_render_settings(seed=1, samples_per_step=spp)
    got, ref = _subset_against_oracle(oracle, ptmi_lib, r, W, H, depth, spp, layers, 1500, seed=11)
    assert np.array_equal(got["pathLength"], ref["pathLength"])
    for c in "rgb":
        np.testing.assert_allclose(got[c], ref[c], rtol=NIF_RTOL, atol=1e-6)
    r.close()


def test_config_c3_4k_deep_paths(oracle, ptmi_lib):
    """C3: 3840x2160, depth 16 (deep-path divergence stress).  Full-size step with constant sky checked exactly on the
    sky rows; NIF subset against the oracle at depth 16; uint16 accumulators hold 1000 spp x 16."""
    W, H, depth = 3840, 2160, 16
    layers = nif_assets.synthetic_nif()
    meta = nif_assets.URBAN_ALLEY_META
    r = ptmi_lib.Renderer(W, H, max_path_length=depth)
    r.set_constant_env((0.5, 1.0, 2.0))
    spp = 6
    r.init_render_settings(samples_per_step=spp)
    work = partition.tile_order_worklist(W, H)
    r.setup(work)
    r.path_trace()
    st = r.read_results(work)
    assert st.paths == W * H * spp
    assert np.all(work["sampleCount"] == spp) and work["pathLength"].max() <= depth * spp
    assert work["pathLength"].max() > 8 * 1                              # deep paths occur
    sky = work[work["v"] < H // 3]                                     # camera looks along -z: upper third is all sky
    assert np.all(sky["pathLength"] == spp)
    assert np.all(sky["r"] == spp * 0.5) and np.all(sky["g"] == spp * 1.0) and np.all(sky["b"] == spp * 2.0)
    # exactness against the oracle on a subset at depth 16 (constant env: bit-exact)
    cfg = oracle.make_config(width=W, height=H, max_path_length=depth, env_rgb=(0.5, 1.0, 2.0))
    sel = np.random.default_rng(5).choice(work.size, 4000, replace=False)
    ref = np.zeros(sel.size, dtype=ptmi_lib.TRACE_DTYPE)
    ref["u"], ref["v"] = work["u"][sel], work["v"][sel]
    oracle.render(cfg, None, ref, 0, spp)
    assert work[sel].tobytes() == ref.tobytes()
    # 1000 spp x depth 16 fits the uint16 fields: run 1000 samples on a small worklist
    small = np.zeros(4096, dtype=ptmi_lib.TRACE_DTYPE)
    small["u"] = np.arange(4096) % W
    small["v"] = H - 1 - (np.arange(4096) // 64)
    r.init_render_settings(seed=9, samples_per_step=1000)
    r.setup(small)
    r.path_trace()
    r.read_results(small)
    assert np.all(small["sampleCount"] == 1000) and small["pathLength"].max() <= 16000
    # NIF at depth 16 against the oracle
    r.init_nif_weights(layers, 12, meta["max"], nif_assets.folded_mean())
    r.init_render_settings(seed=1, samples_per_step=8)
    got, ref = _subset_against_oracle(oracle, ptmi_lib, r, W, H, depth, 8, layers, 1200, seed=12)
    assert np.array_equal(got["pathLength"], ref["pathLength"])
    for c in "rgb":
        np.testing.assert_allclose(got[c], ref[c], rtol=NIF_RTOL, atol=1e-6)
    r.close()


def test_config_c4_partition_invariance(ptmi_lib):
    """C4 shards the 1104x1000 image over 8 ranks: each rank's tiles (its real share: 16x16 tiles dealt round-robin),
    rendered separately on this GPU and handed over through pt_gather_hdr, reassemble to the single-GPU image bit for bit
    (RNG keyed by pixel; no data-path collective).  Full image size, reduced sample count; the 8-GPU run itself needs an
    8-GPU box."""
    W, H, depth, spp, world = 1104, 1000, 8, 6, 8
    layers = nif_assets.synthetic_nif()
    meta = nif_assets.URBAN_ALLEY_META
    whole = partition.tile_order_worklist(W, H)
    r = ptmi_lib.Renderer(W, H, max_path_length=depth)
    r.init_nif_weights(layers, 12, meta["max"], nif_assets.folded_mean())
    r.init_render_settings(samples_per_step=spp)
    r.setup(whole)
    r.path_trace()
    r.read_results(whole)
    film = np.zeros((H, W, 3), np.float32)
    film[whole["v"], whole["u"]] = np.stack([whole["b"], whole["g"], whole["r"]], -1) * (np.float32(1.0) / np.float32(spp))
    parts = []
    for rank in range(world):
        rec = partition.tile_order_worklist(W, H, rank, world)
        r.init_render_settings(seed=5, samples_per_step=spp)
        r.init_render_settings(seed=1, samples_per_step=spp)          # restart the sample sequence per "rank"
        r.setup(rec)
        r.path_trace()
        slot = partition.max_items_per_rank(W, H, world)
        parts.append(r.gather_hdr(slot)[0])                            # the product's hand-off (degenerate at one rank)
        r.read_results(rec)
        np.testing.assert_array_equal(parts[-1][: rec.size], np.stack([rec["b"], rec["g"], rec["r"]], -1) * (np.float32(1.0) / np.float32(spp)))
    assert np.array_equal(partition.assemble_hdr(W, H, world, parts), film)
    r.close()


def test_config_c5_wide_nif_8x1024(oracle, ptmi_lib):
    """C5: NIF 8x1024 fp16 (14,891,011 FLOP per evaluation) at 1104x1000, depth 8: full-size step properties and a
    pixel subset against the oracle."""
    W, H, depth, spp = 1104, 1000, 8, 2
    layers = nif_assets.synthetic_nif(hidden=1024, layer_count=8, seed=31)
    meta = nif_assets.URBAN_ALLEY_META
    r = ptmi_lib.Renderer(W, H, max_path_length=depth)
    r.init_nif_weights(layers, 12, meta["max"], nif_assets.folded_mean())
    r.init_render_settings(samples_per_step=spp)
    work = partition.tile_order_worklist(W, H)
    r.setup(work)
    r.path_trace()
    st = r.read_results(work)
    assert st.nif_flops_per_sample == 14891011 and st.paths == W * H * spp
    assert np.all(work["sampleCount"] == spp)
    img = np.stack([work["r"], work["g"], work["b"]], -1)
    assert np.all(np.isfinite(img)) and img.min() >= 0
    r.init_render_settings(seed=4, samples_per_step=spp)
    r.init_render_settings(seed=1, samples_per_step=spp)
    rng = np.random.default_rng(21)
    rec = np.zeros(600, dtype=ptmi_lib.TRACE_DTYPE)
    rec["u"] = rng.integers(0, W, rec.size)
    rec["v"] = rng.integers(H // 3, H, rec.size)
    ref = rec.copy()
    r.setup(rec)
    r.path_trace()
    r.read_results(rec)
    cfg = oracle.make_config(width=W, height=H, max_path_length=depth, env_mode=oracle.ENV_NIF)
    oracle.render(cfg, oracle.Nif(layers, 12, meta["max"], nif_assets.folded_mean()), ref, 0, spp)
    assert np.array_equal(rec["pathLength"], ref["pathLength"])
    for c in "rgb":
        np.testing.assert_allclose(rec[c], ref[c], rtol=NIF_RTOL, atol=1e-6)
    r.close()


@pytest.mark.parametrize("dtype,hidden,nlayers", [(np.float16, 512, 3), (np.float32, 128, 3)])
def test_chunks_in_flight_on_two_streams_change_nothing(ptmi_lib, dtype, hidden, nlayers):
    """The layer-by-layer NIF paths (wide fp16: pt_nif_gemm.h; float32: pt_nif_f32.h) run the chunks of a queue alternately on
    two streams with a buffer set each.  A race between the two (a shared buffer, a missing join before the accumulate)
    would show as a film that differs from the one-stream film or from run to run: three renders of a queue of five chunks
    -- two streams, two streams again in a fresh handle, one stream (profiling build, PTMI_CHUNK_STREAMS=1) -- must be
    bit-identical in every TraceRecord field."""
    W, H, spp = 448, 400, 4                       # 716,800 paths in one batch: ~5 chunks of 131,072 escaped samples
    layers = nif_assets.synthetic_nif(hidden=hidden, layer_count=nlayers, seed=17, dtype=dtype)
    mean = nif_assets.folded_mean()

    def film(diag, streams):
        if streams:
            os.environ["PTMI_CHUNK_STREAMS"] = streams
        try:
            r = ptmi_lib.Renderer(W, H, max_path_length=6, iterations_per_batch=spp, diag=diag)
            r.init_nif_weights(layers, 12, nif_assets.URBAN_ALLEY_META["max"], mean)
            r.init_render_settings(samples_per_step=spp)
            work = ptmi_lib.worklist(W, H)
            r.setup(work)
            r.path_trace()
            r.path_trace()                         # a second step over the same buffers
            st = r.read_results(work)
            r.close()
        finally:
            os.environ.pop("PTMI_CHUNK_STREAMS", None)
        assert st.escaped > 4 * 131072 // 2        # per step of 4 iterations in one batch: more than two chunks' worth
        return work

    a, b = film(False, None), film(False, None)
    one = film(True, "1")
    assert a.tobytes() == b.tobytes()
    assert a.tobytes() == one.tobytes()
    assert np.all(np.isfinite(a["r"])) and a["r"].max() > 0


def test_redeal_by_measured_path_length_keeps_the_film(ptmi_lib):
    """SURVEY.md row N3 for the one-process-per-GPU layout, on real data: one interval with the static round-robin deal,
    tile costs from the path lengths read_results returns, partition.deal_by_path_length, a second interval under the
    new deal.  Each of the 4 "ranks" is rendered in turn on this GPU.  The film of the second interval equals, bit for
    bit, the film the static deal would have produced for the same sample indices, and the new deal is better balanced
    (reference: LoadBalancer::allocateWorkByPathLength, LoadBalancer.cpp:141-192)."""
    from ipu_path_trace_amd import partition
    W, H, world, spp = 320, 208, 4, 6
    L = nif_assets.synthetic_nif(hidden=64, layer_count=2)
    mean = nif_assets.folded_mean()

    def interval(owner, first_sample_step):
        """Render `spp` samples (sample indices first_sample_step*spp ...) of every rank's tiles; returns film, costs."""
        tiles, cost = [], np.zeros(len(owner))
        for rank in range(world):
            work = partition.worklist_for_owner(W, H, owner, rank)
            r = ptmi_lib.Renderer(W, H, max_work_items=partition.max_items_per_rank(W, H, world), max_path_length=8)
            r.init_nif_weights(L, 12, nif_assets.URBAN_ALLEY_META["max"], mean)
            r.init_render_settings(samples_per_step=spp)
            r.tile_costs_enable(partition.TILE, partition.TILE)
            r.setup(work)
            for _ in range(first_sample_step):            # advance the sample sequence to the interval's first index
                r.path_trace()
                r.clear_accumulators()
            r.path_trace()
            tiles.append(r.gather_hdr(partition.max_items_per_rank(W, H, world))[0])
            on_device = r.tile_costs(W, H)                # N3 without the worklist leaving the device: 8 B per tile
            r.read_results(work)
            np.testing.assert_array_equal(on_device, partition.tile_costs(work, W, H).astype(np.uint64))
            assert on_device.nbytes * 50 < work.nbytes
            cost += on_device
            # what pt_film_accumulate clears from the accumulators is folded into the per-tile sums first
            r.film_accumulate()
            r.read_results(work)
            assert not work["pathLength"].any()
            np.testing.assert_array_equal(r.tile_costs(W, H), on_device)
            r.close()
        return partition.assemble_hdr(W, H, world, tiles, owner=owner), cost

    static = partition.round_robin_owner(W, H, world)
    _, cost = interval(static, 0)
    dealt = partition.deal_by_path_length(cost, world)
    assert not np.array_equal(dealt, static)
    film_static, cost2 = interval(static, 1)
    film_dealt, cost2b = interval(dealt, 1)
    assert film_static.tobytes() == film_dealt.tobytes()
    np.testing.assert_array_equal(cost2, cost2b)          # path lengths are a property of (pixel, sample), not of the deal

    def imbalance(owner):
        per_rank = np.bincount(owner, weights=cost2, minlength=world)
        return per_rank.max() / per_rank.mean()
    assert imbalance(dealt) <= imbalance(static) + 1e-9
    assert np.bincount(dealt, minlength=world).max() - np.bincount(dealt, minlength=world).min() <= 1


def _same_render_subset(O, rec, n_pixels, W, H, depth, spp, layers, seed, min_row=0):
    """Records of `n_pixels` pixels picked FROM a finished full-size render, and the oracle's records for the same pixels."""
    rng = np.random.default_rng(seed)
    idx = rng.choice(np.flatnonzero(rec["v"] >= min_row), n_pixels, replace=False)
    ref = np.zeros(n_pixels, dtype=rec.dtype)
    ref["u"], ref["v"] = rec["u"][idx], rec["v"][idx]
    meta = nif_assets.URBAN_ALLEY_META
    cfg = O.make_config(width=W, height=H, max_path_length=depth, env_mode=O.ENV_NIF)
    O.render(cfg, O.Nif(layers, 12, meta["max"], nif_assets.folded_mean()), ref, 0, spp)
    return rec[idx], ref


@pytest.mark.parametrize("name,W,H,depth,spp,hidden,nlayers,n_check", [
    ("C2", 1104, 1000, 8, 300, 320, 6, 400),      # BASELINE configs[1] at its stated size and samples per step
    ("C3", 3840, 2160, 16, 1000, 320, 6, 120),    # configs[2]: 8.3 G path-samples in one step, depth 16
    ("C5", 1104, 1000, 8, 300, 1024, 8, 40),      # configs[4]: NIF 8x1024
])
def test_configs_at_their_stated_sizes_and_sample_counts(oracle, ptmi_lib, name, W, H, depth, spp, hidden, nlayers, n_check):
    """The BASELINE configurations as stated (full image, full samples per step, their NIF), one step each: size-independent
    properties over every work item, and the oracle on a random subset of pixels taken from that same render (path structure
    identical, radiance within the stated NIF tolerance).  (configs[3] is configs[1]'s image over 8 GPUs: no such box.)"""
    layers = nif_assets.synthetic_nif(hidden=hidden, layer_count=nlayers, seed=2024 if hidden == 320 else 31)
    meta = nif_assets.URBAN_ALLEY_META
    r = ptmi_lib.Renderer(W, H, max_path_length=depth)
    r.init_nif_weights(layers, 12, meta["max"], nif_assets.folded_mean())
    r.init_render_settings(samples_per_step=spp)
    work = ptmi_lib.worklist(W, H)
    r.setup(work)
    r.path_trace()
    st = r.read_results(work)
    r.close()
    assert st.paths == W * H * spp and st.nif_launches >= 1
    assert np.all(work["sampleCount"] == spp)
    assert work["pathLength"].min() >= spp and work["pathLength"].max() <= depth * spp     # uint16 holds spp x depth here
    assert int(work["pathLength"].astype(np.int64).sum()) == st.segments
    assert 0.90 < st.escaped / st.paths < 0.97
    img = np.stack([work["r"], work["g"], work["b"]], -1)
    assert np.all(np.isfinite(img)) and img.min() >= 0
    sky = work["v"] < H // 3                                              # camera looks along -z: the upper third only sees sky
    assert np.all(work["pathLength"][sky] == spp) and np.all(work["r"][sky] > 0)
    got, ref = _same_render_subset(oracle, work, n_check, W, H, depth, spp, layers, seed={"C2": 2, "C3": 3, "C5": 5}[name], min_row=H // 3)
    assert np.array_equal(got["pathLength"], ref["pathLength"]), name
    for c in "rgb":
        np.testing.assert_allclose(got[c], ref[c], rtol=NIF_RTOL, atol=1e-5, err_msg=name)


def test_full_size_radiance_is_linear_in_the_environment_and_additive_over_steps(ptmi_lib):
    """Size-independent properties at BASELINE's full image (1104 x 1000, depth 8), no oracle needed: with a constant sky the
    accumulated radiance is env (.) throughput summed over the samples, so (a) doubling the environment doubles every
    accumulator EXACTLY (a power of two commutes with every fp32 rounding), (b) an environment with one channel carries
    nothing in the other two, (c) one step of 16 samples per pixel equals two steps of 8 bit for bit (the sample sequence and
    each pixel's summation order do not depend on how the samples are cut into steps: codelets.cpp:295-300), and path structure
    (pathLength, sampleCount) does not depend on the environment at all."""
    W, H, depth = 1104, 1000, 8

    def render(env, steps):
        r = ptmi_lib.Renderer(W, H, max_path_length=depth)
        r.set_constant_env(env)
        rec = ptmi_lib.worklist(W, H)
        r.setup(rec)
        for spp in steps:
            r.init_render_settings(seed=5, samples_per_step=spp)
            r.path_trace()
        r.read_results(rec)
        r.close()
        return rec

    one = render((0.75, 0.5, 1.25), [16])
    two = render((1.5, 1.0, 2.5), [16])
    for c in "rgb":
        assert np.array_equal(two[c], one[c] * np.float32(2.0)), c
    assert np.array_equal(two["pathLength"], one["pathLength"]) and np.array_equal(two["sampleCount"], one["sampleCount"])
    red = render((0.75, 0.0, 0.0), [16])
    assert np.array_equal(red["r"], one["r"]) and not red["g"].any() and not red["b"].any()
    split = render((0.75, 0.5, 1.25), [8, 8])
    assert split.tobytes() == one.tobytes()
    assert one["r"].max() > 0 and (one["pathLength"] > 16).any()


def test_full_size_nif_radiance_scales_with_the_decode_constant(ptmi_lib):
    """The same kind of property through the NIF stage at the full C2 shape (6 x 320 fp16 network, 1104 x 1000, depth 8): with a
    linear decode (log_tonemap = 0) and a zero mean the decoded environment is o x max (NifModel.cpp:226-233), so doubling `max`
    doubles every accumulated radiance EXACTLY -- through the fused head, the BGR -> RGB x throughput product
    (codelets.cpp:366-382) and the accumulate pass -- while the path structure stays what it was."""
    W, H, depth, spp = 1104, 1000, 8, 8
    layers = nif_assets.synthetic_nif()

    def render(max_value):
        r = ptmi_lib.Renderer(W, H, max_path_length=depth)
        r.init_nif_weights(layers, 12, max_value, [0.0, 0.0, 0.0], log_tonemap=False)
        r.init_render_settings(seed=9, samples_per_step=spp)
        rec = ptmi_lib.worklist(W, H)
        r.setup(rec)
        r.path_trace()
        st = r.read_results(rec)
        r.close()
        return rec, st

    a, sa = render(1.5)
    b, sb = render(3.0)
    assert (sa.paths, sa.segments, sa.escaped) == (sb.paths, sb.segments, sb.escaped) and sa.escaped > 0.9 * sa.paths
    assert np.array_equal(a["pathLength"], b["pathLength"])
    for c in "rgb":
        assert np.isfinite(a[c]).all() and np.abs(a[c]).max() > 0
        assert np.array_equal(b[c], a[c] * np.float32(2.0)), c
