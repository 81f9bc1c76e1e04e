"""Edge cases of the boundary on the GPU: empty and ragged worklists, padding items, NIF hot-swap, accumulator clear,
device-side HDR export, several handles at once, seeds."""
import numpy as np
import pytest

from ipu_path_trace_amd import nif_assets

pytestmark = pytest.mark.gpu


def _nif():
    return nif_assets.synthetic_nif(), nif_assets.URBAN_ALLEY_META["max"], nif_assets.folded_mean()


def test_empty_and_single_item_worklists(oracle, ptmi_lib):
    r = ptmi_lib.Renderer(64, 64, max_work_items=128)
    r.set_constant_env((1, 1, 1))
    r.init_render_settings(samples_per_step=3)
    empty = np.zeros(0, dtype=ptmi_lib.TRACE_DTYPE)
    r.setup(empty)
    r.path_trace()
    st = r.read_results(empty)
    assert st.paths == 0 and st.escaped == 0
    one = np.zeros(1, dtype=ptmi_lib.TRACE_DTYPE)
    one["u"], one["v"] = 31, 50
    r.setup(one)
    r.path_trace()
    r.read_results(one)
    ref = np.zeros(1, dtype=ptmi_lib.TRACE_DTYPE)
    ref["u"], ref["v"] = 31, 50
    oracle.render(oracle.make_config(width=64, height=64), None, ref, 0, 3)
    assert one.tobytes() == ref.tobytes()
    with pytest.raises(ptmi_lib.PtError):
        r.setup(np.zeros(129, dtype=ptmi_lib.TRACE_DTYPE))     # larger than max_work_items
    with pytest.raises(ptmi_lib.PtError):
        r.read_results(np.zeros(2, dtype=ptmi_lib.TRACE_DTYPE))  # size differs from setup
    r.close()


@pytest.mark.parametrize("n", [63, 65, 255, 257, 4097])
def test_ragged_worklist_sizes_with_padding_items(oracle, ptmi_lib, n):
    """Sizes around wave / workgroup / chunk boundaries; includes (65535, 65535) padding and pre-loaded accumulators."""
    W, H = 300, 200
    rng = np.random.default_rng(n)
    rec = np.zeros(n, dtype=ptmi_lib.TRACE_DTYPE)
    rec["u"] = rng.integers(0, W, n)
    rec["v"] = rng.integers(0, H, n)
    rec["u"][::17] = 65535
    rec["v"][::17] = 65535
    rec["r"] = rng.random(n).astype(np.float32)           # accumulators carried in are kept and added to
    rec["sampleCount"] = 2
    rec["pathLength"] = 5
    ref = rec.copy()
    layers, mx, mean = _nif()
    r = ptmi_lib.Renderer(W, H, max_work_items=n, max_path_length=7, iterations_per_batch=3)
    r.init_nif_weights(layers, 12, mx, mean)
    r.init_render_settings(samples_per_step=5)
    r.setup(rec)
    r.path_trace()
    st = r.read_results(rec)
    cfg = oracle.make_config(width=W, height=H, max_path_length=7, env_mode=oracle.ENV_NIF)
    ost = oracle.render(cfg, oracle.Nif(layers, 12, mx, mean), ref, 0, 5)
    assert (st.paths, st.segments, st.escaped) == (ost.paths, ost.segments, ost.escaped)
    assert np.array_equal(rec["pathLength"], ref["pathLength"]) and np.array_equal(rec["sampleCount"], ref["sampleCount"])
    for c in "rgb":
        np.testing.assert_allclose(rec[c], ref[c], rtol=2e-2, atol=1e-6)
    r.close()


def test_nif_hot_swap_and_environment_switch(oracle, ptmi_lib):
    """init_nif_weights may be called again (PathTracerApp.cpp:548-557); switching to a constant sky and back."""
    W = H = 48
    a = nif_assets.synthetic_nif(seed=1)
    b = nif_assets.synthetic_nif(hidden=128, layer_count=4, seed=2)
    mx, mean = nif_assets.URBAN_ALLEY_META["max"], nif_assets.folded_mean()
    r = ptmi_lib.Renderer(W, H, max_path_length=5)
    u = np.linspace(0.01, 0.99, 500, dtype=np.float32)
    v = u[::-1].copy()
    r.init_nif_weights(a, 12, mx, mean)
    ya = r.nif_infer(u, v)
    r.init_nif_weights(b, 12, mx, mean)
    yb = r.nif_infer(u, v)
    r.init_nif_weights(a, 12, mx, mean)
    np.testing.assert_array_equal(r.nif_infer(u, v), ya)           # deterministic, state fully replaced
    assert np.abs(ya - yb).max() > 1e-3
    np.testing.assert_allclose(yb, oracle.Nif(b, 12, mx, mean).infer(u, v), rtol=2e-2)
    r.set_constant_env((2, 2, 2))
    r.init_render_settings(samples_per_step=2)
    rec = ptmi_lib.worklist(W, H)
    r.setup(rec)
    r.path_trace()
    r.read_results(rec)
    assert rec["r"][0] == 4.0                                       # top-left pixel: sky, 2 samples x 2.0
    r.close()


def test_clear_accumulators_and_device_hdr_export(ptmi_lib):
    import torch
    W = H = 64
    r = ptmi_lib.Renderer(W, H, max_path_length=6)
    r.set_constant_env((1.0, 0.5, 0.25))
    r.init_render_settings(samples_per_step=4)
    rec = ptmi_lib.worklist(W, H)
    r.setup(rec)
    r.path_trace()
    out = torch.zeros((rec.size, 3), dtype=torch.float32, device="cuda")
    r.export_hdr_device(out.data_ptr(), rec.size)
    r.synchronize()
    r.read_results(rec)
    exp = np.stack([rec["b"], rec["g"], rec["r"]], -1) * (np.float32(1.0) / rec["sampleCount"].astype(np.float32))[:, None]
    np.testing.assert_array_equal(out.cpu().numpy(), exp)            # (b,g,r)/sampleCount as AccumulatedImage adds
    r.clear_accumulators()                                         # LoadBalancer.cpp:198-213 on the device
    r.read_results(rec)
    assert np.all(rec["r"] == 0) and np.all(rec["sampleCount"] == 0) and np.all(rec["pathLength"] == 0)
    assert rec["u"][5] == 5                                        # coordinates are kept
    out.fill_(7.0)
    r.export_hdr_device(out.data_ptr(), rec.size)                  # no samples yet: exports 0, not 0/0
    r.synchronize()
    assert torch.count_nonzero(out).item() == 0
    r.close()


def test_two_handles_and_seed_semantics(ptmi_lib):
    W = H = 40
    rs = [ptmi_lib.Renderer(W, H, max_path_length=6) for _ in range(2)]
    recs = []
    for i, r in enumerate(rs):
        r.set_constant_env((1, 1, 1))
        r.init_render_settings(seed=1 + i, samples_per_step=4)
        rec = ptmi_lib.worklist(W, H)
        r.setup(rec)
        recs.append(rec)
    for r in rs:
        r.path_trace()
    for r, rec in zip(rs, recs):
        r.read_results(rec)
    assert recs[0].tobytes() != recs[1].tobytes()                   # different seeds, different samples
    # same seed on the second handle reproduces the first exactly
    rs[1].init_render_settings(seed=1, samples_per_step=4)
    again = ptmi_lib.worklist(W, H)
    rs[1].setup(again)
    rs[1].path_trace()
    rs[1].read_results(again)
    assert again.tobytes() == recs[0].tobytes()
    for r in rs:
        r.close()


def test_rccl_gather_of_hdr_tiles_single_rank(ptmi_lib):
    """pt_gather_hdr at N = 1 on a real RCCL communicator of one rank (ncclGetUniqueId + ncclCommInitRank inside
    libptmi.so): the SAME calls as for N ranks run -- the slot-size all-reduce on the non-blocking communicator, the
    (empty) group of receives, the polled host and stream waits -- and the result is the rank's own tile, the slot
    zero-padded.  N > 1 cannot run on a one-GPU box (RCCL refuses two ranks on one device); the tile placement it feeds
    is covered by tests/test_partition.py."""
    W = H = 48
    r = ptmi_lib.Renderer(W, H, max_path_length=6, max_work_items=W * H + 100)
    r.set_constant_env((0.25, 0.5, 1.0))
    r.init_render_settings(samples_per_step=3)
    rec = ptmi_lib.worklist(W, H)
    r.setup(rec)
    r.path_trace()
    a = r.gather_hdr(W * H + 100)                                   # no communicator yet: same call, one tile
    r.comm_init_rank(ptmi_lib.comm_unique_id(), 0, 1)
    assert r.comm_info() == (0, 1)
    b = r.gather_hdr(W * H + 100)
    r.read_results(rec)
    exp = np.stack([rec["b"], rec["g"], rec["r"]], -1) * (np.float32(1.0) / rec["sampleCount"].astype(np.float32))[:, None]
    assert a.shape == b.shape == (1, W * H + 100, 3)
    np.testing.assert_array_equal(b[0, : W * H], exp)
    np.testing.assert_array_equal(a, b)
    assert not b[0, W * H:].any()                                    # padding of the slot
    with pytest.raises(ptmi_lib.PtError):
        r.gather_hdr(10)                                             # slot smaller than the tile
    with pytest.raises(ptmi_lib.PtError):
        r.gather_hdr(W * H, source=ptmi_lib.HDR_FILM)                # no resident film yet
    # resident film: two steps folded on the device == AccumulatedImage::accumulate on the host, bit for bit
    r.setup(ptmi_lib.worklist(W, H))
    film = np.zeros((W * H, 3), dtype=np.float32)
    for _ in range(2):
        r.path_trace()
        r.read_results(rec)
        film += np.stack([rec["b"], rec["g"], rec["r"]], -1) * (np.float32(1.0) / rec["sampleCount"].astype(np.float32))[:, None]
        r.film_accumulate()
        r.read_results(rec)
        assert not rec["sampleCount"].any() and not rec["r"].any() and not rec["pathLength"].any()
    f = r.gather_hdr(W * H + 100, source=ptmi_lib.HDR_FILM)
    np.testing.assert_array_equal(f[0, : W * H], film)
    assert not f[0, W * H:].any()
    with pytest.raises(ptmi_lib.PtError):
        r.comm_init_rank(ptmi_lib.comm_unique_id(), 0, 1)            # one communicator per handle
    r.close()


def test_comm_init_all_for_the_handles_of_one_process(ptmi_lib):
    """pt_comm_init_all is what `ipu_trace --ipus N` calls (one process driving N GPUs: a group of non-blocking
    ncclCommInitRankConfig calls, polled per rank).  On a one-GPU box: the group of ONE handle comes up and gathers; two
    handles on the same GPU are refused by name before anything is created (RCCL needs one device per rank), and both
    handles stay usable without a communicator."""
    import ctypes as C
    lib = ptmi_lib.load_library()
    W = H = 32

    def renderer():
        r = ptmi_lib.Renderer(W, H, max_path_length=4)
        r.set_constant_env((0.5, 0.5, 0.5))
        r.init_render_settings(samples_per_step=2)
        r.setup(ptmi_lib.worklist(W, H))
        r.path_trace()
        return r

    a, b = renderer(), renderer()
    two = (C.c_void_p * 2)(a.handle, b.handle)
    assert lib.pt_comm_init_all(two, 2) == -1                              # PT_ERR_INVALID_ARGUMENT
    assert b"share device 0" in lib.pt_last_error(a.handle)
    assert a.comm_info() == (0, 1) and b.comm_info() == (0, 1)
    one = (C.c_void_p * 1)(a.handle)
    assert lib.pt_comm_init_all(one, 1) == 0, lib.pt_last_error(a.handle)
    assert a.comm_info() == (0, 1)
    ga, gb = a.gather_hdr(W * H), b.gather_hdr(W * H)                      # a: through its communicator; b: without one
    np.testing.assert_array_equal(ga, gb)
    assert lib.pt_comm_init_all(one, 1) == -1                              # one communicator per handle
    a.close(); b.close()


def test_send_and_receive_on_the_non_blocking_communicator(ptmi_lib):
    """The point-to-point half of pt_gather_hdr (grouped ncclSend / ncclRecv on the non-blocking communicator, completion
    polled with ncclCommGetAsyncError and hipStreamQuery against the deadline) as far as a one-GPU box can run it: a rank
    sending 1.2 MB to itself on a communicator of one rank (test build: pt_diag_comm_self_exchange).  The exchange between
    two ranks needs two GPUs and has never run."""
    import ctypes as C
    diag = ptmi_lib.load_library(diag=True)
    r = ptmi_lib.Renderer(32, 32, max_path_length=4, diag=True)
    r.comm_set_timeout(30000)
    r.comm_init_rank(ptmi_lib.comm_unique_id(), 0, 1)
    ok = C.c_int(0)
    for floats in (3, 300000):
        assert diag.pt_diag_comm_self_exchange(r.handle, floats, C.byref(ok)) == 0, diag.pt_last_error(r.handle)
        assert ok.value == 1
    # the communicator is still good for the gather afterwards
    r.set_constant_env((1, 1, 1))
    r.init_render_settings(samples_per_step=2)
    r.setup(ptmi_lib.worklist(32, 32))
    r.path_trace()
    assert r.gather_hdr(32 * 32).shape == (1, 32 * 32, 3)
    r.close()


def test_failed_path_trace_drains_and_the_handle_stays_usable(oracle, ptmi_lib):
    """A launch failure in the MIDDLE of the batch loop (injected through pt_diag_inject_fault, which exists only in the test
    build libptmi_diag.so -- same sources, -DPTMI_DIAG_BUILD; the product library has no hook) must come back as an error with all
    three streams drained -- the earlier batches are already queued when it happens -- must not advance the sample
    sequence, and must leave the handle usable: after a fresh setup the same step gives the oracle's result."""
    O = oracle
    W = H = 48
    L, mx, mean = _nif()
    r = ptmi_lib.Renderer(W, H, max_path_length=6, iterations_per_batch=2, diag=True)      # 6 spp = 3 batches
    r.init_nif_weights(L, 12, mx, mean)
    r.init_render_settings(samples_per_step=6)
    rec = ptmi_lib.worklist(W, H)
    r.setup(rec)
    diag = ptmi_lib.load_library(diag=True)
    assert diag.pt_diag_inject_fault(r.handle, 1) == 0
    with pytest.raises(ptmi_lib.PtError) as e:
        r.path_trace()
    assert e.value.code == -3 and "injected fault" in str(e.value)
    assert diag.pt_diag_inject_fault(r.handle, -1) == 0
    r.synchronize()                                                             # nothing left running
    rec = ptmi_lib.worklist(W, H)
    r.setup(rec)                                                                # accumulators are undefined after a failure
    r.path_trace()                                                              # same sample indices 0..5 as the failed step
    st = r.read_results(rec)
    cfg = O.make_config(width=W, height=H, max_path_length=6, env_mode=O.ENV_NIF)
    ref = O.worklist(W, H)
    ost = O.render(cfg, O.Nif(L, 12, mx, mean), ref, 0, 6)
    assert np.array_equal(rec["pathLength"], ref["pathLength"]) and (st.paths, st.segments, st.escaped) == (ost.paths, ost.segments, ost.escaped)
    for c in "rgb":
        np.testing.assert_allclose(rec[c], ref[c], rtol=2e-2, atol=1e-6)
    r.close()


def test_tile_costs_and_film_seed_contract(ptmi_lib):
    """pt_tile_costs / pt_film_seed argument checks and semantics: costs need pt_tile_costs_enable and the grid's exact size;
    padding items belong to no tile; pt_setup starts the sums afresh; a seeded film is what the gather returns and what
    later steps add to."""
    W, H = 40, 24                                                              # 3 x 2 tiles of 16 x 16
    r = ptmi_lib.Renderer(W, H, max_path_length=5, max_work_items=W * H + 7)
    r.set_constant_env((1.0, 1.0, 1.0))
    r.init_render_settings(samples_per_step=3)
    rec = np.zeros(W * H + 7, dtype=ptmi_lib.TRACE_DTYPE)
    rec[: W * H] = ptmi_lib.worklist(W, H)
    rec["u"][W * H:] = 65535
    rec["v"][W * H:] = 65535                                                   # padding items (LoadBalancer.cpp:66-71)
    r.setup(rec)
    lib = ptmi_lib.load_library()
    out = np.zeros(6, dtype=np.uint64)
    assert lib.pt_tile_costs(r.handle, out.ctypes.data, 6) == -5               # PT_ERR_NOT_READY: not enabled
    with pytest.raises(ptmi_lib.PtError):
        r.tile_costs_enable(0, 16)
    r.tile_costs_enable(16, 16)
    assert lib.pt_tile_costs(r.handle, out.ctypes.data, 5) == -1               # wrong grid size
    r.path_trace()
    got = r.read_results(rec)
    costs = r.tile_costs(W, H)
    real = rec[: W * H]
    t = (real["v"].astype(np.int64) // 16) * 3 + real["u"].astype(np.int64) // 16
    np.testing.assert_array_equal(costs, np.bincount(t, weights=real["pathLength"], minlength=6).astype(np.uint64))
    assert costs.sum() == got.segments - int(rec["pathLength"][W * H:].sum())  # the padding items' paths are in no tile
    r.setup(rec)                                                               # accumulators carried in are kept (as in the reference) ...
    np.testing.assert_array_equal(r.tile_costs(W, H), costs)                   # ... so their path lengths still count, once
    for c in ("r", "g", "b", "sampleCount", "pathLength"):
        rec[c] = 0
    r.setup(rec)
    assert not r.tile_costs(W, H).any()                                        # a new worklist starts new sums
    # film seed: exactly the current items, then the film is seed + the steps' means
    seed = np.arange(rec.size * 3, dtype=np.float32).reshape(-1, 3)
    with pytest.raises(ptmi_lib.PtError):
        r.film_seed(seed[:-1])
    r.film_seed(seed)
    np.testing.assert_array_equal(r.gather_hdr(rec.size, source=ptmi_lib.HDR_FILM)[0], seed)
    r.path_trace()
    r.read_results(rec)
    mean = np.stack([rec["b"], rec["g"], rec["r"]], -1) * (np.float32(1.0) / rec["sampleCount"].astype(np.float32))[:, None]
    r.film_accumulate()
    np.testing.assert_array_equal(r.gather_hdr(rec.size, source=ptmi_lib.HDR_FILM)[0], seed + mean)
    r.close()


def test_nif_calibration_and_kernel_name(oracle, ptmi_lib):
    """ABI 4: pt_calibrate_nif re-runs the NIF stage of the last step's largest batch alone -- it must report that batch's
    queue length, take a sane time, and leave the worklist's accumulators alone (a step after it continues exactly as a step
    without it would); pt_nif_kernel_name names what the library dispatched for each kind of model.  Before any NIF step, and
    with a constant environment only, there is nothing to calibrate: PT_ERR_NOT_READY."""
    W = H = 64
    layers, mx, mean = _nif()
    r = ptmi_lib.Renderer(W, H, max_path_length=6)
    assert r.nif_kernel_name() == ""
    r.set_constant_env((1, 1, 1))
    r.init_render_settings(samples_per_step=5)
    rec = ptmi_lib.worklist(W, H)
    r.setup(rec)
    r.path_trace()
    with pytest.raises(ptmi_lib.PtError) as e:
        r.calibrate_nif()
    assert e.value.code == -5
    r.init_nif_weights(layers, 12, mx, mean)
    with pytest.raises(ptmi_lib.PtError) as e:
        r.calibrate_nif()                                    # a model, but no step with it yet
    assert e.value.code == -5
    r.setup(rec)
    r.path_trace()
    st = r.stats()
    ms, evals = r.calibrate_nif(launches=3)
    assert 0 < evals <= st.escaped and ms > 0                # one batch of the step (here the whole step: 5 iterations fit one batch... or its largest)
    assert r.nif_kernel_name().startswith("nif_kernel_v3<320, 12")
    r.path_trace()
    got = ptmi_lib.worklist(W, H)
    r.read_results(got)
    # the same two steps without a calibration in between
    r2 = ptmi_lib.Renderer(W, H, max_path_length=6)
    r2.init_nif_weights(layers, 12, mx, mean)
    r2.init_render_settings(samples_per_step=5)
    r2.setup(rec)
    r2.path_trace()                                          # r traced samples 0-4 with the constant sky first: skip them here too
    r2.setup(rec)
    r2.path_trace()
    r2.path_trace()
    want = ptmi_lib.worklist(W, H)
    r2.read_results(want)
    assert got.tobytes() == want.tobytes()
    # the other two dispatch paths name themselves too
    r.init_nif_weights(nif_assets.synthetic_nif(hidden=512, layer_count=3), 12, mx, mean)
    r.nif_infer(np.array([0.3], np.float32), np.array([0.6], np.float32))
    assert "nifg16_layer_kernel" in r.nif_kernel_name() and "hidden 512" in r.nif_kernel_name()
    r.init_nif_weights(nif_assets.synthetic_nif(hidden=64, layer_count=3, dtype=np.float32), 12, mx, mean)
    r.nif_infer(np.array([0.3], np.float32), np.array([0.6], np.float32))
    assert "nif32_layer_kernel" in r.nif_kernel_name()
    r.close()
    r2.close()


def test_out_of_memory_is_its_own_status(ptmi_lib):
    """A worklist capacity the device cannot hold: pt_create reports PT_ERR_OUT_OF_MEMORY (-6, include/ptmi.h), frees what it
    had allocated, and the next, sane pt_create works (the reference's counterpart: Poplar's graph compilation failing with
    an out-of-memory report, src/ipu_utils.hpp:532-535)."""
    with pytest.raises(ptmi_lib.PtError) as e:
        ptmi_lib.Renderer(65535, 65535, max_work_items=1500000000)       # ~370 GB of worklist, queue and batch buffers
    assert e.value.code == -6, (e.value.code, str(e.value))
    assert "hipMalloc" in str(e.value) or "dev_alloc" in str(e.value)
    r = ptmi_lib.Renderer(32, 32, max_path_length=4)
    r.set_constant_env((1.0, 1.0, 1.0))
    r.init_render_settings(samples_per_step=2)
    rec = ptmi_lib.worklist(32, 32)
    r.setup(rec)
    r.path_trace()
    assert r.read_results(rec).paths == 32 * 32 * 2
    r.close()


def test_uint16_record_fields_at_their_limits(oracle, ptmi_lib):
    """TraceRecord::sampleCount and ::pathLength are uint16 (TraceRecord.hpp:10-11).  65535 samples in one step is the most the
    boundary accepts (65536 would wrap sampleCount to 0 and the host divides by it, AccumulatedImage.cpp:69-71): sampleCount
    comes back as 65535, pathLength -- about 1.6 x 65535 path segments per pixel -- WRAPS exactly as the reference's field does,
    pt_stats.segments keeps the exact 64-bit total, and every radiance sum (65535 fp32 additions per pixel, in sample order)
    equals the oracle's bit for bit."""
    W = H = 12
    spp = 65535
    r = ptmi_lib.Renderer(W, H, max_path_length=8)
    r.set_constant_env((0.5, 1.0, 0.25))
    with pytest.raises(ptmi_lib.PtError):
        r.init_render_settings(samples_per_step=65536)
    r.init_render_settings(samples_per_step=spp)
    got = ptmi_lib.worklist(W, H)
    r.setup(got)
    r.path_trace()
    st = r.read_results(got)
    r.close()
    cfg = oracle.make_config(width=W, height=H, max_path_length=8, env_rgb=(0.5, 1.0, 0.25), fold=oracle.FOLD_FORWARD)
    ref = oracle.worklist(W, H)
    ost = oracle.render(cfg, None, ref, 0, spp)
    assert (st.paths, st.segments, st.escaped) == (ost.paths, ost.segments, ost.escaped) and st.paths == W * H * spp
    assert st.segments > 65535 * W * H                                  # more segments than the 16-bit fields can count per pixel ...
    assert (got["sampleCount"] == 65535).all()
    assert got.tobytes() == ref.tobytes()                               # ... and the wrapped pathLength equals the oracle's, as do the sums
    assert int(got["pathLength"].astype(np.int64).sum()) != st.segments  # the wire field really wrapped
