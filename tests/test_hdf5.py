"""the NIF asset loader (Keras H5 -> Dense layers) without libhdf5."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from ipu_path_trace_amd import nif_assets
from tests.h5_writer import write_keras_h5

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ipu_path_trace_amd", "host")
REAL = "/usr/local/lib/python3.10/dist-packages/scipy/io/matlab/tests/data/testhdf5_7.4_GLNX86.mat"


@pytest.fixture(scope="module")
def host():
    subprocess.check_call(["make", "-C", HOST, "-s"])
    L = C.CDLL(os.path.join(HOST, "libpthost.so"))
    st = C.c_size_t
    L.pth_h5_list.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, st]
    L.pth_h5_attr.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, st]
    L.pth_h5_dataset.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, st, C.c_char_p, st]
    L.pth_h5_model.argtypes = [C.c_char_p, C.c_void_p, st, C.c_char_p, st]
    return L


def _dataset(L, f, path, cap=1 << 22):
    dims, el = (C.c_size_t * 8)(), C.c_size_t()
    raw, err = np.zeros(cap, np.uint8), C.create_string_buffer(512)
    r = L.pth_h5_dataset(f.encode(), path.encode(), dims, C.byref(el), raw.ctypes.data, cap, err, 512)
    if r < 0:
        raise RuntimeError(err.value.decode())
    shape = tuple(dims[:r])
    return shape, el.value, raw[: int(np.prod(shape)) * el.value]


@pytest.mark.skipif(not os.path.exists(REAL), reason="scipy's MATLAB v7.3 fixture is not installed")
def test_reader_on_a_file_written_by_the_real_hdf5_library(host):
    """MATLAB v7.3 = HDF5 behind a 512-byte user block, old-style group, contiguous float64 dataset, string attribute."""
    buf = C.create_string_buffer(4096)
    assert host.pth_h5_list(REAL.encode(), b"/", buf, 4096) == 1 and buf.value == b"testdouble\n"
    shape, el, raw = _dataset(host, REAL, "/testdouble")
    assert shape == (9, 1) and el == 8
    np.testing.assert_allclose(raw.view(np.float64), np.linspace(0, 2 * np.pi, 9), rtol=1e-15)
    assert host.pth_h5_attr(REAL.encode(), b"/testdouble", b"MATLAB_class", buf, 4096) == 6 and buf.value == b"double"


@pytest.mark.parametrize("vlen,user_block,cap", [(False, 0, 8), (True, 0, 8), (False, 512, 3)])
def test_keras_h5_roundtrip(host, tmp_path, vlen, user_block, cap):
    """A Keras-shaped converted.hdf5 (fixed- or variable-length model_config, multi-node group B-tree) loads into the
    same Dense layers: names, shapes, activations, raw fp16 bytes."""
    layers = nif_assets.synthetic_nif(hidden=64, layer_count=6, seed=3)
    layers[2] = (layers[2][0], None, True)                       # one layer without a bias
    p = str(tmp_path / "converted.hdf5")
    write_keras_h5(p, layers, vlen_config=vlen, user_block=user_block, snod_capacity=cap)
    buf = C.create_string_buffer(1 << 16)
    assert host.pth_h5_list(p.encode(), b"/model_weights", buf, 1 << 16) == 7
    assert buf.value.decode().split() == sorted("dense" if i == 0 else "dense_%d" % i for i in range(7))
    assert host.pth_h5_attr(p.encode(), b"/", b"keras_version", buf, 1 << 16) > 0 and buf.value == b"2.8.0"
    n = host.pth_h5_attr(p.encode(), b"/", b"model_config", buf, 1 << 16)
    assert n > 100 and b'"Functional"' in buf.value
    shape, el, raw = _dataset(host, p, "/model_weights/dense_3/dense_3/kernel:0")
    assert shape == layers[3][0].shape and el == 2 and raw.tobytes() == layers[3][0].tobytes()
    # through Hdf5Model + NifModel::Data::setupModel: summary "name rows cols half relu bias sha..." per layer
    out = np.zeros(64, np.uint64)
    err = C.create_string_buffer(512)
    nl = host.pth_h5_model(p.encode(), out.ctypes.data, out.size, err, 512)
    assert nl == 7, err.value
    for i, (k, b, relu) in enumerate(layers):
        rows, cols, half, is_relu, has_bias, ksum = [int(v) for v in out[6 * i: 6 * i + 6]]
        assert (rows, cols) == k.shape and half == 1 and is_relu == int(relu) and has_bias == int(b is not None)
        assert ksum == int(k.view(np.uint16).astype(np.uint64).sum())


@pytest.mark.parametrize("opts", [
    dict(chunks=(16, 24)),                                                    # chunked, no filter (edge chunks on both axes)
    dict(chunks=(16, 24), layout_version=1),                                  # the version-1 layout message
    dict(chunks=(8, 8), leaf_fan=5),                                          # two-level chunk B-tree
    dict(chunks=(32, 32), compression="gzip"),                                # h5py compression="gzip"
    dict(chunks=(32, 32), compression="gzip", shuffle=True),                  # ... with shuffle=True
    dict(chunks=(16, 64), compression="gzip", shuffle=True, fletcher32=True),
    dict(chunks=(64, 64), shuffle=True),
    dict(chunks=(32, 32), compression="gzip", fletcher32=True, fletcher_first=True),                # h5repack -f FLET -f GZIP order:
    dict(chunks=(16, 64), compression="gzip", shuffle=True, fletcher32=True, fletcher_first=True),  # checksum bytes inside the deflated stream
])
def test_chunked_and_filtered_datasets(host, tmp_path, opts):
    """Hdf5Model.cpp:96-133 reads the variables through libhdf5, which takes any layout; a Keras file whose weights were
    written chunked (h5py does that for any compression / shuffle / resizable dataset) must load into the same layers."""
    layers = nif_assets.synthetic_nif(hidden=80, layer_count=3, seed=5)        # 48x80, 80x80 (+48 concat), 80x3: ragged chunk edges
    p = str(tmp_path / "chunked.hdf5")
    write_keras_h5(p, layers, **opts)
    for i, (k, b, _) in enumerate(layers):
        name = "dense" if i == 0 else "dense_%d" % i
        shape, el, raw = _dataset(host, p, "/model_weights/%s/%s/kernel:0" % (name, name))
        assert shape == k.shape and el == 2 and raw.tobytes() == k.tobytes(), (name, opts)
        shape, el, raw = _dataset(host, p, "/model_weights/%s/%s/bias:0" % (name, name))
        assert shape == b.shape and raw.tobytes() == b.tobytes()
    out, err = np.zeros(64, np.uint64), C.create_string_buffer(512)
    assert host.pth_h5_model(p.encode(), out.ctypes.data, out.size, err, 512) == len(layers), err.value
    for i, (k, b, relu) in enumerate(layers):
        assert int(out[6 * i + 5]) == int(k.view(np.uint16).astype(np.uint64).sum())


def test_missing_chunks_read_as_zeros_and_unknown_filters_are_named(host, tmp_path):
    from tests.h5_writer import H5Writer
    a = np.arange(40 * 30, dtype=np.float32).reshape(40, 30)
    w = H5Writer()
    root = w.group({"sparse": w.dataset(a, chunks=(16, 16), skip_chunks=(1, 4)),
                    "float32_gzip": w.dataset(a, chunks=(16, 16), compression="gzip", shuffle=True),
                    "zstd": w.dataset(a, chunks=(16, 16), filter_ids=[32015]),
                    "szip": w.dataset(a, chunks=(16, 16), compression="gzip", filter_ids=[4]),
                    "never_written": w.dataset(a, chunks=(16, 16), skip_chunks=tuple(range(6)))})
    p = str(tmp_path / "f.h5")
    open(p, "wb").write(w.finish(root))
    exp = a.copy()
    exp[0:16, 16:30] = 0                                                      # chunk 1 = rows 0-15, columns 16-29
    exp[32:40, 0:16] = 0                                                      # chunk 4 = rows 32-39, columns 0-15
    shape, el, raw = _dataset(host, p, "/sparse")
    assert shape == (40, 30) and el == 4 and np.array_equal(raw.view(np.float32).reshape(40, 30), exp)
    shape, el, raw = _dataset(host, p, "/float32_gzip")
    assert np.array_equal(raw.view(np.float32).reshape(40, 30), a)
    shape, el, raw = _dataset(host, p, "/never_written")
    assert not raw.any()
    with pytest.raises(RuntimeError, match=r"filter 32015 \(zstd\)"):
        _dataset(host, p, "/zstd")
    with pytest.raises(RuntimeError, match=r"filter 4 \(szip\)"):
        _dataset(host, p, "/szip")


def test_a_chunk_index_that_points_at_itself_is_refused_at_once(host, tmp_path):
    """A crafted chunk B-tree whose internal node lists itself 64 times would cost 64^16 visits under a depth limit alone:
    a child must sit exactly one level below its parent, and no file holds more nodes than its size allows."""
    import time
    from tests.h5_writer import H5Writer
    a = np.arange(64 * 64, dtype=np.float32).reshape(64, 64)
    w = H5Writer()
    root = w.group({"loop": w.dataset(a, chunks=(16, 16), cyclic_index=True)})
    p = str(tmp_path / "loop.h5")
    open(p, "wb").write(w.finish(root))
    t = time.perf_counter()
    with pytest.raises(RuntimeError, match="one level below its parent"):
        _dataset(host, p, "/loop")
    assert time.perf_counter() - t < 1.0


def test_unsupported_content_is_reported(host, tmp_path):
    p = tmp_path / "not_hdf5.h5"
    p.write_bytes(b"this is not an HDF5 file" * 10)
    buf = C.create_string_buffer(512)
    assert host.pth_h5_list(str(p).encode(), b"/", buf, 512) == -1 and b"no HDF5 signature" in buf.value
    layers = nif_assets.synthetic_nif(hidden=64, layer_count=2)
    good = str(tmp_path / "m.hdf5")
    write_keras_h5(good, layers)
    assert host.pth_h5_attr(good.encode(), b"/", b"missing_attr", buf, 512) == -1
    dims, el = (C.c_size_t * 8)(), C.c_size_t()
    assert host.pth_h5_dataset(good.encode(), b"/model_weights/nope", dims, C.byref(el), None, 0, buf, 512) == -1
    assert b"no object 'nope'" in buf.value
    trunc = tmp_path / "trunc.hdf5"
    trunc.write_bytes(open(good, "rb").read()[:600])
    assert host.pth_h5_list(str(trunc).encode(), b"/model_weights", buf, 512) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,opts", [(np.float16, {}), (np.float32, dict(chunks=(64, 64), compression="gzip", shuffle=True))])
def test_cli_renders_from_converted_hdf5(host, oracle, tmp_path, dtype, opts):
    """--assets <dir> with nif_metadata.txt + converted.hdf5 (the reference's own asset layout), end to end: a float16 model
    in contiguous datasets, and a float32 model (it then runs in float, NifModel.cpp:314) in gzip-compressed chunks."""
    exe = os.path.join(HOST, "ipu_trace")
    W, H, spp = 64, 48, 4
    assets = tmp_path / "assets.extra"
    assets.mkdir()
    layers = nif_assets.synthetic_nif(dtype=dtype)
    nif_assets.write_metadata(str(assets / "nif_metadata.txt"))
    write_keras_h5(str(assets / "converted.hdf5"), layers, vlen_config=True, **opts)
    out = tmp_path / "img.png"
    r = subprocess.run([exe, "--assets", str(assets), "-w", str(W), "-h", str(H), "-s", str(spp), "--samples-per-step",
                        str(spp), "--max-path-length", "5", "-o", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    film = np.zeros((H, W, 3), dtype=np.float32)
    ww, hh = C.c_size_t(), C.c_size_t()
    host.pth_read_exr.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    assert host.pth_read_exr(str(tmp_path / "img.exr").encode(), film.ctypes.data, film.size, C.byref(ww), C.byref(hh)) == 0
    cfg = oracle.make_config(width=W, height=H, max_path_length=5, env_mode=oracle.ENV_NIF)
    ref = oracle.worklist(W, H)
    oracle.render(cfg, oracle.Nif(layers, 12, nif_assets.URBAN_ALLEY_META["max"], nif_assets.folded_mean()), ref, 0, spp)
    exp = np.stack([ref["b"], ref["g"], ref["r"]], -1).reshape(H, W, 3) / spp
    np.testing.assert_allclose(film, exp, rtol=2e-2, atol=1e-6)


def test_truncated_and_corrupted_files_fail_cleanly(tmp_path):
    """User-supplied assets: every truncation and a few hundred byte-level corruptions of a valid Keras H5 (and of a
    PTNIF side-car) must end in a clean exception.  The loader is built with AddressSanitizer + UBSan here, so an
    out-of-bounds read that happens not to crash is caught as well."""
    exe = str(tmp_path / "h5fuzz")
    srcs = [os.path.join(HOST, f) for f in ("Hdf5Reader.cpp", "Hdf5Model.cpp", "NifModel.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + HOST, "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "h5fuzz_main.cpp")] + srcs + ["-lz", "-Wl,--unresolved-symbols=ignore-all"])
    layers = nif_assets.synthetic_nif(hidden=32, layer_count=2, embedding_dim=4)
    meta = tmp_path / "nif_metadata.txt"
    nif_assets.write_metadata(str(meta), dict(nif_assets.URBAN_ALLEY_META, embedding_dimension=4, hidden_size=32, layer_count=2))
    good = tmp_path / "good.hdf5"
    write_keras_h5(str(good), layers, vlen_config=True)
    chunked = tmp_path / "chunked.hdf5"                                  # chunk B-tree, inflate and un-shuffle under the fuzzer too
    write_keras_h5(str(chunked), layers, chunks=(8, 16), compression="gzip", shuffle=True, fletcher32=True, leaf_fan=3)
    ptn = tmp_path / "good.ptnif"
    nif_assets.write_ptnif(str(ptn), layers, 4)
    rng = np.random.default_rng(0)
    files = [str(good), str(chunked), str(ptn)]
    for src, ext in ((good, ".hdf5"), (chunked, ".c.hdf5"), (ptn, ".ptnif")):
        raw = np.fromfile(str(src), dtype=np.uint8)
        cuts = sorted(set(list(range(0, min(raw.size, 700), 7)) + list(rng.integers(0, raw.size, 60))))
        for k, n in enumerate(cuts):                                   # truncations
            p = tmp_path / ("cut%d%s" % (k, ext))
            raw[:n].tofile(str(p))
            files.append(str(p))
        for k in range(150):                                           # corruptions: 1..4 bytes set to hostile values
            m = raw.copy()
            for pos in rng.integers(0, min(raw.size, 4096), rng.integers(1, 5)):
                m[pos] = rng.choice([0, 1, 0x7f, 0x80, 0xff, int(rng.integers(0, 256))])
            p = tmp_path / ("bad%d%s" % (k, ext))
            m.tofile(str(p))
            files.append(str(p))
    r = subprocess.run([exe, str(meta)] + files, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1"))
    assert r.returncode == 0, r.stderr[-3000:]
    loaded, rejected = [int(x) for x in r.stdout.strip().splitlines()[-1].split()[1::2]]
    assert loaded >= 3 and rejected > 150 and loaded + rejected == len(files)
