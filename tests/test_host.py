"""Host-side C++ mirror of the reference interface (ipu_path_trace_amd/host) through its extern "C" test shim."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from ipu_path_trace_amd import nif_assets
from ipu_path_trace_amd.ptmi import TRACE_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ipu_path_trace_amd", "host")


@pytest.fixture(scope="module")
def host():
    subprocess.check_call(["make", "-C", HOST, "-s"])
    L = C.CDLL(os.path.join(HOST, "libpthost.so"))
    st = C.c_size_t
    L.pth_calculate_max_rays_per_tile.restype = st
    L.pth_calculate_max_rays_per_tile.argtypes = [st, st, st, st]
    L.pth_make_shuffled_worklist.restype = st
    L.pth_make_shuffled_worklist.argtypes = [st, st, st, st, C.c_void_p, st]
    L.pth_balance_and_clear.restype = st
    L.pth_balance_and_clear.argtypes = [C.c_void_p, st, st, C.c_int]
    L.pth_film_roundtrip.argtypes = [C.c_void_p, st, st, st, st, C.c_float, C.c_float, C.c_char_p, C.c_void_p, C.c_void_p]
    L.pth_read_exr.argtypes = [C.c_char_p, C.c_void_p, st, C.POINTER(st), C.POINTER(st)]
    L.pth_read_metadata.argtypes = [C.c_char_p, C.c_void_p]
    L.pth_round_samples.restype = st
    L.pth_round_samples.argtypes = [st, st]
    L.pth_split_pixels.restype = None
    L.pth_split_pixels.argtypes = [st, st, C.c_void_p]
    L.pth_job_build.argtypes = [st, st, st, st, C.c_void_p, C.c_void_p]
    return L


def test_calculate_max_rays_per_tile_table(host):
    """LoadBalancer.cpp:14-36 including its `r += r % workers` rounding (values verified in SURVEY.md 8(c))."""
    f = host.pth_calculate_max_rays_per_tile
    assert f(256, 256, 1472, 6) == 48          # ceil(65536/1472)=45 -> 45 + 45%6 = 48
    assert f(1104, 1000, 1472, 6) == 750
    assert f(3840, 2160, 1472, 6) == 5636       # 5635 -> 5636
    assert f(526, 526, 1472, 6) == 190          # 188 -> 190 (not a multiple of 6)
    assert f(4, 4, 1472, 6) == 6                # minimum = worker count
    assert host.pth_round_samples(100000, 300) == 100200 and host.pth_round_samples(512, 512) == 512


def test_shuffled_padded_worklist(host):
    w, h, tiles, workers = 100, 70, 64, 6
    n = host.pth_make_shuffled_worklist(w, h, tiles, workers, None, 0)
    per = host.pth_calculate_max_rays_per_tile(w, h, tiles, workers)
    assert n == per * tiles >= w * h
    rec = np.zeros(n, dtype=TRACE_DTYPE)
    assert host.pth_make_shuffled_worklist(w, h, tiles, workers, rec.ctypes.data, n) == n
    pad = (rec["u"] == 65535) & (rec["v"] == 65535)   # LoadBalancer.cpp:66-71
    assert pad.sum() == n - w * h
    real = rec[~pad]
    key = real["v"].astype(np.int64) * w + real["u"]
    assert np.array_equal(np.sort(key), np.arange(w * h))          # every pixel exactly once
    assert not np.array_equal(key, np.arange(w * h))               # shuffled (mt19937 seed 142)
    rec2 = np.zeros(n, dtype=TRACE_DTYPE)
    host.pth_make_shuffled_worklist(w, h, tiles, workers, rec2.ctypes.data, n)
    assert rec.tobytes() == rec2.tobytes()                         # deterministic
    assert np.all(rec["r"] == 0) and np.all(rec["sampleCount"] == 0)


def test_balance_by_path_length_and_clear(host):
    rng = np.random.default_rng(0)
    jobs, per = 8, 10
    rec = np.zeros(jobs * per, dtype=TRACE_DTYPE)
    rec["u"] = np.arange(rec.size)
    rec["pathLength"] = rng.integers(1, 200, rec.size)
    rec["r"] = 1.0
    rec["sampleCount"] = 3
    before = rec.copy()
    total = host.pth_balance_and_clear(rec.ctypes.data, rec.size, jobs, 1)
    assert total == int(before["pathLength"].astype(np.int64).sum())
    assert sorted(rec["u"].tolist()) == list(range(rec.size))       # a permutation of the items
    # every job received pairs (shortest, longest): per-job sums of the path lengths are balanced
    lengths = rec["pathLength"].reshape(jobs, per).astype(np.int64)
    assert lengths.sum(axis=1).max() - lengths.sum(axis=1).min() < 0.35 * lengths.sum(axis=1).mean()
    s = np.sort(before["pathLength"])
    assert lengths[0, 0] == s[0] and lengths[0, 1] == s[-1]
    # clearInactiveAccumulators (LoadBalancer.cpp:198-213): zero r,g,b,sampleCount,pathLength, keep u,v, return the sum
    rec2 = before.copy()
    assert host.pth_balance_and_clear(rec2.ctypes.data, rec2.size, jobs, 0) == total
    assert np.all(rec2["pathLength"] == 0) and np.all(rec2["r"] == 0) and np.all(rec2["sampleCount"] == 0)
    assert np.array_equal(rec2["u"], before["u"])


@pytest.mark.parametrize("n,jobs", [(7, 2), (9, 4), (3, 8), (1, 1), (30, 4), (81, 8)])
def test_balance_is_a_permutation_for_any_item_count(host, n, jobs):
    """The reference's dealing loop (LoadBalancer.cpp:160-177) duplicates and drops items when the count is not a
    multiple of 2 x jobs (unreachable with its default geometry); here every input item comes out exactly once."""
    rng = np.random.default_rng(n)
    rec = np.zeros(n, dtype=TRACE_DTYPE)
    rec["u"] = np.arange(n)
    rec["pathLength"] = rng.integers(1, 50, n)
    before = rec.copy()
    total = host.pth_balance_and_clear(rec.ctypes.data, n, jobs, 1)
    assert total == int(before["pathLength"].sum())
    assert sorted(rec["u"].tolist()) == list(range(n))
    assert sorted(rec["pathLength"].tolist()) == sorted(before["pathLength"].tolist())


def test_tile_dealing_matches_the_python_partition(host):
    """The C++ host's balancer across devices (dealTilesByPathLength / tileWorkList, the multi-GPU form of
    LoadBalancer::allocateWorkByPathLength, LoadBalancer.cpp:141-192) and partition.py derive the same deal and the same
    worklists: a render driven by `ipu_trace --ipus N` and one driven by N Python ranks place every pixel alike."""
    from ipu_path_trace_amd import partition
    W, H = 200, 136                                             # 13 x 9 tiles of 16 x 16, ragged right and bottom edges
    n_tiles = int(np.prod(partition.tile_grid(W, H)))
    rng = np.random.default_rng(11)
    cost = rng.integers(0, 5000, n_tiles).astype(np.uint64)
    cost[rng.integers(0, n_tiles, 20)] = 777                    # ties: broken by tile id on both sides
    host.pth_deal_tiles.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    host.pth_tile_worklist.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p]
    host.pth_tile_worklist.restype = C.c_long
    host.pth_max_tile_items.argtypes = [C.c_size_t] * 3
    host.pth_max_tile_items.restype = C.c_size_t
    for devices in (1, 2, 3, 8):
        owner = np.zeros(n_tiles, dtype=np.int32)
        host.pth_deal_tiles(cost.ctypes.data, n_tiles, devices, owner.ctypes.data)
        np.testing.assert_array_equal(owner, partition.deal_by_path_length(cost, devices))
        cap = host.pth_max_tile_items(W, H, devices)
        assert cap == partition.max_items_per_rank(W, H, devices)
        seen = np.zeros((H, W), dtype=np.int32)
        for d in range(devices):
            rec = np.zeros(cap, dtype=TRACE_DTYPE)
            assert host.pth_tile_worklist(W, H, owner.ctypes.data, n_tiles, d, cap, rec.ctypes.data) == cap
            ref = partition.worklist_for_owner(W, H, owner, d)
            assert rec[: ref.size].tobytes() == ref.tobytes()
            assert (rec["u"][ref.size:] == 65535).all() and (rec["v"][ref.size:] == 65535).all()      # padding items
            seen[ref["v"], ref["u"]] += 1
        assert (seen == 1).all()
    # a deal that does not fit the capacity is refused, not truncated
    owner = np.zeros(n_tiles, dtype=np.int32)
    rec = np.zeros(16, dtype=TRACE_DTYPE)
    assert host.pth_tile_worklist(W, H, owner.ctypes.data, n_tiles, 0, 16, rec.ctypes.data) == -1


def test_ipu_path_trace_job_interface(host):
    """IpuPathTraceJob(maxRayCount, args, core), buildGraph, beginTraceJob/endTraceJob, splitTilePixelsOverWorkers
    (IpuPathTraceJob.hpp:43-54, IpuPathTraceJob.cpp:30-52,95-138)."""
    out = (C.c_size_t * 12)()
    host.pth_split_pixels(750, 6, out)
    assert list(out) == [0, 125, 125, 250, 250, 375, 375, 500, 500, 625, 625, 750]
    host.pth_split_pixels(10, 4, out)                                   # leftovers go to the first workers
    assert list(out)[:8] == [0, 3, 3, 6, 6, 8, 8, 10]
    o = (C.c_size_t * 10)()
    f = (C.c_float * 2)()
    assert host.pth_job_build(750, 41, 2, 3000, o, f) == 32             # numChannels 3, numRayDirComponents 2
    assert list(o) == [750, 41, 2, 3000, 750, 3000, 6, 256, 256, 3]     # CLI defaults: 256 x 256, roulette depth 3
    assert f[0] == pytest.approx(1.5) and f[1] == pytest.approx(0.3)


def test_async_task_surfaces_exceptions(host):
    assert host.pth_async_task_rethrows() == 1


def test_film_accumulate_tonemap_and_exr(host, tmp_path):
    w, h, steps = 12, 7, 3
    rec = np.zeros(w * h + 5, dtype=TRACE_DTYPE)
    rr, cc = np.divmod(np.arange(w * h), w)
    rec["u"][: w * h], rec["v"][: w * h] = cc, rr
    rec["u"][w * h:] = 65535
    rec["v"][w * h:] = 65535                                        # padding is skipped (AccumulatedImage.cpp:66-67)
    rng = np.random.default_rng(1)
    for c in "rgb":
        rec[c] = rng.random(rec.size).astype(np.float32) * 4
    rec["sampleCount"] = 4
    hdr = np.zeros((h, w, 3), dtype=np.float32)
    ldr = np.zeros((h, w, 3), dtype=np.uint8)
    out = str(tmp_path / "img.png")
    assert host.pth_film_roundtrip(rec.ctypes.data, rec.size, w, h, steps, 0.5, 2.2, out.encode(), hdr.ctypes.data,
                                   ldr.ctypes.data) == 0
    exp = np.zeros((h, w, 3), dtype=np.float32)
    for _ in range(steps):                                          # hdr(v,u) += (b,g,r)/sampleCount
        exp[rr, cc, 0] += rec["b"][: w * h] * np.float32(0.25)
        exp[rr, cc, 1] += rec["g"][: w * h] * np.float32(0.25)
        exp[rr, cc, 2] += rec["r"][: w * h] * np.float32(0.25)
    np.testing.assert_array_equal(hdr, exp)
    tone = np.power(exp / steps * np.float32(2 ** 0.5), 1 / 2.2) * 255.0
    assert np.abs(ldr.astype(np.int32) - np.clip(np.rint(tone), 0, 255).astype(np.int32)).max() <= 1
    # the files: PNG decodes to the LDR image (RGB order), EXR holds hdr/steps in B,G,R channels
    from PIL import Image
    png = np.asarray(Image.open(out))
    assert png.shape == (h, w, 3) and np.array_equal(png[..., ::-1], ldr)
    back = np.zeros((h, w, 3), dtype=np.float32)
    ww, hh = C.c_size_t(), C.c_size_t()
    assert host.pth_read_exr(str(tmp_path / "img.exr").encode(), back.ctypes.data, back.size, C.byref(ww), C.byref(hh)) == 0
    assert (ww.value, hh.value) == (w, h)
    # cv::imwrite picks the codec by the file name's extension (AccumulatedImage.cpp:49): every built-in writer decodes to the
    # same LDR image, and an extension without a writer is refused
    for name in ("a.bmp", "a.PPM", "a.pnm", "a.tif", "a.TIFF"):
        other = str(tmp_path / name)
        assert host.pth_film_roundtrip(rec.ctypes.data, rec.size, w, h, steps, 0.5, 2.2, other.encode(), hdr.ctypes.data, ldr.ctypes.data) == 0
        img = np.asarray(Image.open(other).convert("RGB"))
        assert img.shape == (h, w, 3) and np.array_equal(img[..., ::-1], ldr), name
    # JPEG is lossy: a smooth 40 x 27 image (not a multiple of 8: edge blocks) must come back within a few grey levels at quality 95
    w2, h2 = 40, 27
    rec2 = np.zeros(w2 * h2, dtype=TRACE_DTYPE)
    r2, c2 = np.divmod(np.arange(w2 * h2), w2)
    rec2["u"], rec2["v"] = c2, r2
    rec2["r"] = (0.2 + 0.8 * c2 / w2) * 4
    rec2["g"] = (0.2 + 0.8 * r2 / h2) * 4
    rec2["b"] = (0.5 + 0.4 * np.sin(c2 / 5.0) * np.cos(r2 / 4.0)) * 4
    rec2["sampleCount"] = 4
    hdr2, ldr2 = np.zeros((h2, w2, 3), np.float32), np.zeros((h2, w2, 3), np.uint8)
    for name in ("b.jpg", "b.JPEG"):
        jp = str(tmp_path / name)
        assert host.pth_film_roundtrip(rec2.ctypes.data, rec2.size, w2, h2, 1, 0.0, 2.2, jp.encode(), hdr2.ctypes.data, ldr2.ctypes.data) == 0
        img = np.asarray(Image.open(jp).convert("RGB")).astype(np.int32)
        err = np.abs(img[..., ::-1] - ldr2.astype(np.int32))
        assert img.shape == (h2, w2, 3) and err.max() <= 12 and err.mean() < 2.0, (name, err.max(), err.mean())
    # TIFF 6.0 wants the IFD on a word boundary: an image with an ODD pixel count (40 x 27 x 3 bytes of strip is even; 3 x 3 is
    # not) gets a pad byte behind its strip, and the header's IFD offset is even
    w3 = h3 = 3
    rec3 = np.zeros(w3 * h3, dtype=TRACE_DTYPE)
    r3, c3 = np.divmod(np.arange(w3 * h3), w3)
    rec3["u"], rec3["v"], rec3["r"], rec3["g"], rec3["b"], rec3["sampleCount"] = c3, r3, 0.1 * (1 + c3), 0.1 * (1 + r3), 0.3, 1
    hdr3, ldr3 = np.zeros((h3, w3, 3), np.float32), np.zeros((h3, w3, 3), np.uint8)
    odd = str(tmp_path / "odd.tif")
    assert host.pth_film_roundtrip(rec3.ctypes.data, rec3.size, w3, h3, 1, 0.0, 2.2, odd.encode(), hdr3.ctypes.data, ldr3.ctypes.data) == 0
    raw = open(odd, "rb").read()
    ifd = int.from_bytes(raw[4:8], "little")
    assert ifd % 2 == 0 and ifd == 8 + 27 + 1 and raw[8 + 27] == 0
    assert np.array_equal(np.asarray(Image.open(odd).convert("RGB"))[..., ::-1], ldr3)
    assert host.pth_film_roundtrip(rec.ctypes.data, rec.size, w, h, steps, 0.5, 2.2, str(tmp_path / "a.webp").encode(), hdr.ctypes.data,
                                   ldr.ctypes.data) != 0
    np.testing.assert_allclose(back, exp / steps, rtol=1e-6)


def test_metadata_parser_matches_reference_fixture(host, tmp_path):
    """The C++ parser on a file with the reference's nif_metadata.txt fields (values from the real fixture)."""
    p = tmp_path / "nif_metadata.txt"
    nif_assets.write_metadata(str(p))
    out = (C.c_double * 8)()
    assert host.pth_read_metadata(str(p).encode(), out) == 0
    assert list(out[:4]) == [12, 320, 6, 1]
    assert out[4] == pytest.approx(3.4299468994140625)
    np.testing.assert_allclose(list(out[5:8]), nif_assets.folded_mean(), rtol=1e-7)
    bad = tmp_path / "bad.txt"
    bad.write_text("{\"embedding_dimension\": 12}")
    assert host.pth_read_metadata(str(bad).encode(), out) == -1
    assert host.pth_read_metadata(str(tmp_path / "missing.txt").encode(), out) == -1


REAL_META = os.path.join(ROOT, "tests", "golden", "urban_alley_01_4k_fp16_yuv", "assets.extra", "nif_metadata.txt")


def test_reference_metadata_file_parses_identically_in_cpp_and_python(host):
    """The reference's own nif_metadata.txt (a data fixture, see its README): both parsers against the values written in
    the file itself, and the hand-copied constants of nif_assets.URBAN_ALLEY_META against the file
    (NifMetaData.cpp:11-71: fields, the -eps fold into mean :48-53, --layer-size / --layer-count from train_command :56-65)."""
    doc = json.load(open(REAL_META))
    assert doc["train_command"].count("--callback-period") == 2 and "embedding_sigma" in doc    # the quirks are in the file
    out = (C.c_double * 8)()
    assert host.pth_read_metadata(REAL_META.encode(), out) == 0
    m = nif_assets.load_metadata(REAL_META)
    enc = doc["encode_params"]
    assert (int(out[0]), int(out[1]), int(out[2]), int(out[3])) == (12, 320, 6, 1)
    assert (m["embedding_dimension"], m["hidden_size"], m["layer_count"], m["log_tone_map"]) == (12, 320, 6, True)
    assert out[4] == np.float32(enc["max"]) == np.float32(m["max"])
    folded = [np.float32(np.float32(x) - np.float32(enc["eps"])) for x in enc["mean"]]
    assert [np.float32(x) for x in out[5:8]] == folded == [np.float32(x) for x in m["mean_folded"]]
    assert m["original_image_shape"] == [2048, 4096, 3] and m["name"].endswith("urban_alley_01_4k.exr")
    # the constants the benchmark and the tests use are these, not a re-typed approximation
    U = nif_assets.URBAN_ALLEY_META
    assert (U["embedding_dimension"], U["hidden_size"], U["layer_count"]) == (12, 320, 6)
    assert U["max"] == enc["max"] and U["mean"] == enc["mean"] and U["eps"] == enc["eps"]
    assert U["log_tone_map"] == enc["log_tone_map"] and U["original_image_shape"] == doc["original_image_shape"]
    assert nif_assets.folded_mean() == [float(x) for x in folded]


def test_cli_contract_without_gpu(host, tmp_path):
    """CLI surface of main.cpp:8-37 + PathTracerApp.cpp:794-830: names, short forms, required options, errors."""
    exe = os.path.join(HOST, "ipu_trace")
    help_text = subprocess.run([exe, "--help"], capture_output=True, text=True).stdout
    for opt in ["--outfile", "--save-interval", "--width", "--height", "--samples", "--samples-per-step",
                "--interactive-samples", "--refractive-index", "--roulette-depth", "--stop-prob", "--aa-noise-scale",
                "--fov", "--exposure", "--gamma", "--env-map-rotation", "--seed", "--aa-noise-type", "--codelet-path",
                "--enable-load-balancing", "--max-path-length", "--assets", "--partials-type",
                "--available-memory-proportion", "--max-nif-batch-size", "--ui-port", "--model", "--ipus", "--save-exe",
                "--load-exe", "--compile-only", "--defer-attach", "--log-level"]:
        assert opt in help_text, opt
    for short in ["-o", "-w", "-h", "-s", "-n", "-a"]:
        assert "[ %s ]" % short in help_text
    r = subprocess.run([exe, "-w", "32"], capture_output=True, text=True)
    assert r.returncode != 0 and "required but missing" in r.stdout
    r = subprocess.run([exe, "-o", "x.png", "--assets", str(tmp_path), "--bogus"], capture_output=True, text=True)
    assert r.returncode != 0 and "unrecognised option" in r.stdout
    r = subprocess.run([exe, "-o", "x.png", "--assets", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode != 0 and "Could not load NIF model" in r.stdout    # PathTracerApp.cpp:69-71
    # values the step loop would divide by are rejected with a message, not a crash
    for opt, msg in (("--ipus", "--ipus must be at least 1"), ("--save-interval", "--save-interval must be at least 1"),
                     ("--samples-per-step", "--samples-per-step must be at least 1")):
        r = subprocess.run([exe, "-o", "x.png", "--assets", str(tmp_path), "--constant-env", "1,1,1", opt, "0"],
                           capture_output=True, text=True)
        assert r.returncode == 1 and msg in r.stdout, (opt, r.returncode, r.stdout[-500:])
    # --devices (this build's addition) must name one GPU ordinal per logical device; refused before any device is touched
    for bad in ("0", "0,x", "0,-1", "0,,1"):
        r = subprocess.run([exe, "-o", "x.png", "--assets", str(tmp_path), "--constant-env", "1,1,1", "--ipus", "2", "--devices", bad],
                           capture_output=True, text=True)
        assert r.returncode == 1 and "--devices" in r.stdout, (bad, r.returncode, r.stdout[-500:])
    assert "--devices" in help_text and "--host-gather" in help_text
    # --compile-only: the reference compiles its graph and stops before attaching (ipu_utils.hpp:523-526); here the options and
    # the assets are validated and nothing is rendered -- it works without a GPU and writes no image
    r = subprocess.run([exe, "-o", str(tmp_path / "never.png"), "--assets", str(tmp_path), "--constant-env", "1,1,1", "--compile-only",
                        "--save-exe", "graph"], capture_output=True, text=True)
    assert r.returncode == 0 and "Compile only mode selected: finished." in r.stdout, r.stdout[-500:]
    assert not (tmp_path / "never.png").exists()
    r = subprocess.run([exe, "-o", str(tmp_path / "x.webp"), "--assets", str(tmp_path), "--constant-env", "1,1,1", "--compile-only"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "could not find a writer for the specified extension" in r.stdout   # cv::imwrite's refusal, at start-up
    r = subprocess.run([exe, "-o", str(tmp_path / "never.png"), "--assets", str(tmp_path), "--compile-only"], capture_output=True, text=True)
    assert r.returncode != 0 and "Could not load NIF model" in r.stdout      # a bad asset directory still fails the "compile"


@pytest.mark.gpu
def test_ipu_trace_end_to_end_matches_oracle(host, oracle, tmp_path):
    """Drop-in CLI on the GPU: shuffled padded worklist, two steps, film accumulate, EXR; against the oracle."""
    O = oracle
    exe = os.path.join(HOST, "ipu_trace")
    W, H, spp, steps = 96, 64, 5, 2
    assets = tmp_path / "assets.extra"
    assets.mkdir()
    layers = nif_assets.synthetic_nif()
    nif_assets.write_metadata(str(assets / "nif_metadata.txt"))
    nif_assets.write_ptnif(str(assets / "converted.ptnif"), layers, 12)
    out = tmp_path / "img.png"
    r = subprocess.run([exe, "--assets", str(assets), "-w", str(W), "-h", str(H), "-s", str(spp * steps),
                        "--samples-per-step", str(spp), "--max-path-length", "6", "--env-map-rotation", "30",
                        "-o", str(out), "--save-interval", "1", "--ipus", "1", "--defer-attach"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "Completed render step 2/2" in r.stdout and "Samples/sec:" in r.stdout
    film = np.zeros((H, W, 3), dtype=np.float32)
    ww, hh = C.c_size_t(), C.c_size_t()
    assert host.pth_read_exr(str(tmp_path / "img.exr").encode(), film.ctypes.data, film.size, C.byref(ww), C.byref(hh)) == 0
    cfg = O.make_config(width=W, height=H, max_path_length=6, env_mode=O.ENV_NIF, env_rotation_degrees=30.0)
    ref = O.worklist(W, H)
    O.render(cfg, O.Nif(layers, 12, nif_assets.URBAN_ALLEY_META["max"], nif_assets.folded_mean()), ref, 0, spp * steps)
    exp = np.stack([ref["b"], ref["g"], ref["r"]], -1).reshape(H, W, 3) / (spp * steps)
    np.testing.assert_allclose(film, exp, rtol=2e-2, atol=1e-6)
    assert os.path.getsize(out) > 1000


def _run_cli(host, tmp_path, name, extra, W=96, H=80, spp=4, steps=3):
    exe = os.path.join(HOST, "ipu_trace")
    assets = tmp_path / "assets.extra"
    if not assets.exists():
        assets.mkdir()
        nif_assets.write_metadata(str(assets / "nif_metadata.txt"))
        nif_assets.write_ptnif(str(assets / "converted.ptnif"), nif_assets.synthetic_nif(), 12)
    out = tmp_path / (name + ".png")
    r = subprocess.run([exe, "--assets", str(assets), "-w", str(W), "-h", str(H), "-s", str(spp * steps),
                        "--samples-per-step", str(spp), "--max-path-length", "7", "-o", str(out), "--save-interval", "2"] + extra,   # (a later --save-interval in `extra` wins)
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    film = np.zeros((H, W, 3), dtype=np.float32)
    ww, hh = C.c_size_t(), C.c_size_t()
    assert host.pth_read_exr(str(tmp_path / (name + ".exr")).encode(), film.ctypes.data, film.size, C.byref(ww), C.byref(hh)) == 0
    return film, r.stdout


@pytest.mark.gpu
def test_cli_resident_film_host_film_and_load_balancing_give_the_same_film(host, oracle, tmp_path):
    """SURVEY.md rows A17/A18/N3 on the GPU.  Four step loops over the same render:
    * default: film resident on the device (pt_film_accumulate), one pt_gather_hdr per save interval;
    * --host-film: the reference's loop (setup -> path_trace -> read_results every step, host film);
    * --enable-load-balancing: STILL the resident film -- per-tile path-length sums (pt_tile_costs: kilobytes) leave the
      device at the save intervals, image tiles are re-dealt by cost and the film follows its pixels (pt_film_seed);
    * --host-film --enable-load-balancing: the reference's loop + LoadBalancer::allocateWorkByPathLength from step 2 on
      (LoadBalancer.cpp:141-192).
    The RNG is keyed by pixel and absolute sample index and the film arithmetic is the same fp32 expressions in the same
    order, so all four films are bit-identical -- also over several re-deals -- and they match the oracle within the NIF
    tolerance."""
    import re
    O = oracle
    W, H, spp, steps = 96, 80, 4, 3
    resident, log = _run_cli(host, tmp_path, "resident", [])
    assert "Saved images at step 2" in log and "Saved images at step 3" in log and "Completed render step 3/3" in log
    assert "Step loop: film resident on the device" in log
    hostfilm, log = _run_cli(host, tmp_path, "hostfilm", ["--host-film"])
    assert "Step loop: the reference's" in log
    balanced, log = _run_cli(host, tmp_path, "balanced", ["--enable-load-balancing", "--log-level", "debug"])
    assert "Load balancing finished" in log and "Step loop: film resident on the device" in log
    shipped, trace_buffer = map(int, re.search(r"Load balancing: (\d+) bytes of tile costs from each device \(the trace buffer is (\d+) bytes\)", log).groups())
    assert shipped == 30 * 8 and shipped * 100 < trace_buffer          # 6 x 5 tiles of 16 x 16: 240 B against 153,600 B
    refbal, log = _run_cli(host, tmp_path, "refbal", ["--host-film", "--enable-load-balancing"])
    assert "Load balancing finished" in log and "Step loop: the reference's" in log
    assert resident.tobytes() == hostfilm.tobytes()
    assert balanced.tobytes() == hostfilm.tobytes()
    assert refbal.tobytes() == hostfilm.tobytes()
    # several re-deals (save interval 1, 5 steps): every pixel's sum continues in step order through the seeds
    many, log = _run_cli(host, tmp_path, "many", ["--enable-load-balancing", "--save-interval", "1"], steps=5)
    plain, _ = _run_cli(host, tmp_path, "plain5", [], steps=5)
    assert log.count("Load balancing finished") == 4 and many.tobytes() == plain.tobytes()
    cfg = O.make_config(width=W, height=H, max_path_length=7, env_mode=O.ENV_NIF)
    ref = O.worklist(W, H)
    O.render(cfg, O.Nif(nif_assets.synthetic_nif(), 12, nif_assets.URBAN_ALLEY_META["max"], nif_assets.folded_mean()), ref, 0, spp * steps)
    exp = np.stack([ref["b"], ref["g"], ref["r"]], -1).reshape(H, W, 3) / (spp * steps)
    np.testing.assert_allclose(resident, exp, rtol=2e-2, atol=1e-6)


@pytest.mark.gpu
def test_interactive_restart_and_nif_hot_reload_over_the_ui_port(host, tmp_path):
    """SURVEY.md row N4 without the reference's transport (packetcomms / videolib are absent): a text client on
    --ui-port drives the state machine of PathTracerApp.cpp:507-564,643-686 -- a changed setting restarts the render at
    step 1 with --interactive-samples per step and a fresh film, settings are re-sent at steps 1 and 5 (where the sample
    count reverts to --samples-per-step), load_nif hot-swaps the weights through pt_upload_nif, exposure does not restart,
    stop ends the run."""
    import socket
    import threading
    import time
    exe = os.path.join(HOST, "ipu_trace")
    assets = tmp_path / "assets.extra"
    assets.mkdir()
    nif_assets.write_metadata(str(assets / "nif_metadata.txt"))
    nif_assets.write_ptnif(str(assets / "converted.ptnif"), nif_assets.synthetic_nif(), 12)
    other = tmp_path / "other.extra"
    other.mkdir()
    nif_assets.write_metadata(str(other / "nif_metadata.txt"))
    nif_assets.write_ptnif(str(other / "converted.ptnif"), nif_assets.synthetic_nif(seed=99), 12)
    W, H = 64, 48
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "ui.png"
    proc = subprocess.Popen([exe, "--assets", str(assets), "-w", str(W), "-h", str(H), "-s", "4000000", "--samples-per-step", "20",
                             "--interactive-samples", "2", "--max-path-length", "5", "-o", str(out), "--save-interval", "3",
                             "--ui-port", str(port), "--log-level", "debug"], stdout=open(str(tmp_path / "cli.log"), "w"), stderr=subprocess.STDOUT, text=True)
    conn = None
    for _ in range(600):                                   # the server starts listening after the device is attached
        try:
            conn = socket.create_connection(("127.0.0.1", port), timeout=1.0)
            break
        except OSError:
            time.sleep(0.1)
    assert conn is not None, "ui server never came up"
    got = {"progress": [], "preview": 0, "hdr_rows": 0, "hdr_header": None, "rates": 0}

    def reader():
        f = conn.makefile("rb")
        while True:
            line = f.readline()
            if not line:
                return
            t = line.decode().split()
            if t[0] == "progress":
                got["progress"].append(float(t[1]))
            elif t[0] == "sample_rate":
                got["rates"] += 1
            elif t[0] == "render_preview":
                assert (int(t[1]), int(t[2]), int(t[3])) == (W, H, W * H * 3)
                assert len(f.read(int(t[3]))) == int(t[3])
                got["preview"] += 1
            elif t[0] == "hdr_header":
                got["hdr_header"] = (int(t[1]), int(t[2]), int(t[3]))
            elif t[0] == "hdr_packet":
                assert len(f.read(int(t[2]))) == int(t[2]) == W * 3 * 4
                got["hdr_rows"] += 1

    th = threading.Thread(target=reader, daemon=True)
    th.start()

    def wait_for(cond, what, seconds=60):
        t0 = time.time()
        while not cond():
            assert time.time() - t0 < seconds and proc.poll() is None, what
            time.sleep(0.02)

    wait_for(lambda: len(got["progress"]) >= 7, "no progress from the first render")           # past step 5: reverted to 20 spp
    conn.sendall(b"exposure 1.5\n")                                                              # host-side only: no restart
    n = len(got["progress"])
    wait_for(lambda: len(got["progress"]) >= n + 2, "render stalled after an exposure change")
    assert got["progress"][-1] > got["progress"][n - 1]                                          # still counting up
    conn.sendall(b"env_rotation 40\n")                                                           # restart at step 1
    m = len(got["progress"])
    wait_for(lambda: any(p < got["progress"][m - 1] for p in got["progress"][m:]), "no restart after env_rotation")
    conn.sendall(("load_nif %s\n" % other).encode())                                             # hot reload + restart
    time.sleep(0.5)
    wait_for(lambda: got["hdr_header"] is not None and got["hdr_rows"] >= H, "no raw film transfer at the save interval")
    conn.sendall(b"stop\n")
    proc.wait(timeout=60)
    log = open(str(tmp_path / "cli.log")).read()   # (a file, not a pipe: an unread pipe fills up and blocks the renderer in its logger)
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        open(os.path.join(ROOT, "gpurun_out", "ui_interactive.log"), "w").write(log + "\nGOT %r\n" % {k: (v if k != "progress" else len(v)) for k, v in got.items()})
    assert proc.returncode == 0, log[-3000:]
    assert "Rendering stopped by remote UI" in log and "Loading NIF: %s" % other in log
    assert log.count("Completed render step 1/200000") >= 3                                      # first run + two restarts
    assert "Interaction stopped reverting samples per step to: 20" in log
    assert got["hdr_header"] == (W, H, H) and got["preview"] >= 8 and got["rates"] >= 8
    assert os.path.getsize(out) > 500                                                            # the last film is left on disk


@pytest.mark.gpu
def test_film_after_a_nif_hot_reload_matches_the_oracle(host, oracle, tmp_path):
    """SURVEY.md row N4, the film itself (PathTracerApp.cpp:548-557: load_nif -> loadNifModels -> init_nif_weights again, and
    the render restarts at step 1 with a fresh film).  A client swaps the NIF for another one (seed 99) in the middle of a
    render and stops it a few steps later; the image left on disk must be the oracle's render WITH THE NEW NIF over exactly
    the sample indices the restarted render took (the sample sequence of a handle runs on through a restart; every step
    logs its first index).  --interactive-samples == --samples-per-step, so every step takes the same count."""
    import re
    import socket
    import threading
    import time
    O = oracle
    exe = os.path.join(HOST, "ipu_trace")
    assets = tmp_path / "assets.extra"
    assets.mkdir()
    nif_assets.write_metadata(str(assets / "nif_metadata.txt"))
    nif_assets.write_ptnif(str(assets / "converted.ptnif"), nif_assets.synthetic_nif(), 12)
    other = tmp_path / "other.extra"
    other.mkdir()
    nif_assets.write_metadata(str(other / "nif_metadata.txt"))
    new_nif = nif_assets.synthetic_nif(seed=99)
    nif_assets.write_ptnif(str(other / "converted.ptnif"), new_nif, 12)
    W, H, S, depth = 64, 48, 6, 5
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "reload.png"
    proc = subprocess.Popen([exe, "--assets", str(assets), "-w", str(W), "-h", str(H), "-s", str(S * 1000000), "--samples-per-step", str(S),
                             "--interactive-samples", str(S), "--max-path-length", str(depth), "-o", str(out), "--save-interval", "1000",
                             "--ui-port", str(port), "--log-level", "debug"], stdout=open(str(tmp_path / "cli.log"), "w"), stderr=subprocess.STDOUT, text=True)
    conn = None
    for _ in range(600):
        try:
            conn = socket.create_connection(("127.0.0.1", port), timeout=1.0)
            break
        except OSError:
            time.sleep(0.1)
    assert conn is not None, "ui server never came up"
    progress = []

    def reader():
        f = conn.makefile("rb")
        while True:
            line = f.readline()
            if not line:
                return
            t = line.decode().split()
            if t[0] == "progress":
                progress.append(float(t[1]))
            elif t[0] == "render_preview":
                f.read(int(t[3]))
            elif t[0] == "hdr_packet":
                f.read(int(t[2]))

    threading.Thread(target=reader, daemon=True).start()

    def wait_for(cond, what, seconds=60):
        t0 = time.time()
        while not cond():
            assert time.time() - t0 < seconds and proc.poll() is None, what
            time.sleep(0.005)

    wait_for(lambda: len(progress) >= 4, "no progress from the first render")
    conn.sendall(("load_nif %s\n" % other).encode())
    n = len(progress)
    wait_for(lambda: any(b < a for a, b in zip(progress[n - 1:], progress[n:])), "no restart after load_nif")
    m = len(progress)
    wait_for(lambda: len(progress) >= m + 6, "the restarted render does not advance")
    conn.sendall(b"stop\n")
    proc.wait(timeout=120)
    log = open(str(tmp_path / "cli.log")).read()   # (a file, not a pipe: an unread pipe fills up and blocks the renderer in its logger)
    assert proc.returncode == 0, log[-3000:]
    assert "Loading NIF: %s" % other in log and "Rendering stopped by remote UI" in log
    last = int(re.findall(r"Saved images at step (\d+)", log)[-1])
    tail = log[log.rindex("Step 1 took", 0, log.rindex("Completed render step 1/")):]   # from step 1 of the LAST render on
    firsts = {int(a): int(c) for a, b, c in re.findall(r"Step (\d+) took (\d+) samples per pixel from sample index (\d+)", tail)}
    assert last >= 6 and all(firsts[k] == firsts[1] + (k - 1) * S for k in range(1, last + 1)), (last, firsts)
    assert firsts[1] >= 4 * S                                    # the first render's steps came before
    film = np.zeros((H, W, 3), dtype=np.float32)
    ww, hh = C.c_size_t(), C.c_size_t()
    assert host.pth_read_exr(str(tmp_path / "reload.exr").encode(), film.ctypes.data, film.size, C.byref(ww), C.byref(hh)) == 0
    cfg = O.make_config(width=W, height=H, max_path_length=depth, env_mode=O.ENV_NIF)
    ref = O.worklist(W, H)
    O.render(cfg, O.Nif(new_nif, 12, nif_assets.URBAN_ALLEY_META["max"], nif_assets.folded_mean()), ref, firsts[1], last * S)
    exp = np.stack([ref["b"], ref["g"], ref["r"]], -1).reshape(H, W, 3) / (last * S)
    np.testing.assert_allclose(film, exp, rtol=2e-2, atol=1e-6)
    # and it is NOT the old NIF's image: the swap really took place
    old = O.worklist(W, H)
    O.render(cfg, O.Nif(nif_assets.synthetic_nif(), 12, nif_assets.URBAN_ALLEY_META["max"], nif_assets.folded_mean()), old, firsts[1], last * S)
    old_img = np.stack([old["b"], old["g"], old["r"]], -1).reshape(H, W, 3) / (last * S)
    assert np.abs(film - old_img).max() > 10 * np.abs(film - exp).max()


@pytest.mark.gpu
def test_restart_then_detach_still_reverts_to_the_full_sample_count(host, tmp_path):
    """PathTracerApp.cpp:656-686: the reversion to --samples-per-step at step 5 and the init_render_settings at steps 1
    and 5 do not depend on a UI server being attached.  A client that changes a setting (restart: the device goes to
    --interactive-samples) and then closes its socket before step 5 must not leave the render on the interactive
    sample count: steps >= 5 take the full count again, and the final image is written."""
    import re
    import socket
    import time
    exe = os.path.join(HOST, "ipu_trace")
    assets = tmp_path / "assets.extra"
    assets.mkdir()
    nif_assets.write_metadata(str(assets / "nif_metadata.txt"))
    nif_assets.write_ptnif(str(assets / "converted.ptnif"), nif_assets.synthetic_nif(), 12)
    W, H, full, inter, steps = 1104, 1000, 96, 48, 9           # big enough that a step takes tens of milliseconds
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "detach.png"
    proc = subprocess.Popen([exe, "--assets", str(assets), "-w", str(W), "-h", str(H), "-s", str(full * steps), "--samples-per-step", str(full),
                             "--interactive-samples", str(inter), "--max-path-length", "5", "-o", str(out), "--save-interval", "100",
                             "--ui-port", str(port), "--log-level", "debug"], stdout=open(str(tmp_path / "cli.log"), "w"), stderr=subprocess.STDOUT, text=True)
    conn = None
    for _ in range(600):
        try:
            conn = socket.create_connection(("127.0.0.1", port), timeout=1.0)
            break
        except OSError:
            time.sleep(0.1)
    assert conn is not None, "ui server never came up"
    f = conn.makefile("rb")

    def next_progress():
        while True:
            line = f.readline()
            assert line, "server closed the connection"
            t = line.decode().split()
            if t[0] == "progress":
                return float(t[1])
            if t[0] == "render_preview":
                f.read(int(t[3]))

    first = next_progress()
    conn.sendall(b"env_rotation 25\n")                       # restart at step 1 with the interactive sample count
    p = next_progress()
    while p > first + 1e-6 or p > 1.5 / steps:                 # wait for step 1 of the RESTARTED render ...
        first, p = min(first, p), next_progress()
    conn.shutdown(socket.SHUT_RDWR)                            # ... and leave at once: detached around step 2
    f.close()                                                  # (a makefile() object keeps the descriptor open)
    conn.close()
    proc.wait(timeout=180)
    log = open(str(tmp_path / "cli.log")).read()   # (a file, not a pipe: an unread pipe fills up and blocks the renderer in its logger)
    assert proc.returncode == 0, log[-3000:]
    assert "Remote UI disconnected." in log
    tail = log[log.rindex("Completed render step 1/%d" % steps):]          # the restarted render
    assert "Remote UI disconnected." in tail, "the client detached before the restart was processed: scenario not reached"
    took = {int(a): int(b) for a, b in re.findall(r"Step (\d+) took (\d+) samples per pixel", tail)}   # (the last occurrence of a step wins)
    gone = tail.index("Remote UI disconnected.")
    revert = tail.index("Interaction stopped reverting samples per step to: %d" % full)
    assert gone < revert, "the detach came after step 5: scenario not reached (steps too fast)"
    assert all(took[k] == full for k in range(5, steps + 1)), took       # full count again once the interaction is over
    assert took[2] == inter or took[3] == inter, took                     # and the interactive count before that
    assert "Saved images at step %d" % steps in log and os.path.getsize(out) > 500


@pytest.mark.gpu
def test_ui_client_cannot_abort_the_render_with_a_bad_value(host, tmp_path):
    """A value pt_set_render_settings would refuse (or that does not parse) is rejected by the server with a warning and
    the state kept; the render carries on and ends normally (a typo from the client aborted the render)."""
    import socket
    import time
    exe = os.path.join(HOST, "ipu_trace")
    assets = tmp_path / "assets.extra"
    assets.mkdir()
    nif_assets.write_metadata(str(assets / "nif_metadata.txt"))
    nif_assets.write_ptnif(str(assets / "converted.ptnif"), nif_assets.synthetic_nif(), 12)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "bad.png"
    proc = subprocess.Popen([exe, "--assets", str(assets), "-w", "64", "-h", "48", "-s", "10000000", "--samples-per-step", "10",
                             "--max-path-length", "4", "-o", str(out), "--ui-port", str(port)],
                            stdout=open(str(tmp_path / "cli.log"), "w"), stderr=subprocess.STDOUT, text=True)
    conn = None
    for _ in range(600):
        try:
            conn = socket.create_connection(("127.0.0.1", port), timeout=1.0)
            break
        except OSError:
            time.sleep(0.1)
    assert conn is not None
    import threading

    drained = {"bytes": 0, "end": None}

    def drain():                                                 # a client that stops reading would block the server's sends
        try:
            while True:
                d = conn.recv(1 << 20)
                if not d:
                    drained["end"] = "eof"
                    return
                drained["bytes"] += len(d)
        except OSError as e:
            drained["end"] = repr(e)

    conn.settimeout(None)                                        # (create_connection's timeout would end the drain on a quiet second)
    threading.Thread(target=drain, daemon=True).start()
    conn.sendall(b"interactive_samples 0\nfov 0\nfov 200\ninteractive_samples 70000\ninteractive_samples abc\nfov\ngamma 0\n"
                 b"env_rotation nan\ninteractive_samples 2.5\n" + b"x" * 10000 + b"\n")
    time.sleep(1.0)
    assert proc.poll() is None, "the render died on a bad value"
    conn.sendall(b"fov 60\n")                                   # a good value still restarts the render
    time.sleep(0.5)
    conn.sendall(b"stop\n")
    proc.wait(timeout=60)
    log = open(str(tmp_path / "cli.log")).read()   # (a file, not a pipe: an unread pipe fills up and blocks the renderer in its logger)
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        open(os.path.join(ROOT, "gpurun_out", "ui_bad_value.log"), "w").write(log + "\nDRAINED %r\n" % drained)
    assert proc.returncode == 0, log[-3000:]
    assert log.count("rejected") >= 9 and "Rendering stopped by remote UI" in log
    assert log.count("Completed render step 1/1000000") >= 2     # first run + the restart for fov 60


@pytest.mark.gpu
def test_cli_on_the_reference_assets_directory(host, tmp_path):
    """`--assets` pointed at the reference's own assets.extra (only nif_metadata.txt is shipped: the trained weights are
    absent, so `--synthetic-nif` supplies seeded stand-ins of the architecture the metadata names: 6 x 320, embedding 12)."""
    exe = os.path.join(HOST, "ipu_trace")
    out = tmp_path / "alley.png"
    r = subprocess.run([exe, "--assets", os.path.dirname(REAL_META), "--synthetic-nif", "-w", "48", "-h", "32", "-s", "4",
                        "--samples-per-step", "2", "-o", str(out), "--log-level", "debug"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "NIF embedding dimension: 12" in r.stdout and "NIF hidden dimension: 320" in r.stdout
    assert "model FLOPS: %d" % (1089283 * 48 * 32) in r.stdout           # NifModel::analyseModel's formula on 6 x 320
    assert os.path.getsize(out) > 300 and os.path.getsize(tmp_path / "alley.exr") > 48 * 32 * 3 * 2
    # without stand-in weights the run fails the way the reference does when converted.hdf5 is missing
    r = subprocess.run([exe, "--assets", os.path.dirname(REAL_META), "-w", "48", "-h", "32", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "Could not load NIF model" in r.stdout


@pytest.mark.gpu
def test_ui_client_script_drives_a_session(host, tmp_path):
    """scripts/ui_client.py is the usable end of the text protocol (the reference's remote UI speaks packetcomms, absent
    here): one session of actions against a running `ipu_trace --ui-port` -- change the field of view (a restart), listen,
    save the latest preview as PPM and the latest complete HDR image as PFM, stop."""
    import socket
    exe = os.path.join(HOST, "ipu_trace")
    assets = tmp_path / "assets.extra"
    assets.mkdir()
    nif_assets.write_metadata(str(assets / "nif_metadata.txt"))
    nif_assets.write_ptnif(str(assets / "converted.ptnif"), nif_assets.synthetic_nif(), 12)
    W, H = 64, 48
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    proc = subprocess.Popen([exe, "--assets", str(assets), "-w", str(W), "-h", str(H), "-s", "40000000", "--samples-per-step", "20",
                             "--interactive-samples", "2", "--max-path-length", "5", "-o", str(tmp_path / "ui.png"), "--save-interval", "3",
                             "--ui-port", str(port)], stdout=open(str(tmp_path / "cli.log"), "w"), stderr=subprocess.STDOUT, text=True)
    try:
        c = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ui_client.py"), "--port", str(port), "--quiet", "--connect-timeout", "120",
                            "fov=60", "wait=4", "save_preview=" + str(tmp_path / "p.ppm"), "save_hdr=" + str(tmp_path / "p.pfm"), "stop"],
                           capture_output=True, text=True, timeout=300)
        assert c.returncode == 0, c.stdout[-2000:] + c.stderr[-2000:]
        assert proc.wait(timeout=120) == 0
    finally:
        if proc.poll() is None:
            proc.kill()
    log = open(str(tmp_path / "cli.log")).read()
    assert "Rendering stopped by remote UI" in log
    ppm = open(str(tmp_path / "p.ppm"), "rb").read()
    assert ppm.startswith(b"P6\n%d %d\n255\n" % (W, H)) and len(ppm) == len(b"P6\n%d %d\n255\n" % (W, H)) + W * H * 3
    pfm = open(str(tmp_path / "p.pfm"), "rb").read()
    head = b"PF\n%d %d\n-1.0\n" % (W, H)
    assert pfm.startswith(head) and len(pfm) == len(head) + W * H * 12
    img = np.frombuffer(pfm[len(head):], dtype="<f4")
    assert np.isfinite(img).all() and img.max() > 0


@pytest.mark.gpu
def test_cli_multi_device_loop_on_one_gpu(host, tmp_path):
    """The whole `--ipus N` step loop on a ONE-GPU box: N logical devices mapped onto GPU 0 (`--devices 0,0,0`), each with its
    own handle, streams, worklist slice and thread (PathTracerApp::onEveryDevice), film resident per device, HDR tiles
    gathered through the host (RCCL needs one GPU per rank: that exchange alone stays for a multi-GPU box -- DESIGN.md 6).
    The film must equal the one-device film bit for bit (RNG keyed by pixel and sample index, per-pixel sums in step order),
    also when the devices trade image tiles by measured path length at the save intervals (N3: pt_tile_costs, tile dealing,
    pt_film_seed across devices), and in the reference's own loop (--host-film: read_results from every device)."""
    one, _ = _run_cli(host, tmp_path, "one", ["--ipus", "1"], steps=5)
    three, log = _run_cli(host, tmp_path, "three", ["--ipus", "3", "--devices", "0,0,0"], steps=5)
    assert "share GPU 0" in log and "gathered through the host" in log and "film resident on the devices" in log
    assert three.tobytes() == one.tobytes()
    two_lb, log = _run_cli(host, tmp_path, "two_lb", ["--ipus", "2", "--devices", "0,0", "--enable-load-balancing", "--log-level", "debug"], steps=5)
    assert log.count("Load balancing finished") == 2 and "image tiles over 2 devices" in log
    assert two_lb.tobytes() == one.tobytes()
    two_host, log = _run_cli(host, tmp_path, "two_host", ["--ipus", "2", "--devices", "0,0", "--host-film"], steps=5)
    assert "the reference's" in log
    assert two_host.tobytes() == one.tobytes()
    # BASELINE config C4's world size: 8 devices (8 handles, 8 host threads, 8 worklist slices and resident films, an 8-way
    # tile deal by measured path length at every save interval, 8 HDR tiles per gather) -- on one GPU, one process
    zeros8 = ",".join(["0"] * 8)
    one_big, _ = _run_cli(host, tmp_path, "one_big", ["--ipus", "1"], W=176, H=144, steps=5)
    eight, log = _run_cli(host, tmp_path, "eight", ["--ipus", "8", "--devices", zeros8], W=176, H=144, steps=5)
    assert "HDR tiles of 8 devices are gathered through the host" in log
    assert eight.tobytes() == one_big.tobytes()
    eight_lb, log = _run_cli(host, tmp_path, "eight_lb", ["--ipus", "8", "--devices", zeros8, "--enable-load-balancing"], W=176, H=144, steps=5)
    assert log.count("Load balancing finished") == 2 and "99 image tiles over 8 devices" in log
    assert eight_lb.tobytes() == one_big.tobytes()
    # a list that does not match --ipus, or is not a list of ordinals, is refused before anything is attached
    exe = os.path.join(HOST, "ipu_trace")
    for bad in ("0", "0,x", "0,-1"):
        r = subprocess.run([exe, "--assets", str(tmp_path / "assets.extra"), "-w", "32", "-h", "32", "-s", "2", "--samples-per-step", "2",
                            "-o", str(tmp_path / "bad.png"), "--ipus", "2", "--devices", bad], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "--devices" in r.stdout, r.stdout[-500:]


@pytest.mark.gpu
def test_cli_two_devices_or_a_clean_refusal(host, tmp_path):
    """`--ipus 2`: on a box with two or more GPUs the film must equal the one-GPU film bit for bit (worklist slices per
    device, RNG keyed by pixel and sample index, resident film gathered by pt_gather_hdr over an RCCL communicator of two
    ranks made by pt_comm_init_all); on a one-GPU box the run must refuse cleanly instead of crashing."""
    import torch
    one, _ = _run_cli(host, tmp_path, "one_gpu", ["--ipus", "1"])
    if torch.cuda.device_count() >= 2:
        two, log = _run_cli(host, tmp_path, "two_gpus", ["--ipus", "2"])
        assert "RCCL communicator over 2 devices" in log
        assert two.tobytes() == one.tobytes()
    else:
        exe = os.path.join(HOST, "ipu_trace")
        r = subprocess.run([exe, "--assets", str(tmp_path / "assets.extra"), "-w", "96", "-h", "80", "-s", "8", "--samples-per-step", "4",
                            "-o", str(tmp_path / "x.png"), "--ipus", "2"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "Could not attach to device" in r.stdout and "out of range" in r.stdout
