"""NIF asset helpers for tests and bench (numpy only; no device code).

* `load_metadata` mirrors src/neural_networks/NifMetaData.cpp:11-71 (JSON fields, the -eps fold
  into mean at :48-53, hidden size from `train_command` at :56-65).
* `synthetic_nif` builds seeded stand-in weights: the trained weights
  (nif_models/.../converted.hdf5) are missing from the reference checkout (.MISSING_LARGE_BLOBS)
  and there is no HDF5 library here, so benchmarks and parity tests run on random-initialised
  weights of the same architecture.
"""
import json

import numpy as np

# nif_models/urban_alley_01_4k_fp16_yuv/assets.extra/nif_metadata.txt:2-12,35-38
URBAN_ALLEY_META = {
    "embedding_dimension": 12,
    "hidden_size": 320,
    "layer_count": 6,
    "eps": 1e-08,
    "log_tone_map": True,
    "max": 3.4299468994140625,
    "mean": [-2.3514461517333984, -2.2660605907440186, -1.9648972749710083],
    "original_image_shape": [2048, 4096, 3],
}


def load_metadata(path):
    with open(path) as f:
        pt = json.load(f)
    enc = pt["encode_params"]
    mean = [np.float32(m) for m in enc["mean"]]
    eps = np.float32(enc["eps"])
    log_tone_map = bool(enc["log_tone_map"])
    if log_tone_map:  # NifMetaData.cpp:48-53
        mean = [np.float32(m - eps) for m in mean]
    hidden = layers = None
    cmd = pt.get("train_command", [])
    for i, tok in enumerate(cmd):
        if tok == "--layer-size":
            hidden = int(cmd[i + 1])
        if tok == "--layer-count":
            layers = int(cmd[i + 1])
    return {
        "name": pt["name"],
        "embedding_dimension": int(pt["embedding_dimension"]),
        "hidden_size": hidden,
        "layer_count": layers,
        "original_image_shape": [int(x) for x in pt["original_image_shape"]],
        "eps": float(eps),
        "log_tone_map": log_tone_map,
        "max": float(np.float32(enc["max"])),
        "mean_folded": [float(m) for m in mean],
    }


def folded_mean(meta=URBAN_ALLEY_META):
    eps = np.float32(meta["eps"])
    return [float(np.float32(np.float32(m) - eps)) for m in meta["mean"]]


def synthetic_nif(hidden=320, layer_count=6, embedding_dim=12, seed=2024, bias_scale=0.05, dtype=np.float16,
                  widths=None, skips=None):
    """Return [(kernel [in,out], bias [out], relu)], Keras-Dense layout (NifModel.cpp:375-401).

    Architecture as inferred in SURVEY.md row A9: `layer_count` hidden ReLU layers of width `hidden`,
    the Fourier-feature input re-concatenated at the middle layer (the shape mismatch that
    NifModel.cpp:305-308 detects), and a linear 3-channel head.  `widths` gives every hidden layer its own
    width and `skips` the set of layers that take concat(x, input) instead (the reference builds whatever
    stack the H5 describes, NifModel.cpp:295-326); `dtype` float32 yields a float32-variable model
    (Hdf5Model.cpp:109-133 accepts both).
    """
    rng = np.random.Generator(np.random.Philox(seed))
    in_dim = 4 * embedding_dim
    if widths is None:
        widths = [hidden] * layer_count
    if skips is None:
        skips = {len(widths) // 2}
    layers = []
    fan_in = in_dim
    for l, width in enumerate(widths):
        if l in skips and l > 0:
            fan_in += in_dim
        k = rng.standard_normal((fan_in, width), dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
        b = rng.standard_normal(width, dtype=np.float32) * np.float32(bias_scale)
        layers.append((k.astype(dtype), b.astype(dtype), True))
        fan_in = width
    if len(widths) in skips:
        fan_in += in_dim
    k = rng.standard_normal((fan_in, 3), dtype=np.float32) * np.float32(0.25 * np.sqrt(2.0 / fan_in))
    b = rng.standard_normal(3, dtype=np.float32) * np.float32(bias_scale)
    layers.append((k.astype(dtype), b.astype(dtype), False))
    return layers


def flops_per_sample(layers):
    """NifModel::analyseModel (NifModel.cpp:129-133)."""
    f = 0
    for k, b, _ in layers:
        f += 2 * k.shape[0] * k.shape[1]
        if b is not None:
            f += k.shape[1]
    return f


def write_ptnif(path, layers, embedding_dim):
    """Flat side-car weight file read by the C++ host (ipu_path_trace_amd/host/NifModel.cpp: setupModel).

    Stands in for <assets>/converted.hdf5 (reference src/keras/Hdf5Model.cpp:62-87), which needs libhdf5.
    Layout: b"PTNIF1\0\0", u32 n_layers, u32 embedding_dim, then per layer u32 rows, cols, dtype (0 = float16,
    1 = float32), relu, has_bias followed by the raw kernel bytes [rows][cols] and the bias bytes [cols].
    """
    import struct
    with open(path, "wb") as f:
        f.write(b"PTNIF1\0\0")
        f.write(struct.pack("<II", len(layers), embedding_dim))
        for k, b, relu in layers:
            dt = np.float32 if np.asarray(k).dtype == np.float32 else np.float16
            k = np.ascontiguousarray(k, dtype=dt)
            f.write(struct.pack("<IIIII", k.shape[0], k.shape[1], int(dt == np.float32), int(bool(relu)), int(b is not None)))
            f.write(k.tobytes())
            if b is not None:
                f.write(np.ascontiguousarray(b, dtype=dt).tobytes())


def write_metadata(path, meta=URBAN_ALLEY_META):
    """nif_metadata.txt with the fields NifMetaData.cpp reads."""
    doc = {
        "embedding_dimension": meta["embedding_dimension"],
        "encode_params": {"eps": meta["eps"], "log_tone_map": meta["log_tone_map"], "max": meta["max"],
                          "mean": meta["mean"], "transfer_function": "log"},
        "name": "synthetic", "original_image_shape": meta["original_image_shape"],
        "train_command": ["train_nif.py", "--layer-count", str(meta["layer_count"]), "--layer-size",
                          str(meta["hidden_size"]), "--embedding-dimension", str(meta["embedding_dimension"])],
    }
    with open(path, "w") as f:
        json.dump(doc, f, indent=2)
