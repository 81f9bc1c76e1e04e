#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle.

The reference holds no golden vectors for this path, so these fixtures are
outputs of the build's own oracle at a fixed seed: they pin the oracle (and through it the HIP path)
against regressions; they are NOT outputs of the reference.  Run: python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ipu_path_trace_amd import nif_assets  # noqa: E402
from oracle import pt_oracle as O  # noqa: E402


def paths_fixture():
    W, H = 1104, 1000
    rng = np.random.default_rng(20240901)
    n = 512
    u = rng.integers(0, W, n).astype(np.uint16)
    v = rng.integers(H // 3, H, n).astype(np.uint16)
    s = rng.integers(0, 30000, n).astype(np.uint32)
    cfg = O.make_config(width=W, height=H, max_path_length=8, seed=1, env_rotation_degrees=15.0)
    P = [O.trace_path(cfg, int(a), int(b), int(c)) for a, b, c in zip(u, v, s)]
    np.savez_compressed(
        os.path.join(HERE, "paths_1104x1000_d8.npz"), u=u, v=v, sample=s,
        length=np.array([p.length for p in P], np.uint32), escaped=np.array([p.escaped for p in P], np.uint32),
        dir=np.array([list(p.dir) for p in P], np.float32), uv=np.array([list(p.uv) for p in P], np.float32),
        throughput=np.array([list(p.throughput) for p in P], np.float32),
        cam=np.array([list(p.cam) for p in P], np.float32))


def c1_fixture():
    """BASELINE config C1: 256x256, 16 spp, depth 4, constant sky, seed 1, reference fold order."""
    W = H = 256
    out = {}
    for name, fold in (("backward", O.FOLD_BACKWARD), ("forward", O.FOLD_FORWARD)):
        cfg = O.make_config(width=W, height=H, max_path_length=4, env_rgb=(1, 1, 1), fold=fold)
        rec = O.worklist(W, H)
        O.render(cfg, None, rec, 0, 16)
        img = np.stack([rec["r"], rec["g"], rec["b"]], -1).reshape(H, W, 3) / 16.0
        out["sha256_" + name] = np.frombuffer(hashlib.sha256(rec.tobytes()).digest(), dtype=np.uint8)
        out["block_mean_" + name] = img.reshape(32, 8, 32, 8, 3).mean(axis=(1, 3)).astype(np.float32)
        out["path_length_sum_" + name] = np.array([int(rec["pathLength"].astype(np.int64).sum())])
    np.savez_compressed(os.path.join(HERE, "c1_256x256_16spp_d4.npz"), **out)


def nif_fixture():
    layers = nif_assets.synthetic_nif()
    nif = O.Nif(layers, 12, nif_assets.URBAN_ALLEY_META["max"], nif_assets.folded_mean())
    rng = np.random.default_rng(99)
    u = rng.random(256, dtype=np.float32)
    v = rng.random(256, dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "nif_6x320_seed2024.npz"), u=u, v=v, bgr=nif.infer(u, v),
                        feats=np.stack([O.nif_encode(12, a, b) for a, b in zip(u[:16], v[:16])]),
                        w0_sha=np.frombuffer(hashlib.sha256(layers[0][0].tobytes()).digest(), dtype=np.uint8))


if __name__ == "__main__":
    paths_fixture()
    c1_fixture()
    nif_fixture()
    print("golden fixtures written to", HERE)
