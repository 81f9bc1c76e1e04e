"""Multi-GPU plumbing on CPU: tile partition, film reassembly and the gather path with gloo (world_size 2)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from ipu_path_trace_amd import partition

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("w,h,world", [(1104, 1000, 1), (1104, 1000, 8), (100, 37, 3), (16, 16, 2), (5, 3, 4)])
def test_tile_partition_covers_every_pixel_once(w, h, world):
    seen = np.zeros((h, w), dtype=np.int32)
    sizes = []
    for r in range(world):
        rec = partition.tile_order_worklist(w, h, r, world)
        sizes.append(rec.size)
        np.add.at(seen, (rec["v"], rec["u"]), 1)
        assert np.all(rec["r"] == 0) and np.all(rec["sampleCount"] == 0)
    assert np.all(seen == 1)
    assert sizes == partition.items_per_rank(w, h, world)
    if w * h >= 256 * world * 8:
        assert max(sizes) - min(sizes) <= 2 * partition.TILE * partition.TILE  # balanced to within two tiles


def test_assemble_roundtrip():
    w, h, world = 70, 50, 3
    truth = np.random.default_rng(0).random((h, w, 3)).astype(np.float32)
    parts = []
    for r in range(world):
        rec = partition.tile_order_worklist(w, h, r, world)
        vals = truth[rec["v"], rec["u"], :]
        parts.append(np.concatenate([vals, np.full((7, 3), -1, np.float32)]))  # padded tail is ignored
    assert np.array_equal(partition.assemble_hdr(w, h, world, parts), truth)


_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ipu_path_trace_amd import partition
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
W, H = 200, 120
rec = partition.tile_order_worklist(W, H, rank, world)
counts = partition.items_per_rank(W, H, world)
# stand-in for pt_export_hdr_device: a film value that depends only on the pixel
vals = np.stack([rec["u"] * 1.0, rec["v"] * 2.0, rec["u"] * 0.5 + rec["v"]], -1).astype(np.float32)
hdr = torch.zeros((max(counts), 3), dtype=torch.float32)
hdr[: rec.size] = torch.from_numpy(vals)
gathered = [torch.empty_like(hdr) for _ in range(world)] if rank == 0 else None
dist.gather(hdr, gathered, dst=0)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == world
if rank == 0:
    film = partition.assemble_hdr(W, H, world, [g.numpy() for g in gathered])
    yy, xx = np.mgrid[0:H, 0:W]
    exp = np.stack([xx * 1.0, yy * 2.0, xx * 0.5 + yy], -1).astype(np.float32)
    assert np.array_equal(film, exp)
    print("GATHER_OK")
dist.destroy_process_group()
"""


def test_hdr_gather_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script), ROOT]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "GATHER_OK" in p.stdout
