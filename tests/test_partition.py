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


def test_deal_by_path_length_balances_and_keeps_tile_counts():
    """N3 (LoadBalancer.cpp:141-192 across ranks): equal tile counts, near-equal cost, identical on every rank."""
    rng = np.random.default_rng(3)
    world = 8
    cost = rng.integers(256, 256 * 8, size=4347).astype(np.float64)
    cost[::world] *= 3.0            # adversarial for round-robin: every 8th tile is heavy -> rank 0 gets them all
    rr = np.arange(cost.size) % world
    owner = partition.deal_by_path_length(cost, world)
    counts = np.bincount(owner, minlength=world)
    assert counts.max() - counts.min() <= 1
    load = np.bincount(owner, weights=cost, minlength=world)
    load_rr = np.bincount(rr, weights=cost, minlength=world)
    assert load.max() / load.mean() < 1.01
    assert load_rr.max() / load_rr.mean() > 1.5
    assert np.array_equal(owner, partition.deal_by_path_length(cost.copy(), world))
    # two tiles per rank: exactly the reference's shortest+longest pairing
    c = np.array([5.0, 1.0, 9.0, 3.0])
    o = partition.deal_by_path_length(c, 2)
    assert o[2] == o[1] and o[0] == o[3] and o[2] != o[0]


def test_redealt_worklists_cover_every_pixel_once_and_reassemble():
    w, h, world = 200, 120, 3
    cost = np.random.default_rng(5).random(partition.tile_grid(w, h)[0] * partition.tile_grid(w, h)[1])
    owner = partition.deal_by_path_length(cost, world)
    seen = np.zeros((h, w), dtype=np.int32)
    truth = np.random.default_rng(6).random((h, w, 3)).astype(np.float32)
    parts = []
    for r in range(world):
        rec = partition.worklist_for_owner(w, h, owner, r)
        assert rec.size <= partition.max_items_per_rank(w, h, world)
        np.add.at(seen, (rec["v"], rec["u"]), 1)
        parts.append(truth[rec["v"], rec["u"], :])
    assert np.all(seen == 1)
    assert np.array_equal(partition.assemble_hdr(w, h, world, parts, owner=owner), truth)
    # tile_costs is the inverse bookkeeping: per-tile sums of what a step returned
    rec = partition.worklist_for_owner(w, h, owner, 1)
    rec["pathLength"] = 2
    tc = partition.tile_costs(rec, w, h)
    assert tc.sum() == 2 * rec.size and np.all(tc[owner != 1] == 0)


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
# re-deal between save intervals: per-tile path-length sums are all-reduced, every rank derives the same deal
rec["pathLength"] = 1 + (rec["u"] // 16 + rec["v"] // 16) % 5
cost = torch.from_numpy(partition.tile_costs(rec, W, H))
dist.all_reduce(cost, op=dist.ReduceOp.SUM)
owner = partition.deal_by_path_length(cost.numpy(), world)
both = [torch.empty(owner.size, dtype=torch.int32) for _ in range(world)]
dist.all_gather(both, torch.from_numpy(owner))
assert all(torch.equal(b, both[0]) for b in both)
rec2 = partition.worklist_for_owner(W, H, owner, rank)
cap = partition.max_items_per_rank(W, H, world)
vals2 = np.stack([rec2["u"] * 1.0, rec2["v"] * 2.0, rec2["u"] * 0.5 + rec2["v"]], -1).astype(np.float32)
hdr2 = torch.zeros((cap, 3), dtype=torch.float32)
hdr2[: rec2.size] = torch.from_numpy(vals2)
gathered2 = [torch.empty_like(hdr2) for _ in range(world)] if rank == 0 else None
dist.gather(hdr2, gathered2, dst=0)
if rank == 0:
    film2 = partition.assemble_hdr(W, H, world, [g.numpy() for g in gathered2], owner=owner)
    assert np.array_equal(film2, exp)
    print("REDEAL_OK")
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 8])   # 8 = BASELINE config C4's world size (8 CPU processes, gloo)
def test_hdr_gather_ranks_gloo(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script), ROOT]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "GATHER_OK" in p.stdout and "REDEAL_OK" in p.stdout
