"""Image partition across GPUs and HDR-tile reassembly (multi-GPU row of SURVEY.md section 8(e)).

The reference splits the (shuffled) worklist over IPUs and merges results only in the host film
(src/PathTracerApp.cpp:205-252, src/AccumulatedImage.cpp:59-74).  Here the image is cut into small
square tiles dealt round-robin to the ranks, so sky-heavy and sphere-heavy regions mix on every GPU
while neighbouring lanes still trace neighbouring pixels; the RNG is keyed by pixel coordinate, so
the image does not depend on the number of ranks.
"""
import numpy as np

from .ptmi import TRACE_DTYPE

TILE = 16


def tile_grid(width, height, tile=TILE):
    """(tiles across, tiles down)."""
    return (width + tile - 1) // tile, (height + tile - 1) // tile


def round_robin_owner(width, height, world, tile=TILE):
    """owner[t] = rank that traces tile t: the static deal used until path lengths are known."""
    tx, ty = tile_grid(width, height, tile)
    return (np.arange(tx * ty) % world).astype(np.int32)


def deal_by_path_length(cost, world):
    """Re-deal tiles so every rank gets the same number of tiles (+-1) and nearly the same total path length.

    The reference balances by sorting work items by path length and pairing the shortest with the longest inside
    each IPU tile (LoadBalancer::allocateWorkByPathLength, src/LoadBalancer.cpp:141-192).  The same idea for W
    ranks: sort the image tiles by their measured cost and deal them in boustrophedon order (0..W-1, W-1..0, ...),
    which for two tiles per rank IS the shortest+longest pairing.  Deterministic (stable sort, ties by tile id),
    so every rank derives the same deal from the same all-reduced costs.
    """
    cost = np.asarray(cost, dtype=np.float64)
    order = np.argsort(-cost, kind="stable")
    pos = np.arange(order.size)
    rnd, idx = np.divmod(pos, world)
    owner = np.empty(order.size, dtype=np.int32)
    owner[order] = np.where(rnd % 2 == 0, idx, world - 1 - idx)
    return owner


def worklist_for_owner(width, height, owner, rank, tile=TILE):
    """Work items (TraceRecord) of `rank` under the deal `owner`: its tiles in tile order, pixels row-major inside."""
    tx, _ = tile_grid(width, height, tile)
    mine = np.flatnonzero(np.asarray(owner) == rank)
    t_r, t_c = np.divmod(mine, tx)
    dy, dx = np.divmod(np.arange(tile * tile), tile)
    cols = (t_c[:, None] * tile + dx[None, :]).ravel()
    rows = (t_r[:, None] * tile + dy[None, :]).ravel()
    keep = (cols < width) & (rows < height)
    rec = np.zeros(int(keep.sum()), dtype=TRACE_DTYPE)
    rec["u"] = cols[keep]
    rec["v"] = rows[keep]
    return rec


def max_items_per_rank(width, height, world, tile=TILE):
    """Capacity that fits any deal with equal tile counts (+-1): pt_config.max_work_items for a re-dealing job."""
    tx, ty = tile_grid(width, height, tile)
    return -(-(tx * ty) // world) * tile * tile


def tile_costs(rec, width, height, tile=TILE):
    """Per-tile sum of the path lengths a step returned in `rec` (zeros for tiles this rank does not own)."""
    tx, ty = tile_grid(width, height, tile)
    t = (rec["v"].astype(np.int64) // tile) * tx + rec["u"].astype(np.int64) // tile
    return np.bincount(t, weights=rec["pathLength"].astype(np.float64), minlength=tx * ty)


def tile_order_worklist(width, height, rank=0, world=1, tile=TILE):
    """Work items (TraceRecord) of `rank`: tiles t with t % world == rank, pixels row-major inside a tile."""
    return worklist_for_owner(width, height, round_robin_owner(width, height, world, tile), rank, tile)


def items_per_rank(width, height, world, tile=TILE):
    return [tile_order_worklist(width, height, r, world, tile).size for r in range(world)]


def assemble_hdr(width, height, world, gathered, tile=TILE, owner=None):
    """gathered[r]: float32 [>= n_r, 3] BGR means of rank r's items (padded rows ignored) -> H x W x 3 BGR film."""
    if owner is None:
        owner = round_robin_owner(width, height, world, tile)
    film = np.zeros((height, width, 3), dtype=np.float32)
    for r in range(world):
        rec = worklist_for_owner(width, height, owner, r, tile)
        film[rec["v"], rec["u"], :] = np.asarray(gathered[r])[: rec.size]
    return film
