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


def tile_order_worklist(width, height, rank=0, world=1, tile=TILE):
    """Work items (TraceRecord) of `rank`: tiles t with t % world == rank, pixels row-major inside a tile."""
    tx = (width + tile - 1) // tile
    ty = (height + tile - 1) // tile
    tiles = np.arange(tx * ty)
    mine = tiles[tiles % world == rank]
    t_r, t_c = np.divmod(mine, tx)
    dy, dx = np.divmod(np.arange(tile * tile), tile)
    cols = (t_c[:, None] * tile + dx[None, :]).ravel()
    rows = (t_r[:, None] * tile + dy[None, :]).ravel()
    keep = (cols < width) & (rows < height)
    rec = np.zeros(int(keep.sum()), dtype=TRACE_DTYPE)
    rec["u"] = cols[keep]
    rec["v"] = rows[keep]
    return rec


def items_per_rank(width, height, world, tile=TILE):
    return [tile_order_worklist(width, height, r, world, tile).size for r in range(world)]


def assemble_hdr(width, height, world, gathered, tile=TILE):
    """gathered[r]: float32 [>= n_r, 3] BGR means of rank r's items (padded rows ignored) -> H x W x 3 BGR film."""
    film = np.zeros((height, width, 3), dtype=np.float32)
    for r in range(world):
        rec = tile_order_worklist(width, height, r, world, tile)
        film[rec["v"], rec["u"], :] = np.asarray(gathered[r])[: rec.size]
    return film
