"""Row-band sharding of a frame across ranks + the one collective of the path (gather to rank 0).

The reference is single-GPU (no collectives anywhere); BASELINE.json's north_star adds
row tiling across the GPUs of a node with a gather as the only exchange step.  Pixels are independent, so
each rank renders its own rows with no communication; ownership is BAND-CYCLIC (band b of `band_rows` rows
belongs to rank b % world) because contiguous 1/N bands are badly imbalanced on real scenes (SURVEY.md 8(e):
21x spread of hit pixels on 20spheres, vs +-0.6 % cyclic).

This module is backend-agnostic host logic (torch.distributed: "nccl" = RCCL on the GPU box, "gloo" in the
CPU tests); the device-side reassembly is rt_assemble() in libmi355rt.so.
"""
import numpy as np


def band_rows_of_rank(height, band_rows, world, rank):
    rows = []
    n_bands = (height + band_rows - 1) // band_rows
    for b in range(rank, n_bands, world):
        rows.extend(range(b * band_rows, min(height, (b + 1) * band_rows)))
    return np.asarray(rows, dtype=np.uint32)


def max_local_rows(height, band_rows, world):
    return max(len(band_rows_of_rank(height, band_rows, world, r)) for r in range(world))


def assemble_index(height, band_rows, world):
    """For every image row y: its position in the rank-major gathered buffer [world * max_local_rows]."""
    mx = max_local_rows(height, band_rows, world)
    idx = np.empty(height, dtype=np.int64)
    for r in range(world):
        rows = band_rows_of_rank(height, band_rows, world, r)
        idx[rows] = r * mx + np.arange(len(rows))
    return idx


def gather_to_root(local, world, rank, group=None, gathered=None):
    """dist.gather of each rank's [max_local_rows, W, C] buffer into rank 0's [world, max_local_rows, W, C]."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local.unsqueeze(0)
    if rank == 0:
        if gathered is None:
            gathered = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, list(gathered.unbind(0)), dst=0, group=group)
        return gathered
    dist.gather(local, None, dst=0, group=group)
    return None


def assemble_torch(gathered, height, band_rows, world):
    """Reference reassembly with torch indexing (used by the CPU tests and to cross-check rt_assemble)."""
    import torch
    idx = torch.from_numpy(assemble_index(height, band_rows, world)).to(gathered.device)
    flat = gathered.reshape((gathered.shape[0] * gathered.shape[1],) + tuple(gathered.shape[2:]))
    return flat.index_select(0, idx)


# ---- sparse transport (RGBA8): host-side mirror of rt_pack_sparse / rt_assemble_sparse (include/mi355rt.h) ----------
# message = uint32 {count, overflow, 0, 0}, uint32 ids[capacity] (padded to 16 bytes), capacity x 256 RGBA8 pixels
# (tile-major, 16 rows of 16 pixels).  Used by the CPU (gloo) tests and to cross-check the device kernels.

def sparse_words(capacity):
    return ((4 + capacity + 3) & ~3) + capacity * 256


def bg_rgba8(bg_color):
    r, g, b = (int(np.float32(c) * np.float32(255.0) + np.float32(0.5)) & 0xFF for c in bg_color)
    return np.uint32(r | (g << 8) | (b << 16) | (255 << 24))


def pack_sparse_numpy(local_rgba8, local_rows, bg_word, capacity):
    """local_rgba8: [>= local_rows, W, 4] uint8 rows of one rank.  Tiles in index order (the device appends in any order)."""
    w = local_rgba8.shape[1]
    px = np.ascontiguousarray(local_rgba8[:local_rows]).view(np.uint32).reshape(local_rows, w)
    tiles_x, tiles_y = (w + 15) // 16, (local_rows + 15) // 16
    msg = np.zeros(sparse_words(capacity), dtype=np.uint32)
    off = (4 + capacity + 3) & ~3
    count = 0
    for t in range(tiles_x * tiles_y):
        tx, ty = t % tiles_x, t // tiles_x
        tile = np.full((16, 16), bg_word, dtype=np.uint32)
        blk = px[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
        tile[:blk.shape[0], :blk.shape[1]] = blk
        if np.any(tile != bg_word):
            if count < capacity:
                msg[4 + count] = t
                msg[off + count * 256: off + (count + 1) * 256] = tile.reshape(-1)
            else:
                msg[1] = 1
            count += 1
    msg[0] = count
    return msg


def assemble_sparse_numpy(msgs, width, height, band_rows, world, bg_word, capacity):
    """msgs: [world, sparse_words(capacity)] uint32 in rank order -> [height, width, 4] uint8."""
    full = np.full((height, width), bg_word, dtype=np.uint32)
    tiles_x = (width + 15) // 16
    off = (4 + capacity + 3) & ~3
    for r in range(world):
        rows = band_rows_of_rank(height, band_rows, world, r)
        for j in range(min(int(msgs[r][0]), capacity)):
            t = int(msgs[r][4 + j])
            tile = msgs[r][off + j * 256: off + (j + 1) * 256].reshape(16, 16)
            tx, ty = t % tiles_x, t // tiles_x
            for k in range(16):
                lr = ty * 16 + k
                if lr < len(rows):
                    x0 = tx * 16
                    n = min(16, width - x0)
                    full[rows[lr], x0:x0 + n] = tile[k, :n]
    return full.view(np.uint8).reshape(height, width, 4)
