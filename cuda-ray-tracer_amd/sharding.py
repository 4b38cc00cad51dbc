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
