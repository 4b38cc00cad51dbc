"""The N > 1 path on CPU: band-cyclic row ownership, the gather to rank 0 (torch.distributed, gloo here / RCCL on
the GPU box) and the reassembly, with the ORACLE standing in for the renderer (the product has no CPU path).
The gathered frame must equal the single-rank frame bit for bit."""
import os
import socket

import numpy as np
import pytest

from conftest import ROOT, scene_path


def test_band_ownership_partitions_the_rows(pkg):
    for h, band, world in ((1080, 16, 8), (250, 8, 3), (7, 8, 2), (4320, 16, 8), (33, 4, 5)):
        seen = np.concatenate([pkg.band_rows_of_rank(h, band, world, r) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(h))
        idx = pkg.assemble_index(h, band, world)
        mx = pkg.max_local_rows(h, band, world)
        assert len(set(idx.tolist())) == h and idx.max() < world * mx
    # cyclic bands balance the hit-heavy rows of 20spheres: every rank gets 1/8 of the rows +- one band
    sizes = [len(pkg.band_rows_of_rank(1080, 16, 8, r)) for r in range(8)]
    assert max(sizes) - min(sizes) <= 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, h, w, band, out_path):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg, O = graft.load_package(), graft.load_oracle()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sc = O.load_scene(scene_path("20spheres")).with_size(w, h)
    rows = pkg.band_rows_of_rank(h, band, world, rank)
    mx = pkg.max_local_rows(h, band, world)
    local = torch.zeros((mx, w, 3), dtype=torch.float32)
    if len(rows):
        local[: len(rows)] = torch.from_numpy(sc.render(rows=rows))
    gathered = pkg.gather_to_root(local, world, rank)
    if rank == 0:
        full = pkg.assemble_torch(gathered, h, band, world)
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,band", [(2, 60, 8), (3, 50, 4)])
def test_gloo_gather_reassembles_the_frame(oracle, tmp_path, world, h, band):
    import torch.multiprocessing as mp
    w = 80
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), h, w, band, out), nprocs=world, join=True)
    want = oracle.load_scene(scene_path("20spheres")).with_size(w, h).render()
    assert np.array_equal(np.load(out), want)


def _sparse_worker(rank, world, port, h, w, band, cap, out_path):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg, O = graft.load_package(), graft.load_oracle()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sc = O.load_scene(scene_path("20spheres")).with_size(w, h)
    rows = pkg.band_rows_of_rank(h, band, world, rank)
    img = sc.render(rows=rows) if len(rows) else np.zeros((0, w, 3), dtype=np.float32)
    rgba8 = np.concatenate([np.floor(img * 255.0 + 0.5).astype(np.uint8), np.full(img.shape[:2] + (1,), 255, np.uint8)], axis=-1)
    bgw = pkg.bg_rgba8(sc.bg_color)
    msg = torch.from_numpy(pkg.pack_sparse_numpy(rgba8, len(rows), bgw, cap).view(np.int32).copy())
    gathered = pkg.gather_to_root(msg, world, rank)      # the same collective, a fixed-size message instead of the rows
    if rank == 0:
        full = pkg.assemble_sparse_numpy(gathered.numpy().view(np.uint32), w, h, band, world, bgw, cap)
        np.save(out_path, full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,band", [(2, 60, 8), (3, 50, 5)])
def test_gloo_sparse_gather_reassembles_the_frame(oracle, tmp_path, world, h, band):
    """The sparse transport (tiles with content + ids in fixed-size messages) through the same gather: the rebuilt RGBA8
    frame equals the quantised oracle frame; band heights that cut through 16-row tiles included."""
    import torch.multiprocessing as mp
    w = 90
    out = str(tmp_path / "full.npy")
    cap = ((w + 15) // 16) * ((h + 15) // 16)
    mp.spawn(_sparse_worker, args=(world, _free_port(), h, w, band, cap, out), nprocs=world, join=True)
    sc = oracle.load_scene(scene_path("20spheres")).with_size(w, h)
    want = np.floor(sc.render() * 255.0 + 0.5).astype(np.uint8)
    got = np.load(out)
    assert np.array_equal(got[..., :3], want) and np.all(got[..., 3] == 255)
