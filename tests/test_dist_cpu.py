"""The N > 1 path on CPU: world_size-2 and -3 `gloo` process groups run the same partition -> gather -> untile logic
bench.py uses on GPUs (lens_trace_amd/dist.py), with the CPU oracle standing in for the per-rank renderer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lens_trace_amd import scene as sc
from lens_trace_amd.dist import TilePlan, gather_to_root, tile_stack_numpy, untile_numpy
from tests.conftest import GOLDEN


def test_balanced_plan_spreads_columns_and_rows_over_all_ranks():
    """TilePlan.balanced: tiles per row coprime to the world size, so that the tiles of any one tile column (and of any one
    tile row) are dealt to the ranks as evenly as their number allows."""
    from math import gcd
    for W, H, world in ((3840, 2160, 8), (3840, 2160, 2), (3840, 2160, 4), (1920, 1080, 8), (1920, 1080, 3), (640, 480, 6), (100, 70, 1)):
        plan = TilePlan.balanced(W, H, 3, world)
        assert plan.tile_h == 64 and plan.tile_w % 8 == 0 and 8 <= plan.tile_w <= 64
        best = min(gcd((W + w - 1) // w, world) for w in range(8, 65, 8))
        assert gcd(plan.tiles_x, world) == best
        if best == 1:
            owner = np.arange(plan.n_tiles).reshape(plan.tiles_y, plan.tiles_x) % world
            for column in owner.T:
                counts = np.bincount(column, minlength=world)
                assert counts.max() - counts.min() <= 1
            for row in owner:
                counts = np.bincount(row, minlength=world)
                assert counts.max() - counts.min() <= 1
        # and it is still a partition of the image
        img = np.random.default_rng(1).random((H, W, 3), dtype=np.float32)
        assert np.array_equal(untile_numpy(plan, [tile_stack_numpy(plan, r, img) for r in range(world)]), img)


def test_tile_plan_covers_every_pixel_once():
    for (W, H, tw, th, n) in [(3840, 2160, 64, 64, 8), (200, 120, 64, 64, 3), (100, 70, 100, 16, 2), (17, 5, 8, 8, 4)]:
        plan = TilePlan(W, H, 3, tw, th, n)
        cover = np.zeros((H, W), dtype=np.int32)
        for r in range(n):
            for t in plan.tiles_of(r):
                x0, y0, w, h = plan.tile_rect(t)
                cover[y0:y0 + h, x0:x0 + w] += 1
        assert (cover == 1).all()
        assert plan.floats_per_rank * n >= W * H * 3
    plan = TilePlan(3840, 2160, 3, 64, 64, 8)
    assert plan.n_tiles == 60 * 34 and len(plan.tiles_of(3)) == 255 and plan.floats_per_rank == 255 * 64 * 64 * 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, tile, frame, ret):
    from oracle import pyoracle as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = sc.load_ltsb(os.path.join(GOLDEN, "cornell_box_O0.ltsb"))
        cam = sc.camera_bytes(0.0, 2.5, -50.0, 0.0, 0.0, 0.0, frame)
        plan = TilePlan(W, H, 3, tile[0], tile[1], world)
        # each rank renders ONLY its own tiles (row ranges of the oracle), like a GPU rank would
        img = np.zeros((H, W, 3), dtype=np.float32)
        for t in plan.tiles_of(rank):
            x0, y0, w, h = plan.tile_rect(t)
            band = po.render(s, cam, W, H, po.ACCUMULATOR, rows=(y0, y0 + h))
            img[y0:y0 + h, x0:x0 + w] = band[y0:y0 + h, x0:x0 + w]
        local = torch.from_numpy(tile_stack_numpy(plan, rank, img))
        stacks = gather_to_root(local, world, rank)
        if rank == 0:
            full = untile_numpy(plan, [t.numpy() for t in stacks])
            want = po.render(s, cam, W, H, po.ACCUMULATOR)
            ret["equal"] = bool(np.array_equal(full, want))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,tile", [(2, 96, 64, (32, 16)), (3, 70, 50, (16, 16))])
def test_gloo_gather_reassembles_frame(world, W, H, tile):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), W, H, tile, 2, ret), nprocs=world, join=True)
    assert ret["equal"] is True
