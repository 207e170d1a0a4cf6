"""Image-tile sharding of one frame over the GPUs of a node (one process per GPU), new relative to the reference
(which is single-device).  Pixels are independent and the scene is small enough to replicate, so the only
communication is ONE gather of per-rank tile stacks to rank 0 per presented frame; there is no all-reduce.

  rank r renders tiles r, r+N, r+2N, ... of the row-major tile grid (interleaved: per-pixel cost varies a lot over
  the image) into a compact stack [k][tile_h][tile_w][depth];
  torch.distributed.gather (backend "nccl" = RCCL over xGMI; "gloo" on CPU in the tests) brings the stacks to rank 0;
  lt_hip_untile (or untile_numpy on CPU) scatters them into the reference's (y*W+x)*depth layout."""
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class TilePlan:
    width: int
    height: int
    depth: int
    tile_w: int
    tile_h: int
    world: int

    @classmethod
    def balanced(cls, width, height, depth, world, tile=64):
        """Tiles of about tile x tile pixels whose round-robin assignment mixes columns and rows over the ranks.
        Tile t goes to rank t % world = (tx + (tiles_x % world) * ty) % world: when tiles_x shares a factor with world
        (3840 / 64 = 60 tiles over 8 ranks) a tile column lands on world / gcd ranks only, and cost that is concentrated in
        one image column -- the centre column of an unrotated camera, see DESIGN.md -- on as few GPUs.  The width is
        narrowed in steps of 8 pixels (the wavefront square) until tiles_x is coprime to world."""
        from math import gcd
        # (where no width makes it coprime -- 640 pixels over 6 ranks -- the widest tile with the smallest common factor)
        w = min(range(tile, 7, -8), key=lambda w: (gcd((width + w - 1) // w, world), -w))
        return cls(width, height, depth, w, tile, world)

    @property
    def tiles_x(self):
        return (self.width + self.tile_w - 1) // self.tile_w

    @property
    def tiles_y(self):
        return (self.height + self.tile_h - 1) // self.tile_h

    @property
    def n_tiles(self):
        return self.tiles_x * self.tiles_y

    def tiles_of(self, rank):
        return list(range(rank, self.n_tiles, self.world))

    @property
    def floats_per_rank(self):
        """Uniform (padded) stack size so that the gather is regular."""
        return ((self.n_tiles + self.world - 1) // self.world) * self.tile_w * self.tile_h * self.depth

    def desc_tile(self, rank):
        """(tile_w, tile_h, tile_first, tile_stride) for lt_hip_render_desc."""
        return (self.tile_w, self.tile_h, rank, self.world)

    def tile_rect(self, tile):
        tx, ty = tile % self.tiles_x, tile // self.tiles_x
        x0, y0 = tx * self.tile_w, ty * self.tile_h
        return x0, y0, min(self.tile_w, self.width - x0), min(self.tile_h, self.height - y0)


def tile_stack_numpy(plan, rank, image):
    """What a rank's render produces, cut out of a full image (CPU model of the device layout)."""
    out = np.zeros(plan.floats_per_rank, dtype=np.float32)
    view = out.reshape(-1, plan.tile_h, plan.tile_w, plan.depth)
    for k, t in enumerate(plan.tiles_of(rank)):
        x0, y0, w, h = plan.tile_rect(t)
        view[k, :h, :w] = image[y0:y0 + h, x0:x0 + w]
    return out


def untile_numpy(plan, stacks):
    """CPU model of lt_hip_untile: stacks [world, floats_per_rank] -> image [H, W, depth]."""
    img = np.zeros((plan.height, plan.width, plan.depth), dtype=np.float32)
    for r in range(plan.world):
        view = np.asarray(stacks[r]).reshape(-1, plan.tile_h, plan.tile_w, plan.depth)
        for k, t in enumerate(plan.tiles_of(r)):
            x0, y0, w, h = plan.tile_rect(t)
            img[y0:y0 + h, x0:x0 + w] = view[k, :h, :w]
    return img


def gather_to_root(local_stack, world, rank):
    """One gather of the ranks' tile stacks to rank 0 (torch tensors on the backend's device).
    Returns the list of stacks on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [local_stack]
    gathered = [torch.empty_like(local_stack) for _ in range(world)] if rank == 0 else None
    dist.gather(local_stack, gathered, dst=0)
    return gathered
