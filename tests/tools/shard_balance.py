#!/usr/bin/env python3
"""Diagnostic: what the tile split costs before any communication.  On ONE GPU, renders each rank's share of the bench
workload (64x64 tiles r, r+N, ...; 16 spp) for N = 1, 2, 4, 8 and prints the kernel time per share: the N-GPU step
can be no faster than the slowest share, so  t(1) / (N * max_r t_r)  bounds the scaling efficiency from above."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lens_trace_amd import _capi as C, synth  # noqa: E402
from lens_trace_amd.dist import TilePlan  # noqa: E402
from lens_trace_amd.renderer import RendererHIP, make_desc  # noqa: E402

W, H, D, SPP = 3840, 2160, 3, 16
TILE_W, TILE_H = int(os.environ.get("LT_TILE_W", "64")), int(os.environ.get("LT_TILE_H", "64"))
scene = synth.heightfield_wall(708).validate()
if os.environ.get("LT_CAM_SHIFT"):   # move the camera off the grid planes x = 0, y = 2.5 (no NaN slab tests on the centre column/row)
    from lens_trace_amd.scene import camera_bytes
    scene.camera = camera_bytes(0.0031, 2.5047, -50.0)
program = C.program_from_path("accumulator")
r = RendererHIP(0)
r.set_scene(scene)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
t1 = None
for world in (1, 2, 4, 8):
    plan = TilePlan(W, H, D, TILE_W, TILE_H, world)
    buf = torch.zeros(plan.floats_per_rank if world > 1 else W * H * D, dtype=torch.float32, device=dev)
    times = []
    for rank in range(world):
        d = make_desc(program, W, H, D, scene.camera, frame_first=1, frame_count=SPP, accumulate=True, accumulate_base=0,
                      tile=plan.desc_tile(rank) if world > 1 else None)
        best = 1e9
        for _ in range(3):
            r.render_device(d, buf.data_ptr(), buf.numel() * 4, stream)
            best = min(best, r.stats()["kernel_ms"])
        times.append(best)
    if world == 1:
        t1 = times[0]
    print("N=%d tile %dx%d: share kernel ms min %.3f max %.3f sum %.3f | t(1)/(N*max) = %.3f | %s" % (
        world, TILE_W, TILE_H, min(times), max(times), sum(times), t1 / (world * max(times)), " ".join("%.1f" % t for t in times)), flush=True)
