#!/usr/bin/env python3
"""Diagnostic: kernel time of ONE rank's share at N = 8 for different tile shapes (interleaved r, r+8, ...) and for a
contiguous band, camera moved off the grid planes (no centre-column stragglers): separates locality / packing effects
of the partition from load imbalance."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lens_trace_amd import _capi as C, synth  # noqa: E402
from lens_trace_amd.renderer import RendererHIP, make_desc  # noqa: E402
from lens_trace_amd.scene import camera_bytes  # noqa: E402

W, H, D, N = 3840, 2160, 3, 8
SPP = int(os.environ.get("LT_SPP", "16"))
scene = synth.heightfield_wall(708).validate()
scene.camera = camera_bytes(0.0031, 2.5047, -50.0)
program = C.program_from_path("accumulator")
r = RendererHIP(0)
r.set_scene(scene)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
buf = torch.zeros(W * H * D, dtype=torch.float32, device=dev)


def run(tile, label):
    d = make_desc(program, W, H, D, scene.camera, frame_first=1, frame_count=SPP, accumulate=True, accumulate_base=0, tile=tile)
    best = 1e9
    for _ in range(3):
        r.render_device(d, buf.data_ptr(), buf.numel() * 4, stream)
        best = min(best, r.stats()["kernel_ms"])
    print("%-46s %.3f ms  (%.3f per launch)" % (label, best, best / SPP), flush=True)


run(None, "whole image")
for tw, th in ((64, 64), (32, 32), (128, 128), (256, 64), (3840, 8), (3840, 16), (3840, 24), (480, 8), (8, 2160), (16, 2160), (64, 2160)):
    tiles = ((W + tw - 1) // tw) * ((H + th - 1) // th)
    for rank in (0, 3):
        run((tw, th, rank, N), "tile %dx%d interleaved, rank %d of 8" % (tw, th, rank))
tw, th = 3840, 8
rows = H // th
for rank in (0, 3, 7):
    # a contiguous band = 1/8 of the 8-pixel rows: first = rank * rows/8, stride 1 needs tiles_in_call limited -> use a tall tile instead
    pass
run((3840, 270, 0, 8), "contiguous band 3840x270, rank 0 of 8")
run((3840, 270, 3, 8), "contiguous band 3840x270, rank 3 of 8")
run((480, 2160, 3, 8), "contiguous strip 480x2160, rank 3 of 8")
