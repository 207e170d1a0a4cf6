#!/usr/bin/env python3
"""One rank's share of the bench frame at N = 8 (tiles 0, 8, 16, ... of the balanced plan), rendered K times back to back on one
GPU with one step in flight and with two (two contexts, two streams, alternating): what overlapping a launch's drain with the
next launch's start is worth when a step is only ~3 ms long."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from lens_trace_amd import _capi as C, synth
from lens_trace_amd.dist import TilePlan
from lens_trace_amd.renderer import RendererHIP, make_desc

W, H, D, SPP, K = 3840, 2160, 3, 16, 40
scene = synth.heightfield_wall(708).validate()
prog = C.program_from_path("accumulator")
dev = torch.device("cuda", 0)
for world in (8, 4):
    plan = TilePlan.balanced(W, H, D, world, 64)
    d = make_desc(prog, W, H, D, scene.camera, frame_first=1, frame_count=SPP, accumulate=True, accumulate_base=0, tile=plan.desc_tile(0))
    for depth in [int(x) for x in os.environ.get("LT_DEPTHS", "1,2").split(",")]:
        rs = [RendererHIP(0) for _ in range(depth)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        bufs = [torch.zeros(plan.floats_per_rank, dtype=torch.float32, device=dev) for _ in range(depth)]
        for r, s, b in zip(rs, streams, bufs):
            r.set_scene(scene)
            for _ in range(3):
                r.render_device(d, b.data_ptr(), b.numel() * 4, s.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            i = k % depth
            if k >= depth:
                rs[i].stats()        # waits for the step that used this slot before
            rs[i].render_device(d, bufs[i].data_ptr(), bufs[i].numel() * 4, streams[i].cuda_stream)
        torch.cuda.synchronize()
        print("N=%d share, %d in flight: %.3f ms per step (shadow-ray walk chosen: %s)" % (
            world, depth, (time.perf_counter() - t0) / K * 1e3, [r.stats()["shadow_packets"] for r in rs]), flush=True)
        for r in rs:
            r.close()
