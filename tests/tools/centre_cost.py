#!/usr/bin/env python3
"""What the image's centre column costs: the bench frame with an unrotated camera (direction.x == 0 exactly on the centre column:
those squares' rays walk the caller's tree under the reference's NaN semantics) against a camera rotated by a hair (no such
pixel; the centre row, direction.y == 0, stays in both)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from lens_trace_amd import _capi as C, scene as sc, synth
from lens_trace_amd.renderer import RendererHIP, make_desc

s = synth.heightfield_wall(708).validate()
r = RendererHIP(0)
r.set_scene(s)
W, H = 3840, 2160
stream = torch.cuda.current_stream().cuda_stream
for yaw in (0.0, 1e-4):
    cam = sc.camera_bytes(0.0, 2.5, -50.0, yaw)
    d = make_desc(C.PROGRAM_ACCUMULATOR, W, H, 3, cam, frame_first=1, frame_count=16, accumulate=True, accumulate_base=0)
    buf = torch.zeros(r.output_floats(d), dtype=torch.float32, device="cuda:0")
    for _ in range(3):
        r.render_device(d, buf.data_ptr(), buf.numel() * 4, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r.render_device(d, buf.data_ptr(), buf.numel() * 4, stream)
    torch.cuda.synchronize()
    print("yaw %g: %.3f ms per 16-sample frame (%s)" % (yaw, (time.perf_counter() - t0) / 5 * 1e3, r.stats()["shadow_packets"]))
