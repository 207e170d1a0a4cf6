#!/usr/bin/env python3
"""Diagnostic: where do idle lanes come from?  Per-pixel work counters (LT_RENDER_FLAG_PIXEL_COUNTERS) of the bench frame for
`basic` (camera ray only) and `accumulator` (camera ray + shadow ray): per 8x8 square (one wavefront), mean / max of the node
visits of the camera rays and of the shadow rays separately, weighted by the wave's time (the max)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lens_trace_amd import scene as sc, synth  # noqa: E402
from lens_trace_amd.renderer import RendererHIP, RenderPropertiesHIP  # noqa: E402

W, H = 3840, 2160
name = sys.argv[1] if len(sys.argv) > 1 else "wall"
scene = {"wall": lambda: synth.heightfield_wall(708), "soup": lambda: synth.triangle_soup(1000000), "blob": synth.blob_in_box}[name]()
r = RendererHIP(0)
cam = sc.camera_with_frame(scene.camera, 1)


def counters(path):
    out = np.zeros((H, W, 4), dtype=np.float32)
    r.render(RenderPropertiesHIP(path, (W, H, 4), out, scene, pCamera=cam, pixelCounters=True))
    return out[..., 2].astype(np.float64), out[..., 0]


prim, _ = counters("basic.cl")
both, rays = counters("accumulator.cl")
shadow = both - prim


def squares(a):
    return a.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)


for label, a in (("camera rays", prim), ("shadow rays (reference visits, no any-hit)", shadow), ("both", both)):
    t = squares(a)
    print("%-46s mean %.1f visits/pixel; per-wave mean/max: time-weighted %.3f" % (label, a.mean(), t.mean(axis=1).sum() / t.max(axis=1).sum()))
ts = squares(shadow)
print("shadow rays: sum of per-wave max %.3g, sum of means %.3g -> a perfect refill would cut shadow iterations to %.0f %%" % (
    ts.max(axis=1).sum(), ts.mean(axis=1).sum(), 100 * ts.mean(axis=1).sum() / ts.max(axis=1).sum()))
