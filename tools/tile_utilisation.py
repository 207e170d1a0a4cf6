#!/usr/bin/env python3
"""Diagnostic: how much SIMD lane time does ray-length variance inside a wavefront cost?  Renders per-pixel work
counters on the GPU (LT_RENDER_FLAG_PIXEL_COUNTERS) for the bench scene and reports, per 8x8 pixel square (= one
wavefront of the one-lane-per-pixel kernel), mean(node visits) / max(node visits)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lens_trace_amd import scene as sc, synth  # noqa: E402
from lens_trace_amd.renderer import RendererHIP, RenderPropertiesHIP  # noqa: E402

W, H = 3840, 2160
scene = synth.heightfield_wall(708) if len(sys.argv) < 2 or sys.argv[1] == "wall" else synth.triangle_soup(1000000)
r = RendererHIP(0)
out = np.zeros((H, W, 4), dtype=np.float32)
r.render(RenderPropertiesHIP("accumulator.cl", (W, H, 4), out, scene, pCamera=sc.camera_with_frame(scene.camera, 1), pixelCounters=True))
nodes = out[..., 2].astype(np.float64)
rays = out[..., 0]
print("rays/pixel %.3f  nodes/pixel mean %.1f  p50 %.0f p90 %.0f p99 %.0f max %.0f" % (rays.mean(), nodes.mean(), *np.percentile(nodes, [50, 90, 99]), nodes.max()))
t = nodes.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(H // 8, W // 8, 64)
util = t.mean(axis=2) / t.max(axis=2)
print("per-wave lane utilisation bound (mean/max of node visits over the 64 pixels): mean %.3f, weighted by wave time %.3f" % (
    util.mean(), t.mean(axis=2).sum() / t.max(axis=2).sum()))
