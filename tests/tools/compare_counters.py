#!/usr/bin/env python3
"""Diagnostic (test infrastructure: uses the oracle): per-pixel work counters (rays, shadow rays, node visits, triangle tests) of the HIP path vs the
CPU oracle on one golden scene.  Usage: compare_counters.py <scene.ltsb> <program> <W> <H> [frame]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lens_trace_amd import scene as sc  # noqa: E402
from lens_trace_amd.renderer import RendererHIP, RenderPropertiesHIP  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

path, prog, W, H = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
frame = int(sys.argv[5]) if len(sys.argv) > 5 else 0
s = sc.load_ltsb(path)
cam = sc.camera_bytes(0, 2.5, -50, 0, 0, 0, frame)
r = RendererHIP(0)
out = np.zeros((H, W, 4), dtype=np.float32)
r.render(RenderPropertiesHIP(prog + ".cl", (W, H, 4), out, s, pCamera=cam, pixelCounters=True))
want = po.pixel_counters(s, cam, W, H, po.PROGRAMS[prog])
d = np.argwhere(out.astype(np.uint32) != want)
print("pixels*channels mismatching:", len(d), "of", want.size, " totals hip", out.sum(axis=(0, 1)), "oracle", want.sum(axis=(0, 1)))
for y, x, ch in d[:20]:
    print("y=%d x=%d ch=%d hip=%s oracle=%s" % (y, x, ch, out[y, x], want[y, x]))
if len(sys.argv) > 7:
    py, px = int(sys.argv[6]), int(sys.argv[7])
    print("probe y=%d x=%d hip=%s oracle=%s" % (py, px, out[py, px], want[py, px]))
